/*
 * shimmer_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see shimmer_oracle.h).
 *
 * Own-words plain-C restatement of the reference algorithms.  It is written for clarity and for being
 * checkable line-by-line against the cited reference lines, not for speed; bench.py times it as the
 * "port" CPU baseline only when oracle/_ref (the real reference) is not available.
 *
 * Parity: PINNED against the real reference (oracle/_ref) by tests/test_oracle_vs_ref.py and against the
 * committed reference outputs in tests/golden/ by tests/test_oracle_golden.py.
 */
#define _GNU_SOURCE
#include "shimmer_oracle.h"

#include <glob.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX64 UINT64_MAX

/* ------------------------------------------------------------------------------------------------------ */
/* growable vectors                                                                                        */
/* ------------------------------------------------------------------------------------------------------ */
#define VEC_PUSH(v, T, val)                                        \
  do {                                                             \
    if ((v)->n == (v)->cap) {                                      \
      (v)->cap = (v)->cap ? (v)->cap * 2 : 64;                     \
      (v)->a = (T *)realloc((v)->a, (v)->cap * sizeof(T));         \
    }                                                              \
    (v)->a[(v)->n++] = (val);                                      \
  } while (0)

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------------ */
/* sequence codec -- src/shmr_utils.c:18-62                                                                */
/* byte p = one-hot(base p) | one-hot(complement(base len-1-p)) << 4 ; A=1 C=2 G=4 T=8, anything else 0   */
/* ------------------------------------------------------------------------------------------------------ */
static uint8_t onehot_fwd(char c) {
  switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 4;
    case 'T': case 't': return 8;
    default: return 0;
  }
}
static uint8_t onehot_rev(char c) {
  switch (c) {
    case 'A': case 'a': return 8;
    case 'C': case 'c': return 4;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 1;
    default: return 0;
  }
}
void orc_encode_biseq(uint8_t *dst, const char *seq, size_t len) {
  for (size_t p = 0; p < len; ++p) dst[p] = (uint8_t)(onehot_rev(seq[len - 1 - p]) << 4 | onehot_fwd(seq[p]));
}
void orc_decode_biseq(const uint8_t *src, char *seq, size_t len, uint8_t strand) {
  static const char nib2base[16] = {'N', 'A', 'C', 'N', 'G', 'N', 'N', 'N', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};
  for (size_t p = 0; p < len; ++p) seq[p] = nib2base[strand == 0 ? (src[p] & 0x0F) : (src[p] >> 4)];
}

/* ------------------------------------------------------------------------------------------------------ */
/* minimizer sketch -- src/mm_sketch.c                                                                     */
/* ------------------------------------------------------------------------------------------------------ */
/* src/mm_sketch.c:23-32 : invertible integer mix, every line masked to 2k bits */
uint64_t orc_hash64(uint64_t key, uint64_t mask) {
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}

/* src/mm_sketch.c:10-21 : bytes 0..3 and A/C/G/T/U (either case) are bases 0..3, everything else ambiguous */
static int nt4_of_ascii(uint8_t c) {
  if (c < 4) return c;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}
static int nt4_of_nibble(uint8_t b) {
  switch (b & 0x0F) {
    case 1: return 0;
    case 2: return 1;
    case 4: return 2;
    case 8: return 3;
    default: return 4; /* decode_biseq gives 'N' for every non-one-hot nibble (shmr_utils.c:53-54) */
  }
}

/* The literal streaming state machine of src/mm_sketch.c:70-151 with is_hpc == 0.
 * code(i) yields 0..3 or 4 (ambiguous).  "run" is the reference's l, "ring" its buf, "cur" its min. */
typedef int (*code_fn)(const void *, int);
static int code_ascii(const void *s, int i) { return nt4_of_ascii(((const uint8_t *)s)[i]); }
static int code_seqdb(const void *s, int i) { return nt4_of_nibble(((const uint8_t *)s)[i]); }

static void sketch_core(const void *src, code_fn code, int len, int w, int k, uint32_t rid, orc_mmv_t *out) {
  const uint64_t mask = (1ULL << (2 * k)) - 1, top = 2ULL * (uint64_t)(k - 1);
  uint64_t fwd = 0, rev = 0;
  orc_mm_t ring[256];
  orc_mm_t cur = {ORC_MAX64, ORC_MAX64};
  int run = 0, ring_pos = 0, cur_pos = 0;
  if (!(len > 0 && w > 0 && w < 256 && k > 0 && k <= 28)) abort(); /* mm_sketch.c:77-78 asserts */
  for (int j = 0; j < w; ++j) ring[j].x = ring[j].y = ORC_MAX64;

  for (int i = 0; i < len; ++i) {
    const int c = code(src, i);
    orc_mm_t e = {ORC_MAX64, ORC_MAX64};
    if (c < 4) {
      const int span = run + 1 < k ? run + 1 : k; /* :100 */
      fwd = (fwd << 2 | (uint64_t)c) & mask;       /* :102 */
      rev = (rev >> 2) | (3ULL ^ (uint64_t)c) << top; /* :103 */
      if (fwd == rev) continue; /* :104-105 strand-ambiguous k-mer: consumes no window slot */
      const int strand = fwd < rev ? 0 : 1;
      ++run;
      if (run >= k) { /* :108-111 (span < 256 always holds without hpc) */
        e.x = orc_hash64(strand ? rev : fwd, mask) << 8 | (uint64_t)span;
        e.y = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)strand;
      }
    } else {
      run = 0; /* :112-113 -- note: cur, ring, fwd, rev are NOT flushed */
    }
    ring[ring_pos] = e; /* :114 */
    if (run == w + k - 1 && cur.x != ORC_MAX64) { /* :116-125 first full window: ties of the pre-update min */
      for (int j = ring_pos + 1; j < w; ++j)
        if (ring[j].x == cur.x && ring[j].y != cur.y) VEC_PUSH(out, orc_mm_t, ring[j]);
      for (int j = 0; j < ring_pos; ++j)
        if (ring[j].x == cur.x && ring[j].y != cur.y) VEC_PUSH(out, orc_mm_t, ring[j]);
    }
    if (e.x <= cur.x) { /* :126-128 */
      if (run >= w + k && cur.x != ORC_MAX64) VEC_PUSH(out, orc_mm_t, cur);
      cur = e;
      cur_pos = ring_pos;
    } else if (ring_pos == cur_pos) { /* :129-147 the minimum slides out */
      if (run >= w + k - 1 && cur.x != ORC_MAX64) VEC_PUSH(out, orc_mm_t, cur);
      cur.x = ORC_MAX64;
      for (int j = ring_pos + 1; j < w; ++j)
        if (cur.x >= ring[j].x) cur = ring[j], cur_pos = j;
      for (int j = 0; j <= ring_pos; ++j)
        if (cur.x >= ring[j].x) cur = ring[j], cur_pos = j;
      if (run >= w + k - 1 && cur.x != ORC_MAX64) {
        for (int j = ring_pos + 1; j < w; ++j)
          if (cur.x == ring[j].x && cur.y != ring[j].y) VEC_PUSH(out, orc_mm_t, ring[j]);
        for (int j = 0; j <= ring_pos; ++j)
          if (cur.x == ring[j].x && cur.y != ring[j].y) VEC_PUSH(out, orc_mm_t, ring[j]);
      }
    }
    if (++ring_pos == w) ring_pos = 0; /* :148 */
  }
  if (cur.x != ORC_MAX64) VEC_PUSH(out, orc_mm_t, cur); /* :150 */
}

void orc_sketch_ascii(const char *seq, int len, int w, int k, uint32_t rid, orc_mmv_t *out) {
  sketch_core(seq, code_ascii, len, w, k, rid, out);
}
/* what shmr_index.c:158-161 computes for one read: decode_biseq(strand 0) then mm_sketch */
void orc_sketch_seqdb(const uint8_t *bytes, int len, int w, int k, uint32_t rid, orc_mmv_t *out) {
  sketch_core(bytes, code_seqdb, len, w, k, rid, out);
}

/* ------------------------------------------------------------------------------------------------------ */
/* hierarchical reduction -- src/shmr_reduce.c:27-90                                                       */
/* ------------------------------------------------------------------------------------------------------ */
void orc_reduce(const orc_mmv_t *in, orc_mmv_t *out, uint8_t rs) {
  orc_mm_t slot[256];
  uint8_t head = 0;
  uint32_t prev_rid = UINT32_MAX, offs = 0;
  uint64_t last_y = ORC_MAX64;
  memset(slot, 0xff, sizeof(slot));
  /* the reference's loop counter is uint32_t (:54,70): lists must stay below 2^32 entries */
  for (uint32_t idx = 0; idx < in->n; ++idx, ++offs) {
    const uint32_t rid = (uint32_t)(in->a[idx].y >> 32);
    if (rid != prev_rid) { /* :71-77 new read: restart the window */
      offs = 0;
      memset(slot, 0xff, (size_t)rs * sizeof(orc_mm_t));
      head = 0;
      prev_rid = rid;
    }
    slot[head] = in->a[idx]; /* :27-31 */
    head = (uint8_t)((head + 1) % rs);
    if (offs < (uint32_t)rs - 1) continue; /* :79-81 */
    /* :33-50 smallest hash (x>>8); ties go to the LOWEST SLOT INDEX, not to the oldest element */
    int best = 0;
    for (int s = 1; s < rs; ++s)
      if ((slot[s].x >> 8) < (slot[best].x >> 8)) best = s;
    if (slot[best].y != last_y) { /* :83-88 */
      VEC_PUSH(out, orc_mm_t, slot[best]);
      last_y = slot[best].y;
    }
  }
}

/* ------------------------------------------------------------------------------------------------------ */
/* klib khash behaviour (no deletions) -- src/khash.h:232-336, hash :373, load factor :180                 */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t nb, size, upper;
  uint64_t *keys, *vals;
  uint8_t *used;
} otab_t;

static inline uint32_t otab_hash(uint64_t key) { return (uint32_t)(key >> 33 ^ key ^ key << 11); }

static void otab_grow(otab_t *t) {
  const uint32_t nn = t->nb ? t->nb * 2 : 4; /* kroundup32(nb+1), floor 4 */
  const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
  if (t->size >= thr) return; /* khash.h:239 "requested size is too small" */
  uint8_t *fresh = (uint8_t *)calloc(nn, 1);
  t->keys = (uint64_t *)realloc(t->keys, (size_t)nn * sizeof(uint64_t));
  t->vals = (uint64_t *)realloc(t->vals, (size_t)nn * sizeof(uint64_t));
  const uint32_t m = nn - 1;
  for (uint32_t j = 0; j < t->nb; ++j) { /* khash.h:258-284 in-place rehash with kick-out */
    if (!t->used[j]) continue;
    uint64_t key = t->keys[j], val = t->vals[j];
    t->used[j] = 0;
    for (;;) {
      uint32_t i = otab_hash(key) & m, step = 0;
      while (fresh[i]) i = (i + (++step)) & m;
      fresh[i] = 1;
      if (i < t->nb && t->used[i]) { /* evict the not-yet-moved resident and carry it on */
        uint64_t tk = t->keys[i], tv = t->vals[i];
        t->keys[i] = key, t->vals[i] = val;
        key = tk, val = tv;
        t->used[i] = 0;
      } else {
        t->keys[i] = key, t->vals[i] = val;
        break;
      }
    }
  }
  free(t->used);
  t->used = fresh;
  t->nb = nn;
  t->upper = thr;
}

/* khash.h:295-336 : the load-factor check runs BEFORE the lookup, also for keys already present */
static uint32_t otab_put(otab_t *t, uint64_t key, int *absent) {
  if (t->size >= t->upper) otab_grow(t);
  const uint32_t m = t->nb - 1;
  uint32_t i = otab_hash(key) & m, step = 0;
  while (t->used[i] && t->keys[i] != key) i = (i + (++step)) & m;
  if (t->used[i]) {
    *absent = 0;
  } else {
    t->used[i] = 1, t->keys[i] = key, t->vals[i] = 0, ++t->size;
    *absent = 1;
  }
  return i;
}
static int otab_get(const otab_t *t, uint64_t key, uint64_t *val) {
  if (!t->nb) return 0;
  const uint32_t m = t->nb - 1;
  uint32_t i = otab_hash(key) & m, step = 0;
  while (t->used[i] && t->keys[i] != key) i = (i + (++step)) & m;
  if (!t->used[i]) return 0;
  *val = t->vals[i];
  return 1;
}
static void otab_release(otab_t *t) {
  free(t->keys), free(t->vals), free(t->used);
  memset(t, 0, sizeof(*t));
}

size_t orc_khash_order(const uint64_t *keys, size_t n, uint64_t *out) {
  otab_t t = {0};
  int absent;
  size_t m = 0;
  for (size_t i = 0; i < n; ++i) otab_put(&t, keys[i], &absent);
  for (uint32_t s = 0; s < t.nb; ++s)
    if (t.used[s]) out[m++] = t.keys[s];
  otab_release(&t);
  return m;
}

/* src/shmr_utils.c:131-160 : count x>>8, list in slot order */
void orc_count(const orc_mmv_t *in, orc_mcv_t *out) {
  otab_t t = {0};
  int absent;
  for (uint32_t idx = 0; idx < in->n; ++idx) { /* uint32 loop counter as in :132,138 */
    uint32_t s = otab_put(&t, in->a[idx].x >> 8, &absent);
    t.vals[s] += 1;
  }
  for (uint32_t s = 0; s < t.nb; ++s)
    if (t.used[s]) {
      orc_mc_t e = {t.keys[s], (uint32_t)t.vals[s], 0};
      VEC_PUSH(out, orc_mc_t, e);
    }
  otab_release(&t);
}

/* ------------------------------------------------------------------------------------------------------ */
/* banded O(ND) confirmation -- src/DWmatch.c:66-204                                                       */
/* ------------------------------------------------------------------------------------------------------ */
void orc_ovlp_match(const uint8_t *q, int32_t q_len, uint8_t q_strand, const uint8_t *t, int32_t t_len,
                    uint8_t t_strand, int32_t band, orc_match_t *out, uint64_t *bases_cmp) {
  const int qs = q_strand ? 4 : 0, ts = t_strand ? 4 : 0; /* :90-91 strand = which nibble */
  const int32_t max_d = (int32_t)(0.3 * (q_len + t_len));  /* :96 (double arithmetic) */
  const int32_t band_size = band * 2;                      /* :98 */
  int32_t *V = (int32_t *)calloc((size_t)max_d * 2 + 1, sizeof(int32_t));
  int32_t *U = (int32_t *)calloc((size_t)max_d * 2 + 1, sizeof(int32_t));
  const int32_t off = max_d;
  int32_t best_m = -1, min_k = 0, max_k = 0, x = 0, y = 0;
  uint32_t longest = 0;
  int started = 0, matched = 0;
  uint64_t cmp = 0;
  memset(out, 0, sizeof(*out));

  for (int32_t d = 0; d < max_d; ++d) {
    if (max_k - min_k > band_size) break; /* :120-122 */
    for (int32_t k = min_k; k <= max_k; k += 2) {
      if (k == min_k || (k != max_k && V[k - 1 + off] < V[k + 1 + off])) /* :125-130 */
        x = V[k + 1 + off];
      else
        x = V[k - 1 + off] + 1;
      y = x - k;
      const int32_t x1 = x, y1 = y;
      while (x < q_len && y < t_len && ((q[x] >> qs) & 0x0F) == ((t[y] >> ts) & 0x0F)) ++x, ++y; /* :135-140 */
      cmp += (uint64_t)(x - x1) + 1;
      if (x - x1 > 16 && !started) { /* :142-146 */
        out->q_bgn = x1, out->t_bgn = y1;
        started = 1;
      }
      if ((uint32_t)(x - x1) > longest) { /* :148-152 */
        longest = (uint32_t)(x - x1);
        out->q_m_end = x, out->t_m_end = y;
      }
      V[k + off] = x;
      U[k + off] = x + y;
      if (x + y > best_m) best_m = x + y;
      if (x >= q_len || y >= t_len) { /* :161-164 first diagonal to reach an end wins */
        matched = 1;
        break;
      }
    }
    /* :166-183 keep diagonals within `band` of the best, then widen by one on each side */
    int32_t nmin = max_k, nmax = min_k;
    for (int32_t k2 = min_k; k2 <= max_k; k2 += 2)
      if (U[k2 + off] >= best_m - band) {
        if (k2 < nmin) nmin = k2;
        if (k2 > nmax) nmax = k2;
      }
    max_k = nmax + 1;
    min_k = nmin - 1;
    if (matched) { /* :185-194 */
      out->q_end = x, out->t_end = y, out->dist = d;
      out->m_size = (out->q_end - out->q_bgn + out->t_end - out->t_bgn + 2 * d) / 2;
      break;
    }
  }
  if (!matched) out->q_bgn = out->t_bgn = 0; /* :196-199 */
  free(V), free(U);
  if (bases_cmp) *bases_cmp += cmp;
}

/* ------------------------------------------------------------------------------------------------------ */
/* overlap stage -- src/shmr_utils.c:295-404 (build_map) + src/shmr_overlap.c:46-231                       */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { uint64_t y0, y1; uint8_t dir; } prec_t;            /* mp128_t, shimmer.h:76-79 */
typedef struct { size_t n, cap; prec_t *a; } precv_t;
typedef struct { size_t n, cap; otab_t *a; } otabv_t;
typedef struct { size_t n, cap; precv_t *a; } bucketv_t;

typedef struct {
  otab_t outer;      /* key0 (full x) -> index into inner */
  otabv_t inner;     /* key1 (full x) -> index into buckets */
  bucketv_t buckets;
} pairmap_t;

/* test hook (orc_pair_records): every insertion of build_map in the order it happens */
typedef struct { size_t n, cap; orc_pair_rec_t *a; } recv_t;
static recv_t *g_record_sink = NULL;

static void pairmap_add(pairmap_t *pm, uint64_t k0, uint64_t k1, prec_t r) {
  int absent;
  if (g_record_sink) {
    orc_pair_rec_t e;
    memset(&e, 0, sizeof(e));
    e.key0 = k0, e.key1 = k1, e.y0 = r.y0, e.dir = r.dir;
    e.npos = ~(uint32_t)((r.y0 & 0xFFFFFFFFu) >> 1);
    VEC_PUSH(g_record_sink, orc_pair_rec_t, e);
  }
  uint32_t s = otab_put(&pm->outer, k0, &absent);
  if (absent) {
    otab_t fresh = {0};
    VEC_PUSH(&pm->inner, otab_t, fresh);
    pm->outer.vals[s] = pm->inner.n - 1;
  }
  otab_t *in = &pm->inner.a[pm->outer.vals[s]];
  uint32_t s1 = otab_put(in, k1, &absent);
  if (absent) {
    precv_t fresh = {0};
    VEC_PUSH(&pm->buckets, precv_t, fresh);
    in->vals[s1] = pm->buckets.n - 1;
  }
  precv_t *b = &pm->buckets.a[in->vals[s1]];
  VEC_PUSH(b, prec_t, r);
}

static inline uint32_t pos_of(uint64_t y) { return (uint32_t)((y & 0xFFFFFFFFu) >> 1); }

/* reverse-strand coordinate of a shimmer: shmr_utils.c:376-396 */
static uint64_t flip_y(uint64_t y, uint64_t x, const uint32_t *rlen) {
  const uint32_t span = (uint32_t)(x & 0xFF), rid = (uint32_t)(y >> 32);
  const uint32_t pos = pos_of(y) + 1;
  const uint32_t rpos = rlen[rid] - pos + span - 1;
  return ((y & 0xFFFFFFFF00000001ULL) | (uint64_t)(rpos << 1)) ^ 1ULL;
}

static uint64_t build_pairmap(pairmap_t *pm, const orc_mm_t *mm, size_t n, const otab_t *mc, const uint32_t *rlen,
                              uint32_t chunk, uint32_t total, uint32_t lower, uint32_t upper) {
  size_t s = 0;
  uint64_t cnt = 0, nrec = 0;
  orc_mm_t a, b;
  for (; s < n; ++s) { /* :309-320 first anchor: lower <= count < upper (STRICT upper) */
    if (!otab_get(mc, mm[s].x >> 8, &cnt)) abort();
    if (cnt >= lower && cnt < upper) break;
  }
  if (s >= n) return 0;
  a = mm[s];
  for (size_t i = s + 1; i < n; ++i) {
    b = mm[i];
    if (!otab_get(mc, b.x >> 8, &cnt)) abort();
    if (cnt < lower || cnt > upper) continue; /* :327 inclusive upper; anchor NOT advanced */
    if ((a.y >> 32) == (b.y >> 32)) {
      /* :332 unsigned 32-bit difference of positions masked to 28 bits */
      const uint32_t gap = (uint32_t)((b.y >> 1) & 0xFFFFFFF) - (uint32_t)((a.y >> 1) & 0xFFFFFFF);
      if (gap < 100) {
        a = b;
        continue;
      }
      if ((a.x >> 8) % total == chunk % total) { /* :337-359 forward record */
        prec_t r = {a.y, b.y, 0};
        pairmap_add(pm, a.x, b.x, r);
        ++nrec;
      }
      if ((b.x >> 8) % total == chunk % total) { /* :362-400 reverse record */
        prec_t r = {flip_y(b.y, b.x, rlen), flip_y(a.y, a.x, rlen), 1};
        pairmap_add(pm, b.x, a.x, r);
        ++nrec;
      }
    }
    a = b; /* :402 */
  }
  return nrec;
}

enum { T_OVERLAP = 0, T_CONTAINS = 1, T_CONTAINED = 2 };
#define END_FUZZ 48 /* shmr_overlap.c:36 */

static inline int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

/* shmr_overlap.c:52-180 greedy best-n inside one (already sorted) bucket */
static void bucket_to_overlaps(const precv_t *b, const uint32_t *rlen, const uint64_t *roff, const uint8_t *seqdb,
                               otab_t *seen, uint32_t bestn, uint32_t band, orc_ovlpv_t *out, orc_stats_t *st) {
  const size_t n = b->n;
  uint8_t *contained = (uint8_t *)calloc(n, 1);
  int absent;
  for (size_t hi = n - 1; hi > 0; --hi) { /* anchors n-2 .. 0 */
    const size_t ai = hi - 1;
    if (contained[ai]) continue;
    const uint64_t ya = b->a[ai].y0;
    const uint32_t rid0 = (uint32_t)(ya >> 32), pos0 = pos_of(ya) + 1, rlen0 = rlen[rid0];
    const uint8_t strand0 = b->a[ai].dir;
    const uint8_t *seq0 = seqdb + roff[rid0];
    size_t got = 0;
    for (size_t pi = ai + 1; pi < n && got < bestn; ++pi) {
      if (contained[pi]) continue;
      const uint64_t yb = b->a[pi].y0;
      const uint32_t rid1 = (uint32_t)(yb >> 32);
      if (rid0 == rid1) continue;
      const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
      uint64_t seen_type;
      if (otab_get(seen, pair, &seen_type)) { /* :103-107 */
        if (seen_type == T_OVERLAP) ++got;
        ++st->n_seen_skip;
        continue;
      }
      const uint32_t pos1 = pos_of(yb) + 1, rlen1 = rlen[rid1];
      const uint8_t strand1 = b->a[pi].dir;
      const uint32_t slen0 = rlen0 - pos0 + pos1, slen1 = rlen1;
      orc_match_t m;
      orc_ovlp_match(seq0 + pos0 - pos1, (int32_t)slen0, strand0, seqdb + roff[rid1], (int32_t)slen1, strand1,
                     (int32_t)band, &m, &st->bases_cmp);
      ++st->n_align;
      if (m.q_bgn < END_FUZZ && m.t_bgn < END_FUZZ &&
          (iabs64((int64_t)slen0 - m.q_end) < END_FUZZ || iabs64((int64_t)slen1 - m.t_end) < END_FUZZ) &&
          m.q_end > 500 && m.t_end > 500) { /* :134-137 */
        uint8_t type;
        if (iabs64((int64_t)rlen0 - ((int64_t)m.q_end - m.q_bgn)) < END_FUZZ * 2 ||
            iabs64((int64_t)rlen1 - ((int64_t)m.t_end - m.t_bgn)) < END_FUZZ * 2) { /* :142-154 */
          if (rlen0 >= rlen1) type = T_CONTAINS, contained[pi] = 1;
          else type = T_CONTAINED, contained[ai] = 1;
        } else {
          type = T_OVERLAP;
          ++got;
        }
        uint32_t s = otab_put(seen, pair, &absent);
        if (!absent) abort(); /* :161 */
        seen->vals[s] = type;
        orc_ovlp_t o;
        memset(&o, 0, sizeof(o));
        o.y0 = ya, o.y1 = yb, o.rl0 = rlen0, o.rl1 = rlen1;
        o.strand0 = strand0, o.strand1 = strand1, o.ovlp_type = type, o.match = m;
        VEC_PUSH(out, orc_ovlp_t, o);
      }
      if (contained[ai]) break; /* :176 */
    }
  }
  free(contained);
}

/* stable, descending by position: what glibc's merge-sort qsort does with mp128_comp (shmr_overlap.c:46-50,217) */
static void sort_bucket(precv_t *b) {
  for (size_t i = 1; i < b->n; ++i) {
    prec_t v = b->a[i];
    size_t j = i;
    while (j > 0 && pos_of(b->a[j - 1].y0) < pos_of(v.y0)) b->a[j] = b->a[j - 1], --j;
    b->a[j] = v;
  }
}

/* the insertion sequence of build_map (src/shmr_utils.c:295-404) for one overlap chunk: what a rank of the multi-GPU job must
 * have received, in order, after the record exchange.  Returns a malloc'd array (orc_free). */
orc_pair_rec_t *orc_pair_records(const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts, const uint32_t *rlen,
                                 uint32_t mychunk, uint32_t total_chunk, uint32_t mc_lower, uint32_t mc_upper, size_t *n_out) {
  otab_t mc = {0};
  pairmap_t pm;
  recv_t sink = {0};
  int absent;
  memset(&pm, 0, sizeof(pm));
  for (size_t i = 0; i < n_counts; ++i) { /* aggregate_mm_count, shmr_utils.c:162-176 */
    uint32_t s = otab_put(&mc, counts[i].mer, &absent);
    mc.vals[s] += counts[i].count;
  }
  g_record_sink = &sink;
  build_pairmap(&pm, mmers, n_mm, &mc, rlen, mychunk, total_chunk, mc_lower, mc_upper);
  g_record_sink = NULL;
  *n_out = sink.n;
  return sink.a ? sink.a : (orc_pair_rec_t *)malloc(1);
}

int orc_overlap(const uint8_t *seqdb, const uint32_t *rlen, const uint64_t *roff, uint32_t nreads,
                const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts, uint32_t mychunk,
                uint32_t total_chunk, uint32_t mc_lower, uint32_t mc_upper, uint32_t bestn, uint32_t ovlp_upper,
                uint32_t band, orc_ovlpv_t *out, orc_stats_t *stats) {
  (void)nreads;
  orc_stats_t st = {0};
  otab_t mc = {0}, seen = {0};
  pairmap_t pm;
  int absent;
  memset(&pm, 0, sizeof(pm));
  for (size_t i = 0; i < n_counts; ++i) { /* aggregate_mm_count, shmr_utils.c:162-176 */
    uint32_t s = otab_put(&mc, counts[i].mer, &absent);
    mc.vals[s] += counts[i].count;
  }
  st.n_records = build_pairmap(&pm, mmers, n_mm, &mc, rlen, mychunk, total_chunk, mc_lower, mc_upper);
  /* process_overlaps, shmr_overlap.c:206-228: ascending slot order of both table levels */
  for (uint32_t s0 = 0; s0 < pm.outer.nb; ++s0) {
    if (!pm.outer.used[s0]) continue;
    otab_t *in = &pm.inner.a[pm.outer.vals[s0]];
    for (uint32_t s1 = 0; s1 < in->nb; ++s1) {
      if (!in->used[s1]) continue;
      precv_t *b = &pm.buckets.a[in->vals[s1]];
      if (b->n <= 2 || b->n > ovlp_upper) continue; /* :216 */
      sort_bucket(b);
      bucket_to_overlaps(b, rlen, roff, seqdb, &seen, bestn, band, out, &st);
      ++st.n_buckets;
    }
  }
  for (size_t i = 0; i < pm.buckets.n; ++i) free(pm.buckets.a[i].a);
  for (size_t i = 0; i < pm.inner.n; ++i) otab_release(&pm.inner.a[i]);
  free(pm.buckets.a), free(pm.inner.a);
  otab_release(&pm.outer), otab_release(&mc), otab_release(&seen);
  if (stats) *stats = st;
  return 0;
}

/* ------------------------------------------------------------------------------------------------------ */
/* files -- formats of src/shmr_utils.c:98-123,178-203 ; idx lines of src/shmr_mkseqdb.c:111-112           */
/* ------------------------------------------------------------------------------------------------------ */
typedef struct { uint32_t n; uint32_t *rid; uint32_t *len; uint64_t *off; uint32_t max_rid; } idx_t;

static int idx_load(const char *path, idx_t *ix) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  size_t cap = 0;
  char name[256];
  uint32_t rid, len;
  unsigned long off;
  memset(ix, 0, sizeof(*ix));
  while (fscanf(f, "%u %255s %u %lu", &rid, name, &len, &off) == 4) { /* shmr_utils.c:259-260 */
    if (ix->n == cap) {
      cap = cap ? cap * 2 : 1024;
      ix->rid = (uint32_t *)realloc(ix->rid, cap * 4);
      ix->len = (uint32_t *)realloc(ix->len, cap * 4);
      ix->off = (uint64_t *)realloc(ix->off, cap * 8);
    }
    ix->rid[ix->n] = rid, ix->len[ix->n] = len, ix->off[ix->n] = off;
    if (rid > ix->max_rid) ix->max_rid = rid;
    ++ix->n;
  }
  fclose(f);
  return 0;
}
static void idx_free(idx_t *ix) { free(ix->rid), free(ix->len), free(ix->off); }

static uint8_t *slurp(const char *path, size_t *size) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t *buf = (uint8_t *)malloc(sz > 0 ? (size_t)sz : 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) {
    free(buf), fclose(f);
    return NULL;
  }
  fclose(f);
  *size = (size_t)sz;
  return buf;
}

static int mmlist_write(const char *path, const orc_mmv_t *v) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  uint64_t n = v->n;
  fwrite(&n, 8, 1, f);
  fwrite(v->a, sizeof(orc_mm_t), v->n, f);
  fclose(f);
  return 0;
}
static int mc_write(const char *path, const orc_mcv_t *v) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  uint64_t n = v->n;
  fwrite(&n, 8, 1, f);
  fwrite(v->a, sizeof(orc_mc_t), v->n, f);
  fclose(f);
  return 0;
}
static int write_level(const char *prefix, int level, int chunk, int total, const orc_mmv_t *v) {
  char path[8300];
  orc_mcv_t mc = {0};
  snprintf(path, sizeof(path), "%s-L%d-%02d-of-%02d.dat", prefix, level, chunk, total);
  if (mmlist_write(path, v)) return -1;
  orc_count(v, &mc);
  snprintf(path, sizeof(path), "%s-L%d-MC-%02d-of-%02d.dat", prefix, level, chunk, total);
  int rc = mc_write(path, &mc);
  free(mc.a);
  return rc;
}

/* shmr_index.c:37-245 */
int orc_index_chunk(const char *seqdb_prefix, const char *out_prefix, int total, int mychunk, int levels,
                    int reduction, int write_l0, int w, int k, uint64_t *bases_done) {
  char path[8300];
  idx_t ix;
  size_t dbsize = 0;
  uint64_t bases = 0;
  if (!(total > 0 && mychunk > 0 && mychunk <= total && reduction < 256 && w >= 24 && k >= 12 && w > k)) return -1;
  snprintf(path, sizeof(path), "%s.idx", seqdb_prefix);
  if (idx_load(path, &ix)) return -1;
  snprintf(path, sizeof(path), "%s.seqdb", seqdb_prefix);
  uint8_t *db = slurp(path, &dbsize);
  if (!db) return idx_free(&ix), -1;
  orc_mmv_t l0 = {0}, l1 = {0}, l2 = {0};
  for (uint32_t i = 0; i < ix.n; ++i) { /* :155-163 */
    if (ix.rid[i] % (uint32_t)total != (uint32_t)mychunk % (uint32_t)total) continue;
    orc_sketch_seqdb(db + ix.off[i], (int)ix.len[i], w, k, ix.rid[i], &l0);
    bases += ix.len[i];
  }
  int rc = 0;
  if (write_l0 == 1) rc |= write_level(out_prefix, 0, mychunk, total, &l0);
  orc_reduce(&l0, &l1, (uint8_t)reduction);
  if (levels == 1) {
    rc |= write_level(out_prefix, 1, mychunk, total, &l1);
  } else if (levels > 1) {
    orc_reduce(&l1, &l2, (uint8_t)reduction);
    rc |= write_level(out_prefix, 2, mychunk, total, &l2);
  }
  free(l0.a), free(l1.a), free(l2.a), free(db), idx_free(&ix);
  if (bases_done) *bases_done = bases;
  return rc ? -1 : 0;
}

/* shmr_overlap.c:233-419 */
int orc_overlap_chunk(const char *seqdb_prefix, const char *shimmer_prefix, const char *out_path, int total,
                      int mychunk, int bestn, int mc_lower, int mc_upper, int band, int ovlp_upper,
                      orc_stats_t *stats, uint64_t *n_out) {
  char path[8300];
  idx_t ix;
  size_t dbsize = 0;
  if (!(total > 0 && mychunk > 0 && mychunk <= total)) return -1;
  snprintf(path, sizeof(path), "%s.idx", seqdb_prefix);
  if (idx_load(path, &ix)) return -1;
  snprintf(path, sizeof(path), "%s.seqdb", seqdb_prefix);
  uint8_t *db = slurp(path, &dbsize);
  if (!db) return idx_free(&ix), -1;
  uint32_t *rlen = (uint32_t *)calloc((size_t)ix.max_rid + 1, 4);
  uint64_t *roff = (uint64_t *)calloc((size_t)ix.max_rid + 1, 8);
  for (uint32_t i = 0; i < ix.n; ++i) rlen[ix.rid[i]] = ix.len[i], roff[ix.rid[i]] = ix.off[i];

  orc_mmv_t mm = {0};
  orc_mcv_t mc = {0};
  glob_t g;
  /* :355-370 every index chunk's list, in glob (name-sorted) order */
  snprintf(path, sizeof(path), "%s-[0-9]*-of-[0-9]*.dat", shimmer_prefix);
  if (glob(path, 0, NULL, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      size_t sz = 0;
      uint8_t *buf = slurp(g.gl_pathv[i], &sz);
      if (!buf || sz < 8) { free(buf); continue; }
      uint64_t n;
      memcpy(&n, buf, 8);
      for (uint64_t j = 0; j < n; ++j) {
        orc_mm_t e;
        memcpy(&e, buf + 8 + 16 * j, 16);
        VEC_PUSH(&mm, orc_mm_t, e);
      }
      free(buf);
    }
    globfree(&g);
  }
  /* :372-384 every MC file */
  snprintf(path, sizeof(path), "%s-MC-[0-9]*-of-[0-9]*.dat", shimmer_prefix);
  if (glob(path, 0, NULL, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      size_t sz = 0;
      uint8_t *buf = slurp(g.gl_pathv[i], &sz);
      if (!buf || sz < 8) { free(buf); continue; }
      uint64_t n;
      memcpy(&n, buf, 8);
      for (uint64_t j = 0; j < n; ++j) {
        orc_mc_t e;
        memcpy(&e, buf + 8 + 16 * j, 16);
        e.pad = 0;
        VEC_PUSH(&mc, orc_mc_t, e);
      }
      free(buf);
    }
    globfree(&g);
  }
  orc_ovlpv_t out = {0};
  orc_overlap(db, rlen, roff, ix.max_rid + 1, mm.a, mm.n, mc.a, mc.n, (uint32_t)mychunk, (uint32_t)total,
              (uint32_t)mc_lower, (uint32_t)mc_upper, (uint32_t)(uint8_t)bestn, (uint32_t)ovlp_upper, (uint32_t)band,
              &out, stats);
  int rc = 0;
  FILE *f = fopen(out_path, "wb");
  if (!f) rc = -1;
  else {
    fwrite(out.a, sizeof(orc_ovlp_t), out.n, f);
    fclose(f);
  }
  if (n_out) *n_out = out.n;
  free(out.a), free(mm.a), free(mc.a), free(rlen), free(roff), free(db), idx_free(&ix);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------ */
/* row f2 -- src/shmr_dedup.c:32-101 (on a non-empty stream; the reference's feof loop re-reads the last   */
/* record once, which the pair table then drops)                                                           */
/* ------------------------------------------------------------------------------------------------------ */
char *orc_dedup(const orc_ovlp_t *recs, size_t n, size_t *text_len, uint64_t *n_unique) {
  otab_t seen = {0};
  size_t cap = 1 << 16, len = 0;
  char *out = (char *)malloc(cap);
  uint64_t uniq = 0;
  int absent;
  for (size_t i = 0; i < n; ++i) {
    const orc_ovlp_t *o = &recs[i];
    const uint32_t rid0 = (uint32_t)(o->y0 >> 32), rid1 = (uint32_t)(o->y1 >> 32);
    const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
    uint64_t dummy;
    if (otab_get(&seen, pair, &dummy)) continue; /* :41-42 */
    const uint32_t pos0 = pos_of(o->y0) + 1, pos1 = pos_of(o->y1) + 1, rlen0 = o->rl0, rlen1 = o->rl1;
    int32_t q_bgn = o->match.q_bgn, q_end = o->match.q_end, t_bgn = o->match.t_bgn, t_end = o->match.t_end;
    uint32_t a_bgn, a_end, b_bgn, b_end;
    q_bgn -= t_bgn; /* :62-63 */
    t_bgn = 0;
    if (o->strand0 == 0) { /* :64-69 -- the variables are uint32_t in the reference: "< 0" never fires */
      a_bgn = (uint32_t)((int32_t)(pos0 - pos1) + q_bgn);
      a_end = (uint32_t)((int32_t)(pos0 - pos1) + q_end);
    } else { /* :70-77 */
      a_bgn = (uint32_t)((int32_t)rlen0 - (int32_t)(pos0 - pos1) - q_end);
      a_end = (uint32_t)((int32_t)rlen0 - (int32_t)(pos0 - pos1) - q_bgn);
    }
    a_end = a_end >= rlen0 ? rlen0 : a_end;
    if (o->strand1 == 0) { /* :78-82 */
      b_bgn = (uint32_t)t_bgn, b_end = (uint32_t)t_end;
    } else { /* :83-88 */
      b_bgn = (uint32_t)((int32_t)rlen1 - t_end), b_end = (uint32_t)((int32_t)rlen1 - t_bgn);
    }
    b_end = b_end >= rlen1 ? rlen1 : b_end;
    const double err_est = 100.0 - 100.0 * (double)o->match.dist / (double)o->match.m_size; /* :89-90 */
    if (len + 256 > cap) out = (char *)realloc(out, cap *= 2);
    len += (size_t)snprintf(out + len, 256, "%09d %09d %d %0.1f %u %d %d %u %u %d %d %u %s\n", (int)rid0, (int)rid1,
                            -(o->match.m_size), err_est, 0u, (int)a_bgn, (int)a_end, rlen0,
                            (unsigned)(o->strand0 == 0 ? o->strand1 : 1 - o->strand1), (int)b_bgn, (int)b_end, rlen1,
                            o->ovlp_type == 0 ? "overlap" : (o->ovlp_type == 1 ? "contains" : "contained"));
    otab_put(&seen, pair, &absent);
    ++uniq;
  }
  otab_release(&seen);
  if (text_len) *text_len = len;
  if (n_unique) *n_unique = uniq;
  return out;
}

/* ------------------------------------------------------------------------------------------------------ */
/* row f4 -- the shimmer4py query helpers, src/shimmer4py.c:44-196 (build_shimmer_map4py, get_shimmers_for_read, */
/* get_mmer_count, get_shimmer_hits) over the same pair map as the overlap stage                            */
/* ------------------------------------------------------------------------------------------------------ */
struct orc_map {
  otab_t mc;
  pairmap_t pm;
  uint64_t n_rec;
};

orc_map_t *orc_map_build(const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts, const uint32_t *rlen,
                         uint32_t mychunk, uint32_t total_chunk, uint32_t lower, uint32_t upper) {
  orc_map_t *m = (orc_map_t *)calloc(1, sizeof(*m));
  int absent;
  for (size_t i = 0; i < n_counts; ++i) { /* aggregate_mm_count, shmr_utils.c:162-176 (shimmer4py.c:109-116) */
    uint32_t s = otab_put(&m->mc, counts[i].mer, &absent);
    m->mc.vals[s] += counts[i].count;
  }
  m->n_rec = build_pairmap(&m->pm, mmers, n_mm, &m->mc, rlen, mychunk, total_chunk, lower, upper); /* :122-123 */
  return m;
}

void orc_map_free(orc_map_t *m) {
  if (!m) return;
  for (size_t i = 0; i < m->pm.buckets.n; ++i) free(m->pm.buckets.a[i].a);
  for (size_t i = 0; i < m->pm.inner.n; ++i) otab_release(&m->pm.inner.a[i]);
  free(m->pm.buckets.a), free(m->pm.inner.a);
  otab_release(&m->pm.outer), otab_release(&m->mc);
  free(m);
}

/* get_mmer_count, shimmer4py.c:148-156 */
uint32_t orc_map_count(const orc_map_t *m, uint64_t mhash) {
  uint64_t v;
  return otab_get(&m->mc, mhash, &v) ? (uint32_t)v : 0;
}

/* get_shimmer_hits, shimmer4py.c:158-196: the key0's inner table in ascending slot order, every bucket sorted (stably, */
/* position descending: the comparator of :38-42 under glibc's merge sort) and appended                                */
size_t orc_map_hits(orc_map_t *m, uint64_t mhash0, uint32_t span, orc_mp256_t **out) {
  const uint64_t key0 = mhash0 << 8 | span;
  uint64_t v;
  size_t n = 0, cap = 0;
  orc_mp256_t *a = NULL;
  if (otab_get(&m->pm.outer, key0, &v)) {
    otab_t *in = &m->pm.inner.a[v];
    for (uint32_t s1 = 0; s1 < in->nb; ++s1) {
      if (!in->used[s1]) continue;
      precv_t *b = &m->pm.buckets.a[in->vals[s1]];
      sort_bucket(b);
      for (size_t j = 0; j < b->n; ++j) {
        if (n == cap) a = (orc_mp256_t *)realloc(a, (cap = cap ? cap * 2 : 16) * sizeof(*a));
        memset(&a[n], 0, sizeof(a[n]));
        a[n].x0 = key0, a[n].x1 = in->keys[s1], a[n].y0 = b->a[j].y0, a[n].y1 = b->a[j].y1, a[n].direction = b->a[j].dir;
        ++n;
      }
    }
  }
  *out = a;
  return n;
}

/* get_ridmm + get_shimmers_for_read, shmr_utils.c:415-443, shimmer4py.c:134-146: a read's shimmers are the run that */
/* starts at its first occurrence and is as long as its number of occurrences in the whole list                       */
void orc_read_shimmers(const orc_mm_t *mmers, size_t n_mm, uint32_t rid, size_t *first, size_t *count) {
  size_t f = 0, c = 0;
  for (size_t s = 0; s < n_mm; ++s)
    if ((uint32_t)(mmers[s].y >> 32) == rid) {
      if (!c) f = s;
      ++c;
    }
  *first = f, *count = c;
}

/* ------------------------------------------------------------------------------------------------------ */
/* row f3 -- shmr_map, src/shmr_map.c:48-161 (process_map) after the build_map of :350-351                  */
/* ------------------------------------------------------------------------------------------------------ */
char *orc_map_reads_to_ref(const orc_mm_t *ref, size_t n_ref, const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts,
                           size_t n_counts, const uint32_t *rlen, uint32_t mychunk, uint32_t total_chunk, uint32_t lower,
                           uint32_t upper, size_t *text_len, uint64_t *n_lines) {
  orc_map_t *m = orc_map_build(mmers, n_mm, counts, n_counts, rlen, mychunk, total_chunk, lower, upper);
  size_t cap = 1 << 16, len = 0;
  char *out = (char *)malloc(cap);
  uint64_t lines = 0, v, c0, c1;
  size_t s = 0;
  for (; s < n_ref; ++s) /* :84-90 first reference shimmer that is a key0 of the map */
    if (otab_get(&m->pm.outer, ref[s].x, &v)) break;
  if (s < n_ref) {
    orc_mm_t a = ref[s], b;
    for (size_t i = s + 1; i < n_ref; ++i) {
      b = ref[i];
      if (!otab_get(&m->mc, b.x >> 8, &c1)) continue;              /* :96 unknown to the reads: anchor kept */
      if (c1 < lower || c1 > upper) continue;                       /* :98 */
      if ((a.y >> 32) != (b.y >> 32)) { a = b; continue; }          /* :100-103 different contigs */
      if (!otab_get(&m->pm.outer, a.x, &v)) { a = b; continue; }    /* :105-109 */
      otab_t *in = &m->pm.inner.a[v];
      if (!otab_get(in, b.x, &v)) { a = b; continue; }              /* :111-116 */
      /* :118 64-bit difference of the positions masked to 28 bits; a negative gap wraps to a huge value */
      if ((((b.y >> 1) & 0xFFFFFFF) - ((a.y >> 1) & 0xFFFFFFF)) < 100) { a = b; continue; }
      const precv_t *bk = &m->pm.buckets.a[v];                      /* insertion order: no sort here */
      if (!otab_get(&m->mc, a.x >> 8, &c0)) abort();                /* :147 */
      for (size_t j = 0; j < bk->n; ++j) {
        if (len + 160 > cap) out = (char *)realloc(out, cap *= 2);
        len += (size_t)snprintf(out + len, 160, "%u %u %u %u %u %u %d %u %u\n", (uint32_t)(a.y >> 32), pos_of(a.y), pos_of(b.y),
                                (uint32_t)(bk->a[j].y0 >> 32), pos_of(bk->a[j].y0), pos_of(bk->a[j].y1), (int)bk->a[j].dir,
                                (uint32_t)c0, (uint32_t)c1);
        ++lines;
      }
      a = b;
    }
  }
  orc_map_free(m);
  if (text_len) *text_len = len;
  if (n_lines) *n_lines = lines;
  return out;
}
