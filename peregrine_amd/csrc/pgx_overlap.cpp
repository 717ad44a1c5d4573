// pgx_overlap.cpp -- the overlap stage: what main() of /root/reference/src/shmr_overlap.c:233-419 does for one
// chunk.  Division of labour:
//   GPU  : every ovlp_match (src/DWmatch.c:66-204) -- >90 % of the reference's time -- in bulk batches (k_align)
//   host : the parts whose RESULT ORDER is defined by sequential containers in the reference and therefore has to
//          be replayed in order: shimmer-pair records (build_map, src/shmr_utils.c:295-404), the klib-khash slot
//          order that defines the bucket visit order (src/khash.h:232-336; shmr_overlap.c:206-215), the stable
//          position sort (shmr_overlap.c:46-50,217) and the greedy best-n selection with its process-global
//          seen-pair table (shmr_overlap.c:52-180).
// The greedy is order dependent but ovlp_match is a pure function of (rid0, dir0, q_off, rid1, dir1, band), so
// the host replays the greedy optimistically ("unknown alignment => assume accepted overlap"), collects the
// alignments it asked for, runs them on the GPU, and replays with the true results until a replay asks for
// nothing new.  That last replay used only true results, hence equals the reference's record sequence.
#include <glob.h>

#include <algorithm>
#include <chrono>

#include "pgx_internal.h"

using namespace pgx;

namespace {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------------------
// small open-addressing maps (host orchestration containers; layout has no observable effect)
// ---------------------------------------------------------------------------------------------------------
static inline uint64_t mix(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 33, h *= 0xc4ceb9fe1a85ec53ULL, h ^= h >> 33;
  return h;
}

template <typename V>
struct U64Map {
  std::vector<uint64_t> keys;
  std::vector<V> vals;
  std::vector<uint8_t> used;
  size_t size = 0, cap = 0;
  void reserve_pow2(size_t c) {
    cap = 16;
    while (cap < c) cap <<= 1;
    keys.assign(cap, 0), vals.assign(cap, V()), used.assign(cap, 0), size = 0;
  }
  void clear() {
    std::fill(used.begin(), used.end(), 0);
    size = 0;
  }
  void grow() {
    std::vector<uint64_t> ok;
    std::vector<V> ov;
    std::vector<uint8_t> ou;
    ok.swap(keys), ov.swap(vals), ou.swap(used);
    const size_t oc = cap;
    reserve_pow2(oc ? oc * 2 : 16);
    for (size_t i = 0; i < oc; ++i)
      if (ou[i]) *slot(ok[i]) = ov[i];
  }
  V *find(uint64_t k) {
    if (!cap) return nullptr;
    size_t i = mix(k) & (cap - 1);
    while (used[i]) {
      if (keys[i] == k) return &vals[i];
      i = (i + 1) & (cap - 1);
    }
    return nullptr;
  }
  V *slot(uint64_t k) {  // find or insert (value default-initialised on insert)
    if ((size + 1) * 2 > cap) grow();
    size_t i = mix(k) & (cap - 1);
    while (used[i]) {
      if (keys[i] == k) return &vals[i];
      i = (i + 1) & (cap - 1);
    }
    used[i] = 1, keys[i] = k, vals[i] = V(), ++size;
    return &vals[i];
  }
};

struct AKey {
  uint64_t a, b;  // a = rid0<<32|rid1 ; b = q_off<<2|dir0<<1|dir1
  bool operator==(const AKey &o) const { return a == o.a && b == o.b; }
};
struct AKeyMap {  // alignment memo: key -> index into the result array
  std::vector<AKey> keys;
  std::vector<uint32_t> vals;
  std::vector<uint8_t> used;
  size_t size = 0, cap = 0;
  void init(size_t c) {
    cap = 1024;
    while (cap < c) cap <<= 1;
    keys.assign(cap, AKey{0, 0}), vals.assign(cap, 0), used.assign(cap, 0), size = 0;
  }
  void grow() {
    AKeyMap n;
    n.init(cap * 2);
    for (size_t i = 0; i < cap; ++i)
      if (used[i]) *n.slot(keys[i], nullptr) = vals[i];
    *this = std::move(n);
  }
  uint32_t *slot(const AKey &k, bool *inserted) {
    if ((size + 1) * 2 > cap) grow();
    size_t i = mix(k.a ^ mix(k.b)) & (cap - 1);
    while (used[i]) {
      if (keys[i] == k) {
        if (inserted) *inserted = false;
        return &vals[i];
      }
      i = (i + 1) & (cap - 1);
    }
    used[i] = 1, keys[i] = k, vals[i] = 0, ++size;
    if (inserted) *inserted = true;
    return &vals[i];
  }
};

// ---------------------------------------------------------------------------------------------------------
// klib khash slot-order emulation, keys + a dense id payload (contract: SURVEY.md 8a-11 / Appendix A1;
// behaviour of src/khash.h:232-336 with no deletions; hash src/khash.h:373; load factor 0.77 src/khash.h:180).
// ---------------------------------------------------------------------------------------------------------
struct SlotTable {
  uint32_t nb = 0, size = 0, upper = 0;
  uint64_t *keys = nullptr;
  uint32_t *ids = nullptr;
  uint8_t *used = nullptr;
  SlotTable() = default;
  SlotTable(const SlotTable &) = delete;
  SlotTable &operator=(const SlotTable &) = delete;
  SlotTable(SlotTable &&o) noexcept { *this = std::move(o); }
  SlotTable &operator=(SlotTable &&o) noexcept {
    std::swap(nb, o.nb), std::swap(size, o.size), std::swap(upper, o.upper);
    std::swap(keys, o.keys), std::swap(ids, o.ids), std::swap(used, o.used);
    return *this;
  }
  ~SlotTable() { free(keys), free(ids), free(used); }
  static uint32_t h32(uint64_t k) { return (uint32_t)(k >> 33 ^ k ^ k << 11); }

  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    uint8_t *fresh = (uint8_t *)calloc(nn, 1);
    keys = (uint64_t *)realloc(keys, (size_t)nn * 8);
    ids = (uint32_t *)realloc(ids, (size_t)nn * 4);
    const uint32_t m = nn - 1;
    for (uint32_t j = 0; j < nb; ++j) {
      if (!used[j]) continue;
      uint64_t key = keys[j];
      uint32_t id = ids[j];
      used[j] = 0;
      for (;;) {  // move the element; an occupied, not yet moved destination is evicted and carried on
        uint32_t i = h32(key) & m, step = 0;
        while (fresh[i]) i = (i + (++step)) & m;
        fresh[i] = 1;
        if (i < nb && used[i]) {
          std::swap(key, keys[i]), std::swap(id, ids[i]);
          used[i] = 0;
        } else {
          keys[i] = key, ids[i] = id;
          break;
        }
      }
    }
    free(used);
    used = fresh, nb = nn, upper = thr;
  }
  // the load check precedes the lookup, so a put of an existing key can still trigger the resize
  uint32_t put(uint64_t key, uint32_t fresh_id, bool *absent) {
    if (size >= upper) enlarge();
    const uint32_t m = nb - 1;
    uint32_t i = h32(key) & m, step = 0;
    while (used[i] && keys[i] != key) i = (i + (++step)) & m;
    if (used[i]) {
      *absent = false;
      return ids[i];
    }
    used[i] = 1, keys[i] = key, ids[i] = fresh_id, ++size;
    *absent = true;
    return fresh_id;
  }
};

// ---------------------------------------------------------------------------------------------------------
// shimmer-pair records (build_map, src/shmr_utils.c:295-404)
// ---------------------------------------------------------------------------------------------------------
struct PairRecs {
  std::vector<uint64_t> key0, key1, y0;
  std::vector<uint8_t> dir;
  size_t n() const { return key0.size(); }
};

static inline uint32_t pos_of(uint64_t y) { return (uint32_t)((y & 0xFFFFFFFFu) >> 1); }

static inline uint64_t flip_y(uint64_t y, uint64_t x, const std::vector<uint32_t> &rlen) {
  const uint32_t span = (uint32_t)(x & 0xFF), rid = (uint32_t)(y >> 32);
  const uint32_t rpos = rlen[rid] - (pos_of(y) + 1) + span - 1;  // shmr_utils.c:378-385
  return ((y & 0xFFFFFFFF00000001ULL) | (uint64_t)(rpos << 1)) ^ 1ULL;
}

void build_pairs(const pgx_mm128 *mm, size_t n, U64Map<uint32_t> &mc, const std::vector<uint32_t> &rlen,
                 const pgx_overlap_params *p, PairRecs &out) {
  const uint32_t T = (uint32_t)p->total_chunk, c = (uint32_t)p->mychunk % T;
  const uint32_t lower = (uint32_t)p->mc_lower, upper = (uint32_t)p->mc_upper;
  auto count_of = [&](const pgx_mm128 &e) -> uint32_t {
    uint32_t *v = mc.find(e.x >> 8);
    PGX_REQUIRE(v, PGX_EARG, "shimmer hash %llu missing from the MC files", (unsigned long long)(e.x >> 8));
    return *v;
  };
  size_t s = 0;
  for (; s < n; ++s) {  // first anchor: lower <= count < upper, STRICT (shmr_utils.c:311-320)
    const uint32_t cnt = count_of(mm[s]);
    if (cnt >= lower && cnt < upper) break;
  }
  if (s >= n) return;
  pgx_mm128 a = mm[s];
  for (size_t i = s + 1; i < n; ++i) {
    const pgx_mm128 b = mm[i];
    const uint32_t cnt = count_of(b);
    if (cnt < lower || cnt > upper) continue;  // inclusive upper; the anchor is not advanced (:327)
    if ((a.y >> 32) == (b.y >> 32)) {
      PGX_REQUIRE((uint32_t)(a.y >> 32) < rlen.size(), PGX_EARG, "rid %u not in the idx", (uint32_t)(a.y >> 32));
      const uint32_t gap = (uint32_t)((b.y >> 1) & 0xFFFFFFF) - (uint32_t)((a.y >> 1) & 0xFFFFFFF);
      if (gap < 100) {  // :332
        a = b;
        continue;
      }
      if ((a.x >> 8) % T == c) {  // forward record, bucket [a.x][b.x]
        out.key0.push_back(a.x), out.key1.push_back(b.x), out.y0.push_back(a.y), out.dir.push_back(0);
      }
      if ((b.x >> 8) % T == c) {  // reverse record, bucket [b.x][a.x], coordinates on the other strand
        out.key0.push_back(b.x), out.key1.push_back(a.x), out.y0.push_back(flip_y(b.y, b.x, rlen)), out.dir.push_back(1);
      }
    }
    a = b;
  }
}

// ---------------------------------------------------------------------------------------------------------
// bucket visit list: ascending slot order of both table levels, buckets with 2 < n <= ovlp_upper, each sorted
// stably by descending position (shmr_overlap.c:206-217)
// ---------------------------------------------------------------------------------------------------------
struct Entry {
  uint32_t rid, pos1;  // pos1 = lastPos + 1
  uint64_t y0;
  uint8_t dir;
};
struct Visit {
  std::vector<uint64_t> start;  // bucket b covers entries [start[b], start[b+1])
  std::vector<Entry> entries;
};

// A reusable inner table: same slot behaviour as SlotTable, but storage is recycled between key0 groups so the
// ~10^5..10^6 tiny second-level tables cost no allocation.
struct ScratchTable {
  std::vector<uint64_t> keys;
  std::vector<uint32_t> ids;
  std::vector<uint8_t> used, fresh;
  uint32_t nb = 0, size = 0, upper = 0;
  void reset() {
    if (nb) std::fill(used.begin(), used.begin() + nb, 0);
    nb = size = upper = 0;
  }
  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    if (keys.size() < nn) keys.resize(nn), ids.resize(nn), used.resize(nn, 0), fresh.resize(nn, 0);
    std::fill(fresh.begin(), fresh.begin() + nn, 0);
    const uint32_t m = nn - 1;
    for (uint32_t j = 0; j < nb; ++j) {
      if (!used[j]) continue;
      uint64_t key = keys[j];
      uint32_t id = ids[j];
      used[j] = 0;
      for (;;) {
        uint32_t i = SlotTable::h32(key) & m, step = 0;
        while (fresh[i]) i = (i + (++step)) & m;
        fresh[i] = 1;
        if (i < nb && used[i]) {
          std::swap(key, keys[i]), std::swap(id, ids[i]);
          used[i] = 0;
        } else {
          keys[i] = key, ids[i] = id;
          break;
        }
      }
    }
    std::copy(fresh.begin(), fresh.begin() + nn, used.begin());
    nb = nn, upper = thr;
  }
  uint32_t put(uint64_t key, uint32_t fresh_id, bool *absent) {
    if (size >= upper) enlarge();
    const uint32_t m = nb - 1;
    uint32_t i = SlotTable::h32(key) & m, step = 0;
    while (used[i] && keys[i] != key) i = (i + (++step)) & m;
    if (used[i]) {
      *absent = false;
      return ids[i];
    }
    used[i] = 1, keys[i] = key, ids[i] = fresh_id, ++size;
    *absent = true;
    return fresh_id;
  }
};

void build_visit(const PairRecs &pr, uint32_t ovlp_upper, Visit &v) {
  const size_t n = pr.n();
  // level 1: one sequential pass over the key0 put sequence (repeats included: a put of a present key can resize)
  SlotTable outer;
  std::vector<uint32_t> id0_of(n), cnt0;
  bool absent;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t id0 = outer.put(pr.key0[i], (uint32_t)cnt0.size(), &absent);
    if (absent) cnt0.push_back(0);
    id0_of[i] = id0;
    ++cnt0[id0];
  }
  // records grouped by key0, scan order kept inside a group
  std::vector<uint64_t> g0(cnt0.size() + 1, 0);
  for (size_t k = 0; k < cnt0.size(); ++k) g0[k + 1] = g0[k] + cnt0[k];
  std::vector<uint32_t> by0(n);
  {
    std::vector<uint64_t> fill(g0.begin(), g0.end() - 1);
    for (size_t i = 0; i < n; ++i) by0[fill[id0_of[i]]++] = (uint32_t)i;
  }
  // level 2: the inner tables are independent; emulate each on a scratch table, in outer slot order
  v.start.clear(), v.entries.clear();
  v.start.push_back(0);
  ScratchTable in;
  std::vector<uint32_t> lid, lcnt, lstart, lmem, tmp;
  for (uint32_t s0 = 0; s0 < outer.nb; ++s0) {
    if (!outer.used[s0]) continue;
    const uint32_t id0 = outer.ids[s0];
    const uint32_t *rec = by0.data() + g0[id0];
    const uint32_t m = cnt0[id0];
    if (m <= 2) continue;  // no bucket of this key0 can have more than 2 records
    in.reset();
    lid.resize(m), lcnt.clear();
    for (uint32_t r = 0; r < m; ++r) {
      const uint32_t b = in.put(pr.key1[rec[r]], (uint32_t)lcnt.size(), &absent);
      if (absent) lcnt.push_back(0);
      lid[r] = b;
      ++lcnt[b];
    }
    lstart.assign(lcnt.size() + 1, 0);
    for (size_t b = 0; b < lcnt.size(); ++b) lstart[b + 1] = lstart[b] + lcnt[b];
    lmem.resize(m);
    {
      tmp.assign(lstart.begin(), lstart.end() - 1);
      for (uint32_t r = 0; r < m; ++r) lmem[tmp[lid[r]]++] = rec[r];
    }
    for (uint32_t s1 = 0; s1 < in.nb; ++s1) {
      if (!in.used[s1]) continue;
      const uint32_t b = in.ids[s1], bn = lcnt[b];
      if (bn <= 2 || bn > ovlp_upper) continue;  // shmr_overlap.c:216
      uint32_t *mb = lmem.data() + lstart[b];
      // stable, descending by position: what glibc's merge-sort qsort yields for mp128_comp (shmr_overlap.c:46-50,217)
      for (uint32_t i = 1; i < bn; ++i) {
        const uint32_t x = mb[i];
        const uint32_t px = pos_of(pr.y0[x]);
        uint32_t j = i;
        while (j > 0 && pos_of(pr.y0[mb[j - 1]]) < px) mb[j] = mb[j - 1], --j;
        mb[j] = x;
      }
      for (uint32_t i = 0; i < bn; ++i) {
        const uint64_t y = pr.y0[mb[i]];
        v.entries.push_back(Entry{(uint32_t)(y >> 32), pos_of(y) + 1, y, pr.dir[mb[i]]});
      }
      v.start.push_back(v.entries.size());
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// greedy replay (shimmer_to_overlap, shmr_overlap.c:52-180) over the visit list with an alignment memo
// ---------------------------------------------------------------------------------------------------------
enum { T_OVERLAP = 0, T_CONTAINS = 1, T_CONTAINED = 2 };
constexpr int END_FUZZ = 48;              // READ_END_FUZZINESS, shmr_overlap.c:36
constexpr uint32_t PENDING = 0xFFFFFFFFu; // memo value of a requested, not yet computed alignment

struct Replay {
  const Visit &v;
  const std::vector<uint32_t> &rlen;
  uint32_t bestn;
  AKeyMap memo;
  std::vector<pgx_match> results;
  std::vector<pgx_align_key> requests;
  U64Map<uint8_t> seen;
  std::vector<pgx_ovlp> out;
  std::vector<uint8_t> contained;
  uint64_t n_lookup = 0, n_skip = 0;

  Replay(const Visit &vv, const std::vector<uint32_t> &rl, uint32_t bn) : v(vv), rlen(rl), bestn(bn) {
    memo.init(1 << 16);
    seen.reserve_pow2(1 << 16);
  }

  static inline int64_t iabs(int64_t x) { return x < 0 ? -x : x; }

  // one pass; returns the number of alignments requested (0 => `out` is exact)
  size_t pass() {
    seen.clear();
    out.clear();
    requests.clear();
    n_lookup = n_skip = 0;
    const size_t nb = v.start.size() - 1;
    for (size_t b = 0; b < nb; ++b) {
      const Entry *e = v.entries.data() + v.start[b];
      const size_t n = v.start[b + 1] - v.start[b];
      contained.assign(n, 0);
      for (size_t hi = n - 1; hi > 0; --hi) {
        const size_t ai = hi - 1;
        if (contained[ai]) continue;
        const uint32_t rid0 = e[ai].rid, pos0 = e[ai].pos1, rlen0 = rlen[rid0];
        size_t got = 0;
        for (size_t pi = ai + 1; pi < n && got < bestn; ++pi) {
          if (contained[pi]) continue;
          const uint32_t rid1 = e[pi].rid;
          if (rid0 == rid1) continue;
          const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
          if (uint8_t *st = seen.find(pair)) {
            if (*st == T_OVERLAP) ++got;
            ++n_skip;
            continue;
          }
          const uint32_t pos1 = e[pi].pos1, rlen1 = rlen[rid1];
          const uint32_t q_off = pos0 - pos1;
          const AKey key{(uint64_t)rid0 << 32 | rid1, (uint64_t)q_off << 2 | (uint64_t)e[ai].dir << 1 | e[pi].dir};
          bool fresh;
          uint32_t *mv = memo.slot(key, &fresh);
          ++n_lookup;
          if (fresh) {
            *mv = PENDING;
            pgx_align_key rq{rid0, rid1, q_off, e[ai].dir, e[pi].dir, {0, 0}};
            requests.push_back(rq);
          }
          if (*mv == PENDING) {  // optimistic guess: accepted, plain overlap
            ++got;
            *seen.slot(pair) = T_OVERLAP;
            continue;
          }
          const pgx_match &m = results[*mv];
          const uint32_t slen0 = rlen0 - q_off, slen1 = rlen1;
          if (m.q_bgn < END_FUZZ && m.t_bgn < END_FUZZ &&
              (iabs((int64_t)slen0 - m.q_end) < END_FUZZ || iabs((int64_t)slen1 - m.t_end) < END_FUZZ) &&
              m.q_end > 500 && m.t_end > 500) {
            uint8_t type;
            if (iabs((int64_t)rlen0 - ((int64_t)m.q_end - m.q_bgn)) < END_FUZZ * 2 ||
                iabs((int64_t)rlen1 - ((int64_t)m.t_end - m.t_bgn)) < END_FUZZ * 2) {
              if (rlen0 >= rlen1) type = T_CONTAINS, contained[pi] = 1;
              else type = T_CONTAINED, contained[ai] = 1;
            } else {
              type = T_OVERLAP;
              ++got;
            }
            *seen.slot(pair) = type;
            pgx_ovlp o;
            memset(&o, 0, sizeof(o));
            o.y0 = e[ai].y0, o.y1 = e[pi].y0, o.rl0 = rlen0, o.rl1 = rlen1;
            o.strand0 = e[ai].dir, o.strand1 = e[pi].dir, o.ovlp_type = type, o.match = m;
            out.push_back(o);
          }
          if (contained[ai]) break;
        }
      }
    }
    return requests.size();
  }

  // store the GPU results of the current request batch
  void absorb(const std::vector<pgx_match> &r) {
    for (size_t i = 0; i < requests.size(); ++i) {
      const pgx_align_key &k = requests[i];
      const AKey key{(uint64_t)k.rid0 << 32 | k.rid1, (uint64_t)k.q_off << 2 | (uint64_t)k.dir0 << 1 | k.dir1};
      *memo.slot(key, nullptr) = (uint32_t)results.size();
      results.push_back(r[i]);
    }
  }
};

void check_params(const pgx_overlap_params *p) {
  PGX_REQUIRE(p, PGX_EARG, "null params");
  PGX_REQUIRE(p->total_chunk > 0 && p->mychunk > 0 && p->mychunk <= p->total_chunk, PGX_EARG,
              "need 0 < mychunk <= total_chunk (shmr_overlap.c:328-329)");
  PGX_REQUIRE(p->align_bandwidth > 0 && p->align_bandwidth < (1 << 20), PGX_EARG, "bad align_bandwidth");
  PGX_REQUIRE(p->ovlp_upper >= 0 && p->mc_lower >= 0 && p->mc_upper >= 0, PGX_EARG, "negative bound");
}

void run_overlap(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
                 const pgx_overlap_params *p, std::vector<pgx_ovlp> &out, pgx_overlap_stats *st) {
  pgx_overlap_stats s;
  memset(&s, 0, sizeof(s));
  const double t0 = now_ms();
  double gpu_ms = 0;
  // aggregate_mm_count (shmr_utils.c:162-176)
  U64Map<uint32_t> mc;
  mc.reserve_pow2(n_counts * 2 + 16);
  for (size_t i = 0; i < n_counts; ++i) *mc.slot(counts[i].mer) += counts[i].count;
  PairRecs pr;
  build_pairs(mmers, n_mm, mc, db->rlen_by_rid, p, pr);
  s.n_pair_records = pr.n();
  const double t1 = now_ms();
  Visit visit;
  build_visit(pr, (uint32_t)p->ovlp_upper, visit);
  s.n_buckets = visit.start.size() - 1;
  if (getenv("PGX_TRACE"))
    fprintf(stderr, "[pgx] pairs %zu in %.2f ms; visit order (%llu buckets) in %.2f ms\n", pr.n(), t1 - t0,
            (unsigned long long)s.n_buckets, now_ms() - t1);
  Replay rp(visit, db->rlen_by_rid, (uint32_t)(uint8_t)p->bestn);  // bestn is a uint8_t in the reference (:245)
  for (;;) {
    const double p0 = now_ms();
    const size_t nreq = rp.pass();
    ++s.rounds;
    if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] replay pass %u: %.2f ms\n", s.rounds, now_ms() - p0);
    if (nreq == 0) break;
    const double g0 = now_ms();
    DevBuf<pgx_align_key> d_keys(nreq);
    DevBuf<pgx_match> d_res(nreq);
    d_keys.upload(rp.requests.data(), nreq);
    dev_align(db, d_keys.p, nreq, p->align_bandwidth, d_res.p);
    std::vector<pgx_match> res(nreq);
    d_res.download(res.data(), nreq);
    sync();
    gpu_ms += now_ms() - g0;
    if (getenv("PGX_TRACE"))
      fprintf(stderr, "[pgx] round %u: %zu alignments, gpu %.3f ms (records so far %zu, lookups %llu)\n", s.rounds, nreq,
              now_ms() - g0, rp.out.size(), (unsigned long long)rp.n_lookup);
    rp.absorb(res);
    s.n_align_gpu += nreq;
  }
  timing_flush();
  s.n_align_needed = rp.n_lookup;
  s.n_seen_skip = rp.n_skip;
  s.n_records = rp.out.size();
  s.gpu_ms = gpu_ms;
  s.host_ms = now_ms() - t0 - gpu_ms;
  out.swap(rp.out);
  if (st) *st = s;
}

template <typename T>
void read_counted_files(const std::string &pattern, std::vector<T> &out) {
  glob_t g;
  memset(&g, 0, sizeof(g));
  if (glob(pattern.c_str(), 0, nullptr, &g) == 0) {  // name-sorted like wordexp in shmr_overlap.c:355-384
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      std::vector<uint8_t> buf;
      if (!read_file(g.gl_pathv[i], buf) || buf.size() < 8) {
        globfree(&g);
        PGX_REQUIRE(false, PGX_EIO, "file '%s' open error", g.gl_pathv[i]);
      }
      uint64_t n;
      memcpy(&n, buf.data(), 8);
      if (8 + n * sizeof(T) > buf.size()) n = (buf.size() - 8) / sizeof(T);
      const size_t o = out.size();
      out.resize(o + n);
      if (n) memcpy(out.data() + o, buf.data() + 8, n * sizeof(T));
    }
  }
  globfree(&g);
}

}  // namespace

extern "C" {

int pgx_overlap_resident(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
                         size_t n_counts, const pgx_overlap_params *p, pgx_ovlp **out, size_t *n_out,
                         pgx_overlap_stats *stats) {
  try {
    require_ready();
    PGX_REQUIRE(db && out && n_out && (n_mm == 0 || mmers) && (n_counts == 0 || counts), PGX_EARG,
                "pgx_overlap_resident: null argument");
    check_params(p);
    std::vector<pgx_ovlp> v;
    run_overlap(db, mmers, n_mm, counts, n_counts, p, v, stats);
    *out = host_copy(v);
    *n_out = v.size();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_overlap_chunk(const char *seqdb_prefix, const char *shimmer_prefix, const char *out_path,
                      const pgx_overlap_params *p, pgx_overlap_stats *stats) {
  pgx_seqdb *db = nullptr;
  int rc = PGX_OK;
  try {
    require_ready();
    PGX_REQUIRE(seqdb_prefix && shimmer_prefix && out_path, PGX_EARG, "pgx_overlap_chunk: null argument");
    check_params(p);
    rc = pgx_seqdb_load(seqdb_prefix, &db);
    if (rc) return rc;
    std::vector<pgx_mm128> mm;
    std::vector<pgx_mm_count> mc;
    read_counted_files(std::string(shimmer_prefix) + "-[0-9]*-of-[0-9]*.dat", mm);
    read_counted_files(std::string(shimmer_prefix) + "-MC-[0-9]*-of-[0-9]*.dat", mc);
    std::vector<pgx_ovlp> v;
    run_overlap(db, mm.data(), mm.size(), mc.data(), mc.size(), p, v, stats);
    FILE *f = fopen(out_path, "wb");
    PGX_REQUIRE(f, PGX_EIO, "file '%s' open error", out_path);
    bool ok = v.empty() || fwrite(v.data(), sizeof(pgx_ovlp), v.size(), f) == v.size();
    ok = (fclose(f) == 0) && ok;
    PGX_REQUIRE(ok, PGX_EIO, "short write to '%s'", out_path);
  } catch (const Fail &f) {
    rc = f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    rc = PGX_ENOMEM;
  }
  pgx_seqdb_free(db);
  return rc;
}

}  // extern "C"
