// pgx_replay.h -- what the three parts of the device replay share (pgx_replay.hip: the schedule; pgx_replay_eval.hip: the evaluation kernels;
// pgx_replay_tables.hip: update / dirty lists / filing / settling / emission): the tables' layouts, the flags, the inline device helpers.
// The design is described at the top of pgx_replay.hip.
#pragma once
#include "pgx_internal.h"

namespace pgx {
namespace rp {

constexpr uint32_t NIL = 0;  // list links and reader heads are stored +1
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int END_FUZZ = 48;  // READ_END_FUZZINESS, shmr_overlap.c:36
enum { T_OVERLAP = 0, T_CONTAINS = 1, T_CONTAINED = 2 };

constexpr uint32_t NIN = 58;
// One read pair = a HOT part {key = (min rid << 32 | max rid) + 1, own = (bucket << 3 | parity << 2 | type) + 1} and a COLD part
// (its readers).  An evaluation's dependent chain only ever waits for the hot part: 16 bytes per slot, one aligned load, and a
// table of a few 100 MB that the 256 MB Infinity Cache and the TLBs hold (the 256-byte slots of round 1 made every probe a
// cold HBM access into a 2 GB table).  Registrations go to the cold part and are not waited for.
struct alignas(16) PHot {
  unsigned long long key;
  uint32_t own;
  uint32_t pad;
};
struct alignas(256) PCold {
  uint32_t cnt;       // registrations so far; the first NIN sit in in[] (bucket + 1)
  uint32_t rhead;     // readers beyond the inline ones: linked nodes
  uint32_t in[NIN];   // (a pair of overlapping 15 kb reads shares ~25-40 buckets: nearly every list fits, and is duplicate-free)
  uint32_t pad[4];
};
static_assert(sizeof(PHot) == 16 && sizeof(PCold) == 256, "pair slot = 16 hot bytes + four cold cache lines");
struct MSlot {  // one alignment: a = rid0 << 32 | rid1 (never 0), b = (q_off << 2 | dir0 << 1 | dir1) + 1, req = request number
  unsigned long long a;
  uint32_t b;
  uint32_t req;
};
struct Item {  // one insertion of a bucket's latest evaluation
  uint32_t pslot;
  uint32_t info;  // ai | pi << 8 | type << 16 | I_GUESS | I_UNFILED
  uint32_t mslot;
  uint32_t next;
};
struct RNode {
  uint32_t next, bucket;
};
constexpr uint32_t I_GUESS = 1u << 18, I_UNFILED = 1u << 19;
#ifndef PGX_SPARSE_CAP
#define PGX_SPARSE_CAP 65536
#endif
constexpr uint32_t SPARSE_CAP = PGX_SPARSE_CAP;   // buckets a sparse pass takes from the list
constexpr uint32_t LIST_CAP = 262144;    // capacity of the list (a window of a dense round is listed whole)
constexpr uint32_t DEV_LIST = 0xFFFFFFFFu, DEV_LIST_WIN = 0xFFFFFFFEu;  // nlist: the device's list, up to SPARSE_CAP / LIST_CAP entries
enum : uint32_t { OV_ITEMS = 1, OV_NODES = 2, OV_REQS = 4, OV_PAIRS = 8, OV_MEMO = 16, OV_QOFF = 32, OV_PASSES = 64 };  // Counters::overflow
constexpr uint8_t F_DUP = 1, F_GUESS = 2, F_UNFILED = 4;
constexpr uint8_t F_BIG = 8;   // a bucket of at least R::big_min entries that holds no read twice: evaluated by a whole workgroup (k_eval_big)

struct alignas(64) Counters {   // three cache lines: the arenas, the dirty statistics, the totals (an atomic holds its line's L2 channel)
  uint32_t item_top, rnode_top, nreq, overflow;
  uint32_t pad0[12];
  uint32_t ndirty, min_dirty, max_dirty, pad;
  uint32_t nbig;          // big dirty buckets the narrow evaluation kernels of this pass left to k_eval_big (R::blist)
  uint32_t nbig_total;    // buckets k_setup marked F_BIG (none: k_eval_big is never launched)
  uint32_t pad1[10];
  unsigned long long lookups, skips, evals, records;
#ifdef PGX_BIG_STATS
  uint32_t big_max_steps, big_long, big_long_n, big_evals;
  unsigned long long pad2[2];
#else
  unsigned long long pad2[4];
#endif
};

struct R {
  uint32_t nb;
  const uint32_t *bid, *bstart;
  const uint64_t *y0;
  const uint8_t *dir;
  const uint32_t *rlen;
  PHot *ph;
  PCold *pc;       // the reader list of hot slot i is pc[i >> cshift]
  uint32_t cshift;
  uint32_t pmask;
  MSlot *mt;
  uint32_t mmask;
  Item *items;
  uint32_t item_cap;
  RNode *rn;
  uint32_t rn_cap;
  pgx_align_key *rq_key;
  pgx_match *rq_res;
  uint32_t req_cap, settled;
  uint32_t memo_used;  // 0: nothing has been filed yet (the first round of the first sweep skips the memo lookups)
  uint8_t *dirty, *evaluated, *parity, *bflags, *ever;
  uint32_t *ihead, *inum, *ohead, *lookups, *skips;
  uint32_t *dlist;  // the dirty buckets, listed by k_count while there are at most LIST_CAP of them (the sparse passes run from the list)
  uint32_t *blist;  // the big ones among a pass's dirty buckets (any order; Counters::nbig of them)
  uint32_t wlist0;  // first wcur slot of the list-mode wavefronts
  uint4 *wcur;  // per wavefront of k_eval: the unused rest of its arena chunks {node cur, node end, item cur, item end}, kept across launches
  Counters *c;
  unsigned long long *spread;  // the totals (evaluations, look-ups, skips) over SPREAD cache lines: [line * 8 + {0, 1, 2}], summed by the host
  uint32_t bestn;
  uint32_t tail;           // != 0: the sweeps have become small: k_file also files the alignment every OTHER reader of a requested pair would ask
                           // for and the row's next `tail` partners
  int predict, predict2;   // margins of predict_contained (0: every pending alignment is guessed a plain overlap)
  uint32_t big_min;        // buckets from this many entries on (and without a repeated read) go to k_eval_big; 0: none do
  uint32_t dup_min;        // the same for buckets that hold a read twice (k_eval_big's LDS pair set instead of one partner at a time)
  uint32_t wbig0;          // first wcur slot of k_eval_big's wavefronts
};

__device__ __forceinline__ uint64_t mix64(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 33, h *= 0xc4ceb9fe1a85ec53ULL, h ^= h >> 33;
  return h;
}
__device__ __forceinline__ uint32_t own_enc(uint32_t j, uint32_t par, uint32_t type) { return ((j << 3) | (par << 2) | type) + 1; }
__device__ __forceinline__ uint32_t own_bucket(uint32_t v) { return (v - 1) >> 3; }
__device__ __forceinline__ uint32_t own_parity(uint32_t v) { return ((v - 1) >> 2) & 1; }
__device__ __forceinline__ uint32_t own_type(uint32_t v) { return (v - 1) & 3; }
__device__ __forceinline__ uint32_t lane_rank(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ long iabs64(long x) { return x < 0 ? -x : x; }

// acceptance test and classification of shimmer_to_overlap (shmr_overlap.c:134-160)
__device__ __forceinline__ bool classify(const pgx_match &m, uint32_t rlen0, uint32_t rlen1, uint32_t q_off, uint32_t *type) {
  const uint32_t slen0 = rlen0 - q_off, slen1 = rlen1;
  *type = T_OVERLAP;
  if (m.q_bgn < END_FUZZ && m.t_bgn < END_FUZZ &&
      (iabs64((long)slen0 - m.q_end) < END_FUZZ || iabs64((long)slen1 - m.t_end) < END_FUZZ) && m.q_end > 500 &&
      m.t_end > 500) {
    if (iabs64((long)rlen0 - ((long)m.q_end - m.q_bgn)) < END_FUZZ * 2 ||
        iabs64((long)rlen1 - ((long)m.t_end - m.t_bgn)) < END_FUZZ * 2)
      *type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
    return true;
  }
  return false;
}

// The type a pending alignment will most likely have (classify above, with the alignment's geometry predicted): the query is
// read 0 from q_off on (slen0 = rlen0 - q_off bases), the target read 1 from its start.  If the target runs out first
// (rlen1 <= slen0) its whole length is covered: contained-type.  If the query runs out first, q_end = slen0 and t_end = slen0 +
// drift: contained-type iff q_off + q_bgn < 96 (the query side) or rlen1 - slen0 < 96 + drift - t_bgn (the target side: a
// target that sticks out by less than the fuzz still counts as covered).  q_bgn / t_bgn (the first 17-base run) are a few bases,
// the drift of 1 % indels over 15 kb is ~ +-10: margins mq / mt, both 88 by default.  (Round 1 tested rlen1 <= slen0 on the
// target side: every pair with 0 < rlen1 - slen0 < ~90, 0.6 % of all, was guessed wrong -- most of the second sweep's work.)
__device__ __forceinline__ bool predict_contained(uint32_t rlen0, uint32_t rlen1, uint32_t q_off, int mq, int mt) {
  return (int)rlen1 - (int)(rlen0 - q_off) < mt || q_off < (uint32_t)mq;
}

// the slot of a read pair, inserting the key if it is new (keys never change once set, so a stale "empty" only costs a
// failed compare-and-swap)
__device__ __forceinline__ uint32_t pair_slot(const R &r, uint64_t pair) {
  const unsigned long long want = pair + 1;
  uint32_t i = (uint32_t)mix64(pair) & r.pmask;
  for (int probes = 0; probes < 1024; ++probes) {
    unsigned long long k = r.ph[i].key;
    if (k == want) return i;
    if (k == 0) {
      k = atomicCAS(&r.ph[i].key, 0ULL, want);
      if (k == 0 || k == want) return i;
    }
    i = (i + 1) & r.pmask;
  }
  atomicOr(&r.c->overflow, OV_PAIRS);
  return i;
}

// read-only lookup of a pair (speculative partners must not fill the table with pairs the walk never examines); the
// slot's first 16 bytes -- key, owner, overflow head -- arrive in one load
__device__ __forceinline__ uint32_t pair_find(const R &r, uint64_t pair, uint32_t *own) {
  const unsigned long long want = pair + 1;
  uint32_t i = (uint32_t)mix64(pair) & r.pmask;
  for (int probes = 0; probes < 1024; ++probes) {
    const uint4 h = *reinterpret_cast<const uint4 *>(&r.ph[i]);
    const unsigned long long k = (unsigned long long)h.y << 32 | h.x;
    if (k == want) {
      *own = h.z;
      return i;
    }
    if (k == 0) break;
    i = (i + 1) & r.pmask;
  }
  *own = 0;
  return NONE;
}

// read-only lookup of an alignment in the memo (nothing inserts while k_eval / k_settle / k_emit run): slot and request
__device__ __forceinline__ uint32_t memo_find(const R &r, unsigned long long a, uint32_t b, uint32_t *req) {
  uint32_t i = (uint32_t)mix64(a ^ mix64(b)) & r.mmask;
  for (int probes = 0; probes < 1024; ++probes) {
    const uint4 h = *reinterpret_cast<const uint4 *>(&r.mt[i]);
    const unsigned long long cur = (unsigned long long)h.y << 32 | h.x;
    if (cur == 0) break;
    if (cur == a && h.z == b + 1) {
      *req = h.w;
      return i;
    }
    i = (i + 1) & r.mmask;
  }
  *req = NONE;
  return NONE;
}

struct Ent {
  uint32_t rid, pos1;
};
__device__ __forceinline__ Ent entry_of(uint64_t y) { return Ent{(uint32_t)(y >> 32), (((uint32_t)y) >> 1) + 1}; }

constexpr uint32_t NCH = 256;  // reader-node arena chunk of a wavefront
constexpr uint32_t SPREAD = 256;
constexpr uint32_t ICH = 128;  // item arena piece of a wavefront slot (8 chunks of 16)

#ifndef PGX_REPLAY_GL
#define PGX_REPLAY_GL 16
#endif
constexpr int GL = PGX_REPLAY_GL;  // lanes per bucket
constexpr uint32_t GPW = 64 / GL, GPB = 256 / GL;
__device__ __forceinline__ uint64_t gbits(uint64_t wave_mask, int gbase) { return (wave_mask >> gbase) & ((1ULL << GL) - 1ULL); }

// group g of the launch -> its bucket (a range of buckets, or the dirty list)
__device__ __forceinline__ uint64_t bucket_of_group(const R &r, uint32_t lo, uint32_t hi, uint32_t nlist, uint32_t g) {
  if (nlist) {
    const uint32_t n = nlist == DEV_LIST ? min(r.c->ndirty, SPARSE_CAP) : nlist == DEV_LIST_WIN ? min(r.c->ndirty, LIST_CAP) : nlist;  // (DEV_LIST*: as many as the last count listed)
    return g < n ? (uint64_t)r.dlist[g] : (uint64_t)hi;
  }
  return (uint64_t)lo + g;
}

// readers of a pair later than bucket j become dirty (k_update only: nothing registers while it runs)
__device__ __forceinline__ void mark_readers(const R &r, uint32_t slot, uint32_t j) {
  const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
  const uint4 h1 = *reinterpret_cast<const uint4 *>(w);  // cnt, rhead, in[0], in[1]
  const uint32_t c = min(h1.x, NIN);
  // (entries are bucket + 1 <= nb: the upper test is a guard, not a rule)
  if (c > 0 && h1.z > j + 1 && h1.z <= r.nb) r.dirty[h1.z - 1] = 1;
  if (c > 1 && h1.w > j + 1 && h1.w <= r.nb) r.dirty[h1.w - 1] = 1;
  for (uint32_t q = 2; q < c; q += 8) {  // in[q .. q+8): two 16-byte loads in flight
    const uint4 a = *reinterpret_cast<const uint4 *>(w + 2 + q), b = *reinterpret_cast<const uint4 *>(w + 6 + q);
    const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k)
      if (q + k < c && x[k] > j + 1 && x[k] <= r.nb) r.dirty[x[k] - 1] = 1;
  }
  if (c < NIN) return;
  for (uint32_t nd = h1.y; nd != NIL; nd = r.rn[nd - 1].next) {
    const uint32_t rb = r.rn[nd - 1].bucket;
    if (rb > j) r.dirty[rb] = 1;
  }
}

// 128-bit masks over a row's partners / a bucket's entries (k_eval_rows, k_eval_big)
struct M128 {
  uint64_t lo, hi;
};
__device__ __forceinline__ int popc128(M128 m) { return __popcll(m.lo) + __popcll(m.hi); }
__device__ __forceinline__ bool any128(M128 m) { return (m.lo | m.hi) != 0; }
__device__ __forceinline__ int ctz128(M128 m) { return m.lo ? __builtin_ctzll(m.lo) : 64 + __builtin_ctzll(m.hi); }   // (m != 0)
__device__ __forceinline__ int nth128(M128 m, uint32_t nth) {   // position of the nth set bit (nth >= 1, nth <= popc128(m))
  const uint32_t cl = (uint32_t)__popcll(m.lo);
  uint64_t w = m.lo;
  int base = 0;
  if (nth > cl) w = m.hi, nth -= cl, base = 64;
  for (uint32_t k = 1; k < nth; ++k) w &= w - 1;
  return base + __builtin_ctzll(w);
}
__device__ __forceinline__ M128 upto128(int stop) {   // bits 0 .. stop (stop >= 127: all)
  M128 m;
  m.lo = stop >= 63 ? ~0ULL : ((2ULL << stop) - 1ULL);
  m.hi = stop < 64 ? 0ULL : (stop >= 127 ? ~0ULL : ((2ULL << (stop - 64)) - 1ULL));
  return m;
}
__device__ __forceinline__ M128 and128(M128 a, M128 b) { return M128{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ M128 andn128(M128 a, M128 b) { return M128{a.lo & ~b.lo, a.hi & ~b.hi}; }   // a & ~b
__device__ __forceinline__ M128 shr128(M128 m, uint32_t s) {   // m >> s, s <= 128
  if (s >= 128) return M128{0, 0};
  if (s >= 64) return M128{m.hi >> (s - 64), 0};
  if (s == 0) return m;
  return M128{m.lo >> s | m.hi << (64 - s), m.hi >> s};
}




// ---- the same evaluation by a WHOLE WORKGROUP, for big buckets (round 3) -----------------------------------------------------------
// A bucket of a repeat family holds ~100 entries and its walk examines ~5,000 pairs, nearly all of them "seen" skips that do not
// count towards bestn: every row scans most of its partners.  With a wavefront per bucket (k_eval_rows: four rows x 16 partners,
// a row that needs more continued alone, 64 partners per step) that is 100-200 dependent steps of ~20 us -- 3.4 ms per
// evaluation, and a sparse pass lasts as long as its largest bucket: 0.47 s of a 1.07 s step at C4 scale went there
// (profiles/r03a_kernel_stats_bench_c4s.txt), and the ~15 tail sweeps of a human-scale chunk are little else.  Here eight
// wavefronts take FOUR rows x 128 partners per step (a bucket holds at most 128 entries, so a row is always complete within its
// step): the rows are committed in order through masks exchanged in LDS, exactly like k_eval_rows' four-row form; a partner that an
// earlier row of the step found contained is dropped from the later rows' masks in place (round 4).
#ifndef PGX_BIG_NW
#define PGX_BIG_NW 8
#endif
constexpr int BIG_NW = PGX_BIG_NW;              // wavefronts per bucket: rows slot = wave / 2, partner half = wave % 2
constexpr int BIG_NR = BIG_NW / 2;              // rows of a step
constexpr uint32_t BIG_WG = 512;                // workgroups of a launch (persistent: they stride over the list / the range, 512 entries at a time; 1024 / 2048: c4s 302-305 ms against 306, c4 unchanged)
// Buckets that hold a read TWICE (tandem arrays, low-complexity runs: the same shimmer pair several times within a read) can
// meet a read pair more than once within one evaluation, and the second meeting must see the first one's insertion.  The
// narrower kernels therefore run them one partner at a time -- up to 5,000 dependent steps for a 100-entry bucket, and those
// few hundred buckets were what a sparse pass at C4 scale really waited for (3.5 ms per pass, 136 passes per step).  Here the
// pairs this evaluation has inserted so far sit in an LDS set that every probe consults, and duplicates WITHIN a step are
// found by letting the would-be inserters claim their pair in a second LDS table: the step is cut in front of the first lane (in
// walk order) whose pair an earlier lane of the same step claims, the cut row is continued alone from that partner in the next
// step -- when the insertion is in the set -- and everything before the cut is exact.  Progress per step >= one new pair, so a
// bucket of d distinct read pairs takes at most ~d steps instead of rows x partners.
constexpr uint32_t SET_CAP = 2048, CLAIM_CAP = 1024;   // LDS tables of k_eval_big (open addressing, power-of-two sizes)
constexpr uint32_t CB = 4096;  // buckets per block of the count kernels (256 lanes x 16)

// ---- the kernels (defined in pgx_replay_eval.hip / pgx_replay_tables.hip, launched by pgx_replay.hip) ----
__global__ void k_setup(R r, uint32_t *hist);
__global__ void k_init_slots(uint4 *__restrict__ wcur, uint32_t nslots, uint32_t n_dense, uint32_t wlist0, Counters *c);
__global__ void k_eval(R r, uint32_t lo, uint32_t hi, uint32_t nlist);
template <int GLT, int PW>
__global__ void k_eval_rows(R r, uint32_t lo, uint32_t hi, uint32_t nlist);
__global__ void k_eval_big(R r, uint32_t lo, uint32_t hi, uint32_t nlist);
__global__ void k_update(R r, uint32_t lo, uint32_t hi, uint32_t nlist);
__global__ void k_count_a(R r, uint32_t rlo, uint32_t rhi, uint32_t *__restrict__ blk);
__global__ void k_count_b(R r, uint32_t rlo, uint32_t rhi, const uint32_t *__restrict__ blk, uint32_t nblk);
__global__ void k_file(R r, uint32_t limit);
__global__ void k_settle(R r);
__global__ void k_emit(R r, const uint32_t *__restrict__ off, pgx_ovlp *__restrict__ out);
__global__ void k_count_pairs(const PHot *__restrict__ ph, uint32_t cap, unsigned long long *__restrict__ out);

}  // namespace rp
}  // namespace pgx
