// pgx_replay.hip -- the greedy best-n walk over the shimmer-pair buckets (shimmer_to_overlap + the seen-pair table of
// build_ovlp, /root/reference/src/shmr_overlap.c:52-228) on the GPU, one lane per bucket.
//
// The reference visits the buckets once, in order, sharing one seen-pair table.  A bucket's evaluation is a pure function of
// (a) which of the pairs it examines were inserted by EARLIER buckets (and with which type) and (b) the alignment results it
// looks up.  So the walk is solved as a fixed point (the same formulation as the host replay in pgx_overlap.cpp, which stays
// as the fallback), in phases separated by kernel boundaries -- no intra-kernel ordering is needed anywhere:
//
//   k_eval    every dirty bucket of a window re-runs shimmer_to_overlap against the pair table AS IT STANDS (read-only in
//             this kernel): pair present <=> owner < bucket.  It registers itself as a reader of every pair it examines
//             (lock-free push), guesses unknown alignments (accepted; type predicted from the geometry) and leaves the
//             list of pairs it would insert.
//   k_update  applies the evaluated buckets' lists to the table (compare-and-swap, lowest bucket wins), withdraws what a
//             bucket no longer inserts, and marks dirty every later reader of a pair whose state changed, the displaced
//             owner, and the bucket itself when an earlier bucket got in first.
//   ... repeated window by window until no bucket is dirty; then
//   k_file    files the alignments the lists still need in the memo table (insert-only) and numbers the requests,
//   dev_align the banded O(ND) kernel (pgx_align.hip) on the new requests,
//   k_settle  checks every guess against its result; a wrong guess makes the bucket dirty again
//   ... until a sweep files nothing or every guess was right; k_emit writes the ovlp_t records in bucket order.
//
// At the fixed point every bucket was last evaluated against a table that has not changed since in any way it could
// observe, and the owner of a pair is the lowest bucket inserting it: by induction over the bucket order that is the
// sequential walk.  Changes only ever propagate to LATER buckets, so the lowest unstable bucket rises monotonically.
#include <chrono>
#include <optional>

#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {
namespace {

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

// ---- flags: buckets holding a read more than once (only those can meet a pair twice within one evaluation) ----------
__global__ __launch_bounds__(256) void k_setup(R r, uint32_t *hist) {   // hist (trace only): [dup][min(n / 8, 15)] bucket counts
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= r.nb) return;
  const uint32_t b = r.bid[j], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
  bool dup = false;
  for (uint32_t i = 0; i + 1 < n && !dup; ++i) {
    const uint32_t ri = (uint32_t)(r.y0[s0 + i] >> 32);
    for (uint32_t k = i + 1; k < n; ++k)
      if ((uint32_t)(r.y0[s0 + k] >> 32) == ri) {
        dup = true;
        break;
      }
  }
  const bool big = dup ? r.dup_min && n >= r.dup_min : r.big_min && n >= r.big_min;
  r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (big ? F_BIG : 0));
  if (big) atomicAdd(&r.c->nbig_total, 1u);
  r.dirty[j] = 1;
  if (hist) atomicAdd(&hist[(dup ? 16 : 0) + min(n / 8, 15u)], 1u);
}

// ---- shimmer_to_overlap (shmr_overlap.c:52-180) for every dirty bucket in [lo, hi) -------------------------------------
// A group of GL lanes per bucket.  The rows (ai, descending) are sequential -- they communicate through the "contained"
// flags -- but the partners of one row are examined GL at a time, speculatively: lane l takes partner pbase + l, and the
// sequential semantics (stop once bestn overlaps are counted, or when the row's own read turns out contained) are then
// resolved with ballots.  What a lane beyond the stop did is harmless: a pair key, a reader registration (only ever costs a
// spurious re-evaluation), loads.  Buckets that hold a read twice can meet a pair twice within one evaluation: they run one
// partner at a time.  A single evaluation is a chain of dependent memory round trips, so the group form is what bounds
// the latency of a pass: rows x ~4 round trips instead of examinations x ~4.
constexpr uint32_t NCH = 256;  // reader-node arena chunk of a wavefront
constexpr uint32_t SPREAD = 256;
constexpr uint32_t ICH = 128;  // item arena piece of a wavefront slot (8 chunks of 16)

// every wavefront slot of k_eval starts with its own piece of the item arena (no atomic at all for its first ICH items); the
// shared counter starts behind the pieces
// (slots [0, n_dense): the dense rounds' wavefronts, GPW buckets each; [wlist0, nslots): the list-mode wavefronts; the slots
// between them are only used by the one-bucket-per-wavefront dense variant and start empty)
__global__ void k_init_slots(uint4 *__restrict__ wcur, uint32_t nslots, uint32_t n_dense, uint32_t wlist0, Counters *c) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w == 0) c->item_top = (n_dense + (nslots - wlist0)) * ICH;
  if (w >= nslots) return;
  if (w < n_dense || w >= wlist0) {
    const uint32_t piece = w < n_dense ? w : n_dense + (w - wlist0);
    wcur[w] = make_uint4(0u, 0u, piece * ICH, (piece + 1) * ICH);
  } else {
    wcur[w] = make_uint4(0u, 0u, 0u, 0u);
  }
}

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

// ---- shimmer_to_overlap (shmr_overlap.c:52-180) for every dirty bucket of the launch ------------------------------------
// A group of 16 lanes per bucket.  The rows (ai, descending) are sequential -- they communicate through the "contained"
// flags -- but the partners of one row are examined 16 at a time, speculatively: lane l takes partner pbase + l, and the
// sequential semantics (stop once bestn overlaps are counted, or when the row's own read turns out contained) are then
// resolved with ballots.  What a lane beyond the stop did is harmless: loads.  Buckets that hold a read twice can meet a
// pair twice within one evaluation: they run one partner at a time.  A single evaluation is a chain of dependent memory
// round trips, which is what bounds a sparse pass: the bucket's entries are staged in LDS once, a probe brings key and
// owner in one 16-byte load, and registrations / item stores are not waited for.
__global__ __launch_bounds__(256) void k_eval(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  __shared__ uint32_t s_rid[GPB][128], s_pos[GPB][128], s_rl[GPB][128];
  __shared__ uint8_t s_dir[GPB][128];
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1), gbase = lane & ~(GL - 1), gib = threadIdx.x / GL;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPW + (uint32_t)(lane / GL));
  const uint32_t j = (uint32_t)jj;
  bool alive = jj < hi && r.dirty[j] && !r.c->overflow;
  if (alive && (r.bflags[j] & F_BIG)) {   // a big bucket: left to k_eval_big, which runs behind this kernel from the list written here
    if (gl == 0) {
      const uint32_t at = atomicAdd(&r.c->nbig, 1u);
      if (at < LIST_CAP) r.blist[at] = j;   // (beyond the list: the bucket stays dirty and is listed again by the next pass)
    }
    alive = false;
  }
  {
    const uint64_t am = __ballot(alive);
    if (!am) return;
    if (lane == (int)__builtin_ctzll(am)) atomicAdd(&r.spread[(wave % SPREAD) * 8], (unsigned long long)(__popcll(am) / GL));
  }
  // reader-node arena: wave-uniform cursor; the wavefront that evaluates these buckets next time continues where this one stops
  const uint32_t wave_id = nlist ? r.wlist0 + wave : (uint32_t)(((uint64_t)lo + (uint64_t)wave * GPW) / GPW);
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;
  uint32_t icur = wc.z, iend = wc.w;   // item arena of this wavefront slot (multiples of 16), same idea
  uint32_t s0 = 0, n = 0;
  bool dup = false, first_eval = true;
  if (alive) {
    const uint32_t b = r.bid[j];
    s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    dup = (r.bflags[j] & F_DUP) != 0;
    first_eval = r.ever[j] == 0;
    for (uint32_t i = (uint32_t)gl; i < n; i += GL) {  // the bucket's entries -> LDS
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[gib][i] = rid, s_pos[gib][i] = (((uint32_t)y) >> 1) + 1, s_dir[gib][i] = r.dir[s0 + i], s_rl[gib][i] = r.rlen[rid];
    }
    if (gl == 0) {
      r.dirty[j] = 0;
      r.evaluated[j] = 1;
      r.ever[j] = 1;
      r.parity[j] ^= 1;
      r.ohead[j] = r.ihead[j];
    }
  }
  uint64_t clo = 0, chi = 0;  // "contained" flags of the bucket's entries (n <= 128)
  auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
  uint32_t head = NIL, num = 0, lookups = 0, skips = 0;
  uint32_t chunk = 0;  // base of the item chunk holding insertion ordinals [num & ~15, ...)
  bool any_guess = false, any_unfiled = false;
  int ai = (int)n - 1;  // (the first row opened is n - 2)
  bool row_open = false;
  uint32_t pbase = 0, got = 0, rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0;
  bool p_reg = false;  // this lane has a registration whose list position (p_idx) has not been looked at yet
  uint32_t p_idx = 0, p_slot = 0;
  auto resolve_pending = [&]() -> bool {  // false: the reader-node arena is exhausted
    if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
    const uint64_t rm = __ballot(p_reg);  // (what is left goes to the linked overflow)
    if (rm) {
      const uint32_t total = (uint32_t)__popcll(rm);
      if (rcur + total > rend) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if ((uint64_t)base + NCH > r.rn_cap) {
          atomicOr(&r.c->overflow, OV_NODES);
          return false;
        }
        rcur = base, rend = base + NCH;
      }
      if (p_reg) {
        const uint32_t node = rcur + lane_rank(rm);
        const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
        r.rn[node] = RNode{old, j};
      }
      rcur += total;
      p_reg = false;
    }
    return true;
  };
  for (;;) {
    if (alive && !row_open) {
      do --ai;
      while (ai >= 0 && cget((uint32_t)ai));
      if (ai < 0 || r.bestn == 0) {  // the bucket is done
        if (gl == 0) {
          r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
          r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
        }
        alive = false;
      } else {
        rid0 = s_rid[gib][ai], pos0 = s_pos[gib][ai], rlen0 = s_rl[gib][ai], dir0 = s_dir[gib][ai];
        got = 0, pbase = (uint32_t)ai + 1, row_open = true;
      }
    }
    if (!__ballot(alive)) {
      if (!resolve_pending()) return;
      if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
      break;
    }
    // ---- one batch of partners ----
    const uint32_t step = dup ? 1u : (uint32_t)GL;
    const uint32_t pi = pbase + (uint32_t)gl;
    bool valid = alive && (uint32_t)gl < step && pi < n && !cget(pi);
    uint32_t rid1 = 0, pos1 = 0;
    if (valid) {
      rid1 = s_rid[gib][pi], pos1 = s_pos[gib][pi];
      if (rid1 == rid0) valid = false;
    }
    uint32_t slot = NONE, v = 0;
    const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
    if (valid) slot = pair_find(r, pair, &v);
    const uint64_t vm = __ballot(valid);
    bool present = false, accepted = false, guessed = false;
    uint32_t ptype = 0, type = 0, mslot = NONE;
    if (valid) {
      present = v != 0 && own_bucket(v) < j;
      ptype = present ? own_type(v) : 0;
      if (!present && dup && slot != NONE)  // inserted earlier in THIS evaluation?
        for (uint32_t it = head; it != NIL; it = r.items[it - 1].next)
          if (r.items[it - 1].pslot == slot) {
            present = true, ptype = (r.items[it - 1].info >> 16) & 3;
            break;
          }
      if (!present) {
        const uint32_t rlen1 = s_rl[gib][pi], dir1 = s_dir[gib][pi];
        const uint32_t q_off = pos0 - pos1;
        if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
        uint32_t req = NONE;
        if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
        if (req < r.settled) {
          accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
        } else {
          accepted = true, guessed = true, type = T_OVERLAP;
          if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
        }
      }
    }
    if (!resolve_pending()) return;  // (the previous batch's registrations: their atomics have returned behind the loads above)
    // ---- the sequential semantics of the row over this batch, lowest partner first ----
    const uint64_t Vg = gbits(vm, gbase);
    const uint64_t P = gbits(__ballot(valid && present), gbase);
    const uint64_t PO = gbits(__ballot(valid && present && ptype == T_OVERLAP), gbase);
    const uint64_t A = gbits(__ballot(valid && !present && accepted), gbase);
    const uint64_t AO = gbits(__ballot(valid && !present && accepted && type == T_OVERLAP), gbase);
    const uint64_t AC = gbits(__ballot(valid && !present && accepted && type == T_CONTAINED), gbase);
    uint64_t AP = gbits(__ballot(valid && !present && accepted && type == T_CONTAINS), gbase);
    const uint64_t inc = PO | AO;
    int stop = GL;  // the last partner the sequential loop processes in this batch (GL: all of them, and the row goes on)
    if (alive && row_open) {
      const uint32_t need = r.bestn - got;  // >= 1
      if ((uint32_t)__popcll(inc) >= need) {
        uint64_t m = inc;
        for (uint32_t k = 1; k < need; ++k) m &= m - 1;
        stop = __builtin_ctzll(m);
      }
      if (AC) stop = min(stop, (int)__builtin_ctzll(AC));
    }
    const uint64_t proc = (2ULL << stop) - 1ULL;
    const uint64_t ins = A & proc;
    {  // the partners the sequential loop really examined register as readers of their pairs (the lists are only read by
       // k_update, after this kernel); a bucket listed by an earlier evaluation is not listed again.  The list position comes
       // from an atomic whose result is only looked at after the NEXT batch's loads have been issued (resolve_pending).
      bool reg = valid && ((proc >> gl) & 1);
      if (reg && slot == NONE) slot = pair_slot(r, pair);  // a pair the walk really examines gets its slot now
      if (reg && !first_eval) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(w);  // cnt, rhead, in[0], in[1]
        const uint32_t c = min(h1.x, NIN);
        if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
        for (uint32_t q = 2; q < c && reg; q += 8) {
          const uint4 a = *reinterpret_cast<const uint4 *>(w + 2 + q), b = *reinterpret_cast<const uint4 *>(w + 6 + q);
          const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k)
            if (q + k < c && x[k] == j + 1) reg = false;
        }
      }
      if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
    }
    // ---- the batch's insertions: the bucket's items fill 16-aligned chunks of 16 in insertion order (k_update walks a
    // bucket's list a chunk at a time, one lane per item) ----
    const bool my_ins = valid && !present && accepted && ((proc >> gl) & 1);
    const uint32_t cins = (uint32_t)__popcll(ins);
    const bool need_chunk = cins != 0 && (num == 0 || ((num + cins - 1) >> 4) != ((num - 1) >> 4));  // group-uniform
    uint32_t fresh = 0;
    {
      const uint64_t cm = __ballot(need_chunk && gl == 0);
      if (cm) {
        const uint32_t want = 16u * (uint32_t)__popcll(cm);
        if (icur + want > iend) {  // refill: one atomic on the shared counter per ICH items instead of one per chunk
          const uint32_t take = want > ICH ? want : ICH;
          uint32_t base = 0;
          if (lane == (int)__builtin_ctzll(cm)) base = atomicAdd(&r.c->item_top, take);
          base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(cm), 64);
          if ((uint64_t)base + take > r.item_cap) {
            atomicOr(&r.c->overflow, OV_ITEMS);
            return;
          }
          icur = base, iend = base + take;   // (what was left of the old piece, < want, is not used)
        }
        fresh = icur + 16u * (uint32_t)__popcll(cm & ((1ULL << gbase) - 1ULL));  // (gl == 0 lanes: one bit per group)
        icur += want;
      }
    }
    if (cins) {
      if (my_ins) {
        const uint32_t ord = num + (uint32_t)__popcll(ins & ((1ULL << gl) - 1ULL));  // insertion ordinal within the bucket
        const bool in_fresh = need_chunk && (num == 0 || (ord >> 4) != ((num - 1) >> 4));
        const uint32_t idx = (in_fresh ? fresh : chunk) + (ord & 15);
        uint32_t next;
        if (ord == 0) next = NIL;
        else if ((ord & 15) == 0) next = chunk + 16;  // the last item of the previous chunk, + 1
        else next = idx;                              // the item before this one, + 1
        uint32_t info = (uint32_t)ai | pi << 8 | type << 16;
        if (guessed) info |= I_GUESS;
        if (mslot == NONE) info |= I_UNFILED;
        r.items[idx] = Item{slot, info, mslot, next};
      }
      const uint32_t last = num + cins - 1;
      if (need_chunk && (num == 0 || (last >> 4) != ((num - 1) >> 4))) chunk = fresh;
      head = chunk + (last & 15) + 1;
      num += cins;
    }
    {
      const uint64_t g1 = gbits(__ballot(my_ins && guessed), gbase), g2 = gbits(__ballot(my_ins && mslot == NONE), gbase);
      any_guess |= g1 != 0, any_unfiled |= g2 != 0;
    }
    if (alive && row_open) {
      got += (uint32_t)__popcll(inc & proc);
      skips += (uint32_t)__popcll(P & proc);
      lookups += (uint32_t)__popcll(Vg & ~P & proc);
      AP &= proc;
      if (AP) {  // partners found contained: entry pbase + l
        if (pbase < 64) {
          clo |= AP << pbase;
          if (pbase) chi |= AP >> (64 - pbase);
        } else {
          chi |= AP << (pbase - 64);
        }
      }
      if (AC & proc) {
        if (ai < 64) clo |= 1ULL << ai;
        else chi |= 1ULL << (ai - 64);
      }
      if (stop < GL || pbase + step >= n) row_open = false;
      else pbase += step;
    }
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


// ---- the same evaluation with FOUR ROWS PER STEP, for the sparse passes (a wavefront per bucket: lanes are plentiful there and
// a pass lasts as long as its longest bucket's chain of rows).  Lane l works row l / PW, partner l % PW; the rows are committed
// in order while each one is complete within its PW partners and no earlier row of the step set a contained flag (then the
// rows below are looked at again with the new flags); a row that needs more partners is continued alone, GLT per step.
template <int GLT, int PW>
__global__ __launch_bounds__(256) void k_eval_rows(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  constexpr uint32_t GPWT = 64 / GLT, GPBT = 256 / GLT;
  constexpr int SH = PW == 16 ? 4 : 2;  // log2(PW)
  constexpr uint64_t RM = (1ULL << PW) - 1ULL;  // one row's lanes
  __shared__ uint32_t s_rid[GPBT][128], s_pos[GPBT][128], s_rl[GPBT][128];
  __shared__ uint8_t s_dir[GPBT][128];
  const int lane = threadIdx.x & 63, gl = lane & (GLT - 1), gbase = lane & ~(GLT - 1), gib = threadIdx.x / GLT;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPWT + (uint32_t)(lane / GLT));
  const uint32_t j = (uint32_t)jj;
  bool alive = jj < hi && r.dirty[j] && !r.c->overflow;
  if (alive && (r.bflags[j] & F_BIG)) {   // a big bucket: left to k_eval_big, which runs behind this kernel from the list written here
    if (gl == 0) {
      const uint32_t at = atomicAdd(&r.c->nbig, 1u);
      if (at < LIST_CAP) r.blist[at] = j;   // (beyond the list: the bucket stays dirty and is listed again by the next pass)
    }
    alive = false;
  }
  {
    const uint64_t am = __ballot(alive);
    if (!am) return;
    if (lane == (int)__builtin_ctzll(am)) atomicAdd(&r.spread[(wave % SPREAD) * 8], (unsigned long long)(__popcll(am) / GLT));
  }
  // reader-node arena: wave-uniform cursor; the wavefront that evaluates these buckets next time continues where this one stops
  const uint32_t wave_id = nlist ? r.wlist0 + wave : (uint32_t)(((uint64_t)lo + (uint64_t)wave * GPWT) / GPWT);
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;
  uint32_t icur = wc.z, iend = wc.w;   // item arena of this wavefront slot (multiples of 16), same idea
  uint32_t s0 = 0, n = 0;
  bool dup = false, first_eval = true;
  if (alive) {
    const uint32_t b = r.bid[j];
    s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    dup = (r.bflags[j] & F_DUP) != 0;
    first_eval = r.ever[j] == 0;
    for (uint32_t i = (uint32_t)gl; i < n; i += GLT) {  // the bucket's entries -> LDS
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[gib][i] = rid, s_pos[gib][i] = (((uint32_t)y) >> 1) + 1, s_dir[gib][i] = r.dir[s0 + i], s_rl[gib][i] = r.rlen[rid];
    }
    if (gl == 0) {
      r.dirty[j] = 0;
      r.evaluated[j] = 1;
      r.ever[j] = 1;
      r.parity[j] ^= 1;
      r.ohead[j] = r.ihead[j];
    }
  }
  auto gbits = [&](uint64_t wave_mask, int) { return GLT == 64 ? wave_mask : ((wave_mask >> gbase) & ((1ULL << (GLT & 63)) - 1ULL)); };
  uint64_t clo = 0, chi = 0;  // "contained" flags of the bucket's entries (n <= 128)
  auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
  auto cset = [&](uint32_t i) {
    if (i < 64) clo |= 1ULL << i;
    else chi |= 1ULL << (i - 64);
  };
  uint32_t head = NIL, num = 0, lookups = 0, skips = 0;
  uint32_t chunk = 0;  // base of the item chunk holding insertion ordinals [num & ~15, ...)
  bool any_guess = false, any_unfiled = false;
  int done_to = (int)n - 1;  // rows >= done_to are finished (the first row is n - 2)
  bool row_open = false;     // a single row (cur_row) is in progress, sixteen partners per step from pbase
  int cur_row = 0, a0 = -1, a1 = -1, a2 = -1, a3 = -1, nrows = 0;
  uint32_t pbase = 0, got = 0;
  bool p_reg = false;  // this lane has a registration whose list position (p_idx) has not been looked at yet
  uint32_t p_idx = 0, p_slot = 0;
  auto resolve_pending = [&]() -> bool {  // false: the reader-node arena is exhausted
    if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
    const uint64_t rm = __ballot(p_reg);  // (what is left goes to the linked overflow)
    if (rm) {
      const uint32_t total = (uint32_t)__popcll(rm);
      if (rcur + total > rend) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if ((uint64_t)base + NCH > r.rn_cap) {
          atomicOr(&r.c->overflow, OV_NODES);
          return false;
        }
        rcur = base, rend = base + NCH;
      }
      if (p_reg) {
        const uint32_t node = rcur + lane_rank(rm);
        const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
        r.rn[node] = RNode{old, j};
      }
      rcur += total;
      p_reg = false;
    }
    return true;
  };
  for (;;) {
    if (alive && !row_open) {  // the next (up to four) rows that are not contained
      nrows = 0, a0 = a1 = a2 = a3 = -1;
      int x = done_to;
      while (nrows < 4) {
        do --x;
        while (x >= 0 && cget((uint32_t)x));
        if (x < 0) break;
        if (nrows == 0) a0 = x;
        else if (nrows == 1) a1 = x;
        else if (nrows == 2) a2 = x;
        else a3 = x;
        ++nrows;
      }
      if (nrows == 0 || r.bestn == 0) {  // the bucket is done
        if (gl == 0) {
          r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
          r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
        }
        alive = false;
      } else if (dup) {
        cur_row = a0, got = 0, pbase = (uint32_t)a0 + 1, row_open = true;
      }
    }
    if (!__ballot(alive)) {
      if (!resolve_pending()) return;
      if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
      break;
    }
    // ---- this step's (row, partner) of the lane ----
    const bool single = row_open;  // group-uniform
    const uint32_t step = dup ? 1u : (uint32_t)GLT;
    const int q = gl >> SH;
    const int myrow = single ? cur_row : (q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3);
    const uint32_t pi = single ? pbase + (uint32_t)gl : (uint32_t)(myrow + 1 + (gl & (PW - 1)));
    bool valid = alive && (single ? (uint32_t)gl < step : q < nrows) && pi < n && !cget(pi);
    uint32_t rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0, rid1 = 0, pos1 = 0;
    if (valid) {
      rid0 = s_rid[gib][myrow], pos0 = s_pos[gib][myrow], rlen0 = s_rl[gib][myrow], dir0 = s_dir[gib][myrow];
      rid1 = s_rid[gib][pi], pos1 = s_pos[gib][pi];
      if (rid1 == rid0) valid = false;
    }
    uint32_t slot = NONE, v = 0;
    const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
    if (valid) slot = pair_find(r, pair, &v);
    const uint64_t vm = __ballot(valid);
    bool present = false, accepted = false, guessed = false;
    uint32_t ptype = 0, type = 0, mslot = NONE;
    if (valid) {
      present = v != 0 && own_bucket(v) < j;
      ptype = present ? own_type(v) : 0;
      if (!present && dup && slot != NONE)  // inserted earlier in THIS evaluation?
        for (uint32_t it = head; it != NIL; it = r.items[it - 1].next)
          if (r.items[it - 1].pslot == slot) {
            present = true, ptype = (r.items[it - 1].info >> 16) & 3;
            break;
          }
      if (!present) {
        const uint32_t rlen1 = s_rl[gib][pi], dir1 = s_dir[gib][pi];
        const uint32_t q_off = pos0 - pos1;
        if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
        uint32_t req = NONE;
        if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
        if (req < r.settled) {
          accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
        } else {
          accepted = true, guessed = true, type = T_OVERLAP;
          if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
        }
      }
    }
    if (!resolve_pending()) return;  // (the previous step's registrations: their atomics have returned behind the loads above)
    // ---- the sequential semantics over this step, lowest lane first ----
    const uint64_t Vg = gbits(vm, gbase);
    const uint64_t P = gbits(__ballot(valid && present), gbase);
    const uint64_t PO = gbits(__ballot(valid && present && ptype == T_OVERLAP), gbase);
    const uint64_t A = gbits(__ballot(valid && !present && accepted), gbase);
    const uint64_t AO = gbits(__ballot(valid && !present && accepted && type == T_OVERLAP), gbase);
    const uint64_t AC = gbits(__ballot(valid && !present && accepted && type == T_CONTAINED), gbase);
    const uint64_t AP = gbits(__ballot(valid && !present && accepted && type == T_CONTAINS), gbase);
    const uint64_t inc = PO | AO;
    uint64_t proc = 0;  // the lanes the sequential walk really visits in this step
    if (alive && single) {
      int stop = GLT;  // the last partner the row processes in this step (GLT: all of them, and the row goes on)
      const uint32_t need = r.bestn - got;  // >= 1
      if ((uint32_t)__popcll(inc) >= need) {
        uint64_t m = inc;
        for (uint32_t k = 1; k < need; ++k) m &= m - 1;
        stop = __builtin_ctzll(m);
      }
      if (AC) stop = min(stop, (int)__builtin_ctzll(AC));
      proc = stop >= 63 ? ~0ULL : ((2ULL << stop) - 1ULL);
      got += (uint32_t)__popcll(inc & proc);
      for (uint64_t m = AP & proc; m; m &= m - 1) cset(pbase + (uint32_t)__builtin_ctzll(m));  // partners found contained
      if (AC & proc) cset((uint32_t)cur_row);
      if (stop < GLT || pbase + step >= n) row_open = false, done_to = cur_row;
      else pbase += step;
    } else if (alive) {
      int committed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k >= nrows || committed != k) continue;
        const int row = k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
        // partners an EARLIER row of this step found contained (or that were such a row) are not examined by this row: their lanes are
        // dropped from its masks here (round 3 ended the step at the first row that changed a flag; a row's partners lie above it, so
        // the rows themselves are never flagged by an earlier row of the step)
        const uint32_t gone = (uint32_t)shr128(M128{clo, chi}, (uint32_t)row + 1).lo & (uint32_t)RM;
        const uint32_t inc_k = (uint32_t)((inc >> (PW * k)) & RM) & ~gone, ac_k = (uint32_t)((AC >> (PW * k)) & RM) & ~gone,
                       ap_k = (uint32_t)((AP >> (PW * k)) & RM) & ~gone;
        int stop = PW;
        if ((uint32_t)__popc(inc_k) >= r.bestn) {
          uint32_t m = inc_k;
          for (uint32_t t = 1; t < r.bestn; ++t) m &= m - 1;
          stop = __builtin_ctz(m);
        }
        if (ac_k) stop = min(stop, (int)__builtin_ctz(ac_k));
        if (stop == PW && (uint32_t)row + 1 + PW < n) continue;  // the row needs more partners: it is continued alone (committed stays k)
        const uint32_t proc_k = (stop < PW ? (2u << stop) - 1u : (uint32_t)RM) & ~gone;
        proc |= (uint64_t)proc_k << (PW * k);
        ++committed;
        for (uint32_t m = ap_k & proc_k; m; m &= m - 1) cset((uint32_t)row + 1 + (uint32_t)__builtin_ctz(m));   // partners found contained
        if (ac_k & proc_k) cset((uint32_t)row);
      }
      if (committed == 0) cur_row = a0, got = 0, pbase = (uint32_t)a0 + 1, row_open = true;  // (nothing done in this step)
      else done_to = committed == 1 ? a0 : committed == 2 ? a1 : committed == 3 ? a2 : a3;
    }
    skips += (uint32_t)__popcll(P & proc);
    lookups += (uint32_t)__popcll(Vg & ~P & proc);
    const uint64_t ins = A & proc;
    {  // the partners the walk really examined register as readers of their pairs (the lists are only read by k_update, after
       // this kernel); a bucket listed by an earlier evaluation is not listed again.  The list position comes from an atomic
       // whose result is only looked at after the NEXT step's loads have been issued (resolve_pending).
      bool reg = valid && ((proc >> gl) & 1);
      if (reg && slot == NONE) slot = pair_slot(r, pair);  // a pair the walk really examines gets its slot now
      if (reg && !first_eval) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(w);  // cnt, rhead, in[0], in[1]
        const uint32_t c = min(h1.x, NIN);
        if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
        for (uint32_t qq = 2; qq < c && reg; qq += 8) {
          const uint4 a = *reinterpret_cast<const uint4 *>(w + 2 + qq), b = *reinterpret_cast<const uint4 *>(w + 6 + qq);
          const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k)
            if (qq + k < c && x[k] == j + 1) reg = false;
        }
      }
      if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
    }
    // ---- the step's insertions: the bucket's items fill 16-aligned chunks of 16 in insertion order (lane order is the
    // sequential order; k_update walks a bucket's list a chunk at a time, one lane per item) ----
    const bool my_ins = valid && !present && accepted && ((proc >> gl) & 1);
    const uint32_t cins = (uint32_t)__popcll(ins);  // up to 64 in one step here: it may open several 16-item chunks
    static_assert(GLT == 64, "one bucket per wavefront: the chunk allocation below is wave-uniform");
    if (cins) {
      const uint32_t cur_no = num ? (num - 1) >> 4 : 0;                 // number of the chunk `chunk` (meaningless while num == 0)
      const uint32_t first_new = num ? cur_no + 1 : 0;                   // number of the first chunk this step has to open
      const uint32_t last = num + cins - 1, last_no = last >> 4;
      const uint32_t nnew = last_no + 1 > first_new ? last_no + 1 - first_new : 0;
      uint32_t fresh = 0;
      if (nnew) {
        const uint32_t want = 16u * nnew;
        if (icur + want > iend) {
          const uint32_t take = want > ICH ? want : ICH;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&r.c->item_top, take);
          base = (uint32_t)__shfl((int)base, 0, 64);
          if ((uint64_t)base + take > r.item_cap) {
            atomicOr(&r.c->overflow, OV_ITEMS);
            return;
          }
          icur = base, iend = base + take;
        }
        fresh = icur, icur += want;
      }
      auto base_of = [&](uint32_t no) { return no >= first_new ? fresh + 16u * (no - first_new) : chunk; };
      if (my_ins) {
        const uint32_t ord = num + (uint32_t)__popcll(ins & ((1ULL << gl) - 1ULL));  // insertion ordinal within the bucket
        const uint32_t idx = base_of(ord >> 4) + (ord & 15);
        uint32_t next;
        if (ord == 0) next = NIL;
        else if ((ord & 15) == 0) next = base_of((ord >> 4) - 1) + 16;  // the last item of the previous chunk, + 1
        else next = idx;                                                // the item before this one, + 1
        uint32_t info = (uint32_t)myrow | pi << 8 | type << 16;
        if (guessed) info |= I_GUESS;
        if (mslot == NONE) info |= I_UNFILED;
        r.items[idx] = Item{slot, info, mslot, next};
      }
      chunk = base_of(last_no);
      head = chunk + (last & 15) + 1;
      num += cins;
    }
    {
      const uint64_t g1 = gbits(__ballot(my_ins && guessed), gbase), g2 = gbits(__ballot(my_ins && mslot == NONE), gbase);
      any_guess |= g1 != 0, any_unfiled |= g2 != 0;
    }
  }
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
__global__ __launch_bounds__(64 * BIG_NW) void k_eval_big(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  enum { MV = 0, MP, MPO, MA, MAO, MAC, MAP, MGU, MUF, MDF, NM };
  __shared__ uint32_t s_rid[128], s_pos[128], s_rl[128];
  __shared__ uint8_t s_dir[128];
  __shared__ uint64_t s_m[BIG_NW][NM];
  __shared__ uint32_t s_fresh, s_abort, s_bail;
  __shared__ unsigned long long s_setk[SET_CAP];      // pairs inserted by this evaluation (key + 1; 0: empty) ...
  __shared__ uint8_t s_sett[SET_CAP];                 // ... and their types
  __shared__ unsigned long long s_clk[CLAIM_CAP];     // this step's claims: pair (key + 1) ...
  __shared__ uint32_t s_clw[CLAIM_CAP];               // ... and the lowest walk index claiming it
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t wave_id = r.wbig0 + blockIdx.x * BIG_NW + (uint32_t)w;
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;   // reader-node arena of this wavefront
  uint32_t icur = wc.z, iend = wc.w;   // item arena of the workgroup (only thread 0's copy is used)
  if (threadIdx.x == 0) s_abort = 0;
  // its buckets: the big list of the pass.  Two kinds of entries: a bucket id, noted by a narrow kernel that ran over a RANGE and met the
  // bucket (k_eval / k_eval_rows skip big buckets); and a position in the dirty list | 2^31, noted by the count that made the list
  // (k_count_b) -- those count only in a list-mode launch, and only below `lo` = the number of list entries the narrow kernel and the
  // k_update of this pass cover (an evaluation k_update does not see would be lost)
  const uint32_t list_limit = nlist ? lo : 0u;
  const uint32_t nbig = min(r.c->nbig, LIST_CAP);
  for (uint32_t g = blockIdx.x; g < nbig; g += gridDim.x) {
  {
    const uint32_t e = r.blist[g];
    if ((e & 0x80000000u) && (e & 0x7FFFFFFFu) >= list_limit) continue;   // (workgroup-uniform)
    const uint32_t j = (e & 0x80000000u) ? (r.dlist[e & 0x7FFFFFFFu] & 0x7FFFFFFFu) : e;
    __syncthreads();   // (the previous bucket's LDS is done with)
    if (threadIdx.x == 0) {   // one thread decides for the workgroup (a flag another workgroup raises meanwhile must not split it)
      if (r.c->overflow) s_abort = 1;
      s_bail = 0, s_fresh = (j < hi && r.dirty[j]) ? 1u : 0u;   // (s_fresh doubles as "go": a listed bucket is dirty unless the list is stale)
    }
    __syncthreads();
    if (s_abort) break;
    if (!s_fresh) continue;
    const uint32_t b = r.bid[j], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    const bool first_eval = r.ever[j] == 0;
    const bool dup = (r.bflags[j] & F_DUP) != 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[i] = rid, s_pos[i] = (((uint32_t)y) >> 1) + 1, s_dir[i] = r.dir[s0 + i], s_rl[i] = r.rlen[rid];
    }
    if (dup) {
      for (uint32_t i = threadIdx.x; i < SET_CAP; i += blockDim.x) s_setk[i] = 0;
      for (uint32_t i = threadIdx.x; i < CLAIM_CAP; i += blockDim.x) s_clk[i] = 0, s_clw[i] = 0xFFFFFFFFu;
    }
    __syncthreads();   // (entries staged; everybody has read ever[] / dirty[] / bflags[] before thread 0 changes them)
    if (threadIdx.x == 0) {
      r.dirty[j] = 0, r.evaluated[j] = 1, r.ever[j] = 1, r.parity[j] ^= 1, r.ohead[j] = r.ihead[j];
      atomicAdd(&r.spread[(blockIdx.x % SPREAD) * 8], 1ULL);
    }
    // ---- workgroup-uniform state (every thread holds a copy and updates it identically) ----
    uint64_t clo = 0, chi = 0;   // "contained" flags of the bucket's entries
    auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
    auto cset = [&](uint32_t i) {
      if (i < 64) clo |= 1ULL << i;
      else chi |= 1ULL << (i - 64);
    };
    uint32_t head = NIL, num = 0, lookups = 0, skips = 0, chunk = 0;
    bool any_guess = false, any_unfiled = false;
    int done_to = (int)n - 1;   // rows >= done_to are finished (the first row is n - 2)
    bool row_open = false;       // one row (cur_row) is being continued alone, from partner pbase, with `got` overlaps counted so far
    int cur_row = 0;
    uint32_t pbase = 0, got = 0;
    bool p_reg = false;          // this lane has a registration whose list position (p_idx) has not been looked at yet
    uint32_t p_idx = 0, p_slot = 0;
    auto resolve_pending = [&]() {   // as in k_eval_rows; an exhausted arena raises s_abort instead of returning
      if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
      const uint64_t rm = __ballot(p_reg);
      if (rm) {
        const uint32_t total = (uint32_t)__popcll(rm);
        if (rcur + total > rend) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
          base = (uint32_t)__shfl((int)base, 0, 64);
          if ((uint64_t)base + NCH > r.rn_cap) {
            atomicOr(&r.c->overflow, OV_NODES);
            s_abort = 1;
            p_reg = false;
            return;
          }
          rcur = base, rend = base + NCH;
        }
        if (p_reg) {
          const uint32_t node = rcur + lane_rank(rm);
          const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
          r.rn[node] = RNode{old, j};
        }
        rcur += total;
        p_reg = false;
      }
    };
#ifdef PGX_BIG_STATS
    uint32_t st_steps = 0, st_rows = 0, st_cut = 0, st_cont = 0, st_full = 0;
#endif
    for (;;) {
      // the rows of this step: the open row alone, or the next (up to four) rows that are not contained
      int a[BIG_NR], nrows = 0;
#pragma unroll
      for (int k = 0; k < BIG_NR; ++k) a[k] = -1;
      if (row_open) {
        a[0] = cur_row, nrows = 1;
      } else {
        for (int x = done_to; nrows < BIG_NR;) {
          do --x;
          while (x >= 0 && cget((uint32_t)x));
          if (x < 0) break;
          a[nrows++] = x;
        }
      }
      if (nrows == 0 || r.bestn == 0) break;
      // ---- this step's (row, partner) of the lane: slot q = wave / 2 takes row a[q], partners first + (wave % 2) * 64 + lane ----
      const int q = w >> 1, off = (w & 1) * 64 + lane;   // off: partner offset within the row's 128 (= its bit in the row's masks)
      const int myrow = a[q];
      const uint32_t first = row_open ? pbase : (uint32_t)(myrow + 1);
      const uint32_t pi = first + (uint32_t)off;
      const uint32_t wi = (uint32_t)(q * 128 + off);     // position in walk order
      bool valid = q < nrows && pi < n && !cget(pi);
      uint32_t rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0, rid1 = 0, pos1 = 0;
      if (valid) {
        rid0 = s_rid[myrow], pos0 = s_pos[myrow], rlen0 = s_rl[myrow], dir0 = s_dir[myrow];
        rid1 = s_rid[pi], pos1 = s_pos[pi];
        if (rid1 == rid0) valid = false;
      }
      uint32_t slot = NONE, v = 0;
      const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
      if (valid) slot = pair_find(r, pair, &v);
      bool present = false, accepted = false, guessed = false;
      uint32_t ptype = 0, type = 0, mslot = NONE;
      if (valid) {
        present = v != 0 && own_bucket(v) < j;
        ptype = present ? own_type(v) : 0;
        if (!present && dup) {   // inserted earlier in THIS evaluation?
          for (uint32_t i = (uint32_t)mix64(pair) & (SET_CAP - 1);; i = (i + 1) & (SET_CAP - 1)) {
            const unsigned long long kk = s_setk[i];
            if (kk == 0) break;
            if (kk == pair + 1) {
              present = true, ptype = s_sett[i];
              break;
            }
          }
        }
        if (!present) {
          const uint32_t rlen1 = s_rl[pi], dir1 = s_dir[pi];
          const uint32_t q_off = pos0 - pos1;
          if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
          uint32_t req = NONE;
          if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
          if (req < r.settled) {
            accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
          } else {
            accepted = true, guessed = true, type = T_OVERLAP;
            if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
          }
        }
      }
      resolve_pending();   // (the previous step's registrations: their atomics have returned behind the loads above)
      const bool ins0 = valid && !present && accepted;
      uint32_t my_claim = NONE;
      if (dup) {   // would-be inserters claim their pair: the lowest walk index wins
        if (ins0) {
          for (uint32_t i = (uint32_t)(mix64(pair) >> 20) & (CLAIM_CAP - 1);; i = (i + 1) & (CLAIM_CAP - 1)) {
            unsigned long long kk = s_clk[i];
            if (kk == 0) kk = atomicCAS(&s_clk[i], 0ULL, (unsigned long long)pair + 1), kk = kk ? kk : pair + 1;
            if (kk == pair + 1) {
              atomicMin(&s_clw[i], wi);
              my_claim = i;
              break;
            }
          }
        }
      }
      {
        const uint64_t mv = __ballot(valid), mp = __ballot(valid && present), mpo = __ballot(valid && present && ptype == T_OVERLAP);
        const uint64_t ma = __ballot(ins0), mao = __ballot(ins0 && type == T_OVERLAP), mac = __ballot(ins0 && type == T_CONTAINED);
        const uint64_t map = __ballot(ins0 && type == T_CONTAINS), mgu = __ballot(ins0 && guessed), muf = __ballot(ins0 && mslot == NONE);
        if (lane == 0)
          s_m[w][MV] = mv, s_m[w][MP] = mp, s_m[w][MPO] = mpo, s_m[w][MA] = ma, s_m[w][MAO] = mao, s_m[w][MAC] = mac, s_m[w][MAP] = map,
          s_m[w][MGU] = mgu, s_m[w][MUF] = muf, s_m[w][MDF] = 0;
      }
      __syncthreads();
      if (dup) {   // a lane whose pair a LOWER lane of this step would insert is FLAGGED: its row is cut in front of it (below).  That
                   // holds for EVERY lane that found the pair absent, also one whose own alignment is rejected: sequentially it would
                   // have found the pair seen and skipped it.
        bool dflag = false;
        if (my_claim != NONE) {
          dflag = s_clw[my_claim] < wi;
        } else if (valid && !present) {
          for (uint32_t i = (uint32_t)(mix64(pair) >> 20) & (CLAIM_CAP - 1);; i = (i + 1) & (CLAIM_CAP - 1)) {
            const unsigned long long kk = s_clk[i];
            if (kk == 0) break;
            if (kk == pair + 1) {
              dflag = s_clw[i] < wi;
              break;
            }
          }
        }
        const uint64_t mdf = __ballot(dflag);
        if (lane == 0) s_m[w][MDF] = mdf;
        __syncthreads();
      }
      if (s_abort) break;
      // ---- the sequential semantics over this step: the rows in order, each over its 128 partners, lowest first (uniform) ----
      M128 proc[BIG_NR];
#pragma unroll
      for (int k = 0; k < BIG_NR; ++k) proc[k] = M128{0, 0};
      int committed = 0;         // rows completed in this step
      bool open_next = false;    // the row after them was cut: it is continued alone
      int open_row = 0;
      uint32_t open_pbase = 0, open_got = 0;
      for (int k = 0; k < nrows; ++k) {
        // A partner that an EARLIER row of this step found contained (or that was such a row) is not examined by this row: its lane is dropped
        // from the row's masks right here.  (Round 3 ended the step at the first row that changed a flag and looked at the rows below again
        // in the next one.  Rows themselves are never flagged by an earlier row of the step: a row's partners lie above it.)
        const uint32_t first_k = row_open ? pbase : (uint32_t)(a[k] + 1);
        const M128 gone = shr128(M128{clo, chi}, first_k);
        const M128 inc = andn128(M128{s_m[2 * k][MPO] | s_m[2 * k][MAO], s_m[2 * k + 1][MPO] | s_m[2 * k + 1][MAO]}, gone);
        const M128 ac = andn128(M128{s_m[2 * k][MAC], s_m[2 * k + 1][MAC]}, gone), ap = andn128(M128{s_m[2 * k][MAP], s_m[2 * k + 1][MAP]}, gone);
        // the row's cut: its first flagged lane.  (Round 3 took ONE cut for the step, the lowest flagged lane of all rows -- which most often lay
        // beyond the stop of its row, among lanes the walk never visits, and still ended the step there: 69 % of all steps ended with rows left,
        // 1.5 of 4 rows committed per step, profiles/r04w_big_stats_c4s.txt.  A flag whose lower claimant turns out unvisited is void but harmless:
        // the lane is looked at again in the next step.)
        const M128 df = andn128(M128{s_m[2 * k][MDF], s_m[2 * k + 1][MDF]}, gone);
        const int cut = any128(df) ? ctz128(df) : 128;   // first offset of the row that may not be processed
        if (cut == 0) break;                             // the cut lies in front of this row
        const uint32_t need = r.bestn - (row_open ? got : 0u);   // >= 1
        int stop = 128;
        if ((uint32_t)popc128(inc) >= need) stop = nth128(inc, need);
        if (any128(ac)) stop = min(stop, ctz128(ac));
        const bool complete = stop < cut || (cut == 128);   // the row ends before the cut (or there is none in it)
        proc[k] = andn128(complete ? upto128(stop) : upto128(cut - 1), gone);
        const M128 apk = and128(ap, proc[k]);
        for (uint64_t m = apk.lo; m; m &= m - 1) cset(first_k + (uint32_t)__builtin_ctzll(m));   // partners found contained
        for (uint64_t m = apk.hi; m; m &= m - 1) cset(first_k + 64u + (uint32_t)__builtin_ctzll(m));
        const bool rowc = any128(and128(ac, proc[k]));
        if (rowc) cset((uint32_t)a[k]);
        if (!complete) {
          open_next = true, open_row = a[k], open_pbase = first_k + (uint32_t)cut, open_got = (row_open ? got : 0u) + (uint32_t)popc128(and128(inc, proc[k]));
          break;
        }
        ++committed;
      }
#ifdef PGX_BIG_STATS
      ++st_steps, st_rows += (uint32_t)committed, st_cut += open_next ? 1u : 0u, st_full += (committed == nrows) ? 1u : 0u;
      st_cont += (!open_next && committed < nrows) ? 1u : 0u;
#endif
      if (committed) done_to = a[committed - 1];
      row_open = open_next;
      if (open_next) cur_row = open_row, pbase = open_pbase, got = open_got;
      // per wavefront: what the walk really visits, and the insertions in walk order (row, then partner)
      uint32_t cins = 0, before = 0;
      uint64_t myproc = 0;
      for (int ww = 0; ww < BIG_NW; ++ww) {
        const int k = ww >> 1;
        const uint64_t pw = (ww & 1) ? proc[k].hi : proc[k].lo;
        const uint32_t c = (uint32_t)__popcll(s_m[ww][MA] & pw);
        if (ww < w) before += c;
        if (ww == w) myproc = pw;
        cins += c;
        skips += (uint32_t)__popcll(s_m[ww][MP] & pw);
        lookups += (uint32_t)__popcll(s_m[ww][MV] & ~s_m[ww][MP] & pw);
        any_guess |= (s_m[ww][MGU] & pw) != 0, any_unfiled |= (s_m[ww][MUF] & pw) != 0;
      }
      // the item chunks this step opens (one allocation for the workgroup, by thread 0)
      const uint32_t cur_no = num ? (num - 1) >> 4 : 0, first_new = num ? cur_no + 1 : 0;
      const uint32_t last = num + cins - 1, last_no = last >> 4;
      const uint32_t nnew = cins && last_no + 1 > first_new ? last_no + 1 - first_new : 0;
      if (threadIdx.x == 0 && nnew) {
        const uint32_t want = 16u * nnew;
        if (icur + want > iend) {
          const uint32_t take = want > ICH ? want : ICH;
          const uint32_t base = atomicAdd(&r.c->item_top, take);
          if ((uint64_t)base + take > r.item_cap) atomicOr(&r.c->overflow, OV_ITEMS), s_abort = 1;
          icur = base, iend = base + take;
        }
        s_fresh = icur, icur += want;
      }
      const bool visited = valid && ((myproc >> lane) & 1);
      {  // the partners the walk really examined register as readers of their pairs (as in k_eval_rows)
        bool reg = visited;
        if (reg && slot == NONE) slot = pair_slot(r, pair);
        if (reg && !first_eval) {
          const uint32_t *pw = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
          const uint4 h1 = *reinterpret_cast<const uint4 *>(pw);  // cnt, rhead, in[0], in[1]
          const uint32_t c = min(h1.x, NIN);
          if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
          for (uint32_t qq = 2; qq < c && reg; qq += 8) {
            const uint4 xa = *reinterpret_cast<const uint4 *>(pw + 2 + qq), xb = *reinterpret_cast<const uint4 *>(pw + 6 + qq);
            const uint32_t x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
              if (qq + k < c && x[k] == j + 1) reg = false;
          }
        }
        if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
      }
      const bool my_ins = ins0 && visited;
      if (dup) {
        if (my_claim != NONE) s_clk[my_claim] = 0, s_clw[my_claim] = 0xFFFFFFFFu;   // (the claim table is empty again for the next step)
        if (my_ins) {   // this evaluation's insertions, for the probes of the steps to come
          uint32_t tries = 0;
          for (uint32_t i = (uint32_t)mix64(pair) & (SET_CAP - 1);; i = (i + 1) & (SET_CAP - 1)) {
            const unsigned long long kk = atomicCAS(&s_setk[i], 0ULL, (unsigned long long)pair + 1);
            if (kk == 0 || kk == pair + 1) {
              s_sett[i] = (uint8_t)type;
              break;
            }
            if (++tries >= SET_CAP) {   // the set is full (not with <= 128 entries and bestn <= ~8; kept as a guard): leave the bucket to k_eval_rows
              s_bail = 1;
              break;
            }
          }
        }
      }
      __syncthreads();   // (s_fresh is there; nobody reads this step's masks any more; the set holds this step's insertions)
      if (s_abort) break;
      if (cins) {
        const uint32_t fresh = s_fresh;
        auto base_of = [&](uint32_t no) { return no >= first_new ? fresh + 16u * (no - first_new) : chunk; };
        if (my_ins) {
          const uint32_t ord = num + before + (uint32_t)__popcll(s_m[w][MA] & myproc & ((1ULL << lane) - 1ULL));   // insertion ordinal within the bucket
          const uint32_t idx = base_of(ord >> 4) + (ord & 15);
          uint32_t next;
          if (ord == 0) next = NIL;
          else if ((ord & 15) == 0) next = base_of((ord >> 4) - 1) + 16;  // the last item of the previous chunk, + 1
          else next = idx;                                                // the item before this one, + 1
          uint32_t info = (uint32_t)myrow | pi << 8 | type << 16;
          if (guessed) info |= I_GUESS;
          if (mslot == NONE) info |= I_UNFILED;
          r.items[idx] = Item{slot, info, mslot, next};
        }
        chunk = base_of(last_no);
        head = chunk + (last & 15) + 1;
        num += cins;
      }
      __syncthreads();   // (s_m[w][MA] was read above: the next step may overwrite the masks now)
      if (s_bail) break;
    }
    resolve_pending();
#ifdef PGX_BIG_STATS
    if (threadIdx.x == 0) {   // [3] steps, [4] rows committed, [5] steps cut at a duplicate, [6] steps ended by a containment, [7] steps that committed all their rows
      unsigned long long *line = r.spread + (blockIdx.x % SPREAD) * 8;
      atomicAdd(line + 3, (unsigned long long)st_steps), atomicAdd(line + 4, (unsigned long long)st_rows), atomicAdd(line + 5, (unsigned long long)st_cut);
      atomicAdd(line + 6, (unsigned long long)st_cont), atomicAdd(line + 7, (unsigned long long)st_full);
      atomicMax(&r.c->big_max_steps, st_steps);
      if (st_steps >= 64) atomicAdd(&r.c->big_long, 1u), atomicAdd(&r.c->big_long_n, n);
      atomicAdd(&r.c->big_evals, 1u);
    }
#endif
    if (threadIdx.x == 0) {
      if (s_bail) {   // (guard path: evaluated again by k_eval_rows, one partner at a time; the lists written so far are simply dropped)
        r.dirty[j] = 1, r.evaluated[j] = 0, r.parity[j] ^= 1;
        r.bflags[j] = (uint8_t)((r.bflags[j] & ~F_BIG));
      } else {
        r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
        r.bflags[j] = (uint8_t)(F_BIG | (dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
      }
    }
  }
  if (s_abort) break;
  }
  if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
}

// ---- apply the evaluated buckets' lists to the pair table: a group per bucket, a lane per item ---------------------------
__device__ __forceinline__ void apply_insertion(const R &r, uint32_t j, uint32_t pnew, const Item &im) {
  const uint32_t slot = im.pslot, type = (im.info >> 16) & 3;
  const uint32_t mine = own_enc(j, pnew, type);
  uint32_t v = r.ph[slot].own;
  for (;;) {
    if (v != 0 && own_bucket(v) < j) {  // an earlier bucket got in first: this evaluation is stale
      r.dirty[j] = 1;
      return;
    }
    const uint32_t prev = atomicCAS(&r.ph[slot].own, v, mine);
    if (prev == v) {
      if (v == 0 || own_bucket(v) > j) mark_readers(r, slot, j);  // absent -> present, or a later owner displaced (it reads the pair too)
      else if ((own_type(v) == T_OVERLAP) != (type == T_OVERLAP)) mark_readers(r, slot, j);  // ours before: readers see the type class
      return;
    }
    v = prev;
  }
}
__global__ __launch_bounds__(256) void k_update(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPW + (uint32_t)(lane / GL)) & 0x7FFFFFFFull;   // (a listed big bucket: k_count_b's mark off)
  const uint32_t j = (uint32_t)jj;
  const bool alive = jj < hi && r.evaluated[j] && !r.c->overflow;
  if (blockIdx.x == 0 && threadIdx.x == 0) r.c->nbig = 0;   // (k_eval_big has consumed the pass's list; the next pass starts a new one)
  // (the dirty statistics are NOT touched here: the list-mode blocks of this very launch read ndirty as their list length, and
  // the count that follows writes absolute values)
  if (!__ballot(alive)) return;
  const uint32_t pnew = alive ? r.parity[j] : 0, pold = pnew ^ 1;
  // what this evaluation inserts: take or refresh ownership (lowest bucket wins)
  uint32_t cur = alive ? r.ihead[j] : NIL;
  while (__ballot(cur != NIL)) {
    if (cur != NIL) {
      const uint32_t last = cur - 1, base = last & ~15u, cnt = (last & 15) + 1;
      const Item first = r.items[base];
      for (uint32_t o = (uint32_t)gl; o < cnt; o += GL) apply_insertion(r, j, pnew, o == 0 ? first : r.items[base + o]);
      cur = first.next;
    }
  }
  // what the previous evaluation inserted and this one did not refresh: withdraw
  cur = alive ? r.ohead[j] : NIL;
  while (__ballot(cur != NIL)) {
    if (cur != NIL) {
      const uint32_t last = cur - 1, base = last & ~15u, cnt = (last & 15) + 1;
      const uint32_t nxt = r.items[base].next;
      for (uint32_t o = (uint32_t)gl; o < cnt; o += GL) {
        const uint32_t slot = r.items[base + o].pslot;
        const uint32_t v = r.ph[slot].own;
        if (v != 0 && own_bucket(v) == j && own_parity(v) == pold)
          if (atomicCAS(&r.ph[slot].own, v, 0u) == v) mark_readers(r, slot, j);
      }
      cur = nxt;
    }
  }
  if (alive && gl == 0) r.evaluated[j] = 0, r.ohead[j] = NIL;
}

// ---- the dirty buckets: how many, in which range, and (while they fit) their list, LOWEST FIRST -------------------------
// Two small launches: blocks of CB buckets count theirs (16 flags per lane, one 16-byte load), then every block adds up the
// counts of the blocks before it and writes its ids at that offset.  The list is exactly ascending, so when more than LIST_CAP
// buckets are dirty the list keeps the LOWEST ones (evaluating those first wastes the fewest evaluations -- the round-1 form
// took list positions by atomics in arrival order, and atomics on one address cost ~12 ns each: with every block holding a
// dirty bucket a count took 0.46 ms, ten times per step).
constexpr uint32_t CB = 4096;  // buckets per block of the count kernels (256 lanes x 16)
__device__ __forceinline__ uint32_t dirty16(const R &r, uint32_t j0, uint32_t end) {  // bit i: bucket j0 + i < end is dirty (j0 a multiple of 16)
  if (j0 >= end) return 0;
  uint32_t m = 0;
  if (j0 + 16 <= end) {
    const uint4 v = *reinterpret_cast<const uint4 *>(r.dirty + j0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) m |= ((w[q] >> (8 * b)) & 0xFFu) ? 1u << (4 * q + b) : 0u;
  } else {
    for (uint32_t i = 0; j0 + i < end; ++i) m |= r.dirty[j0 + i] ? 1u << i : 0u;
  }
  return m;
}
__global__ __launch_bounds__(256) void k_count_a(R r, uint32_t rlo, uint32_t rhi, uint32_t *__restrict__ blk) {  // blk[3 b + {0, 1, 2}] = count, lowest, highest
  __shared__ uint32_t s_c[4], s_lo[4], s_hi[4];
  const uint32_t j0 = rlo + blockIdx.x * CB + threadIdx.x * 16;   // (rlo: a multiple of 16)
  const uint32_t m = dirty16(r, j0, rhi);
  uint32_t c = (uint32_t)__popc(m), lo = m ? j0 + (uint32_t)__builtin_ctz(m) : 0xFFFFFFFFu, hi = m ? j0 + 31u - (uint32_t)__builtin_clz(m) : 0u;
  for (int o = 32; o; o >>= 1) {
    c += (uint32_t)__shfl_xor((int)c, o, 64);
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_c[w] = c, s_lo[w] = lo, s_hi[w] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) r.c->nbig = 0;   // (k_count_b lists the big buckets by their POSITION in the list it writes: positions of an older list must be gone)
    blk[3 * blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    blk[3 * blockIdx.x + 1] = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
    blk[3 * blockIdx.x + 2] = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
  }
}
__global__ __launch_bounds__(256) void k_count_b(R r, uint32_t rlo, uint32_t rhi, const uint32_t *__restrict__ blk, uint32_t nblk) {
  __shared__ uint32_t s_part[4], s_lo[4], s_hi[4], s_tot[4], s_w[4];
  // the counts of the blocks before this one (and, in block 0, the totals of all of them)
  uint32_t before = 0, total = 0, lo = 0xFFFFFFFFu, hi = 0;
  const bool totals = blockIdx.x == 0;
  for (uint32_t b = threadIdx.x; b < nblk; b += 256) {
    const uint32_t c = blk[3 * b];
    if (b < blockIdx.x) before += c;
    if (totals) total += c, lo = min(lo, blk[3 * b + 1]), hi = max(hi, blk[3 * b + 2]);
  }
  for (int o = 32; o; o >>= 1) {
    before += (uint32_t)__shfl_xor((int)before, o, 64);
    if (totals) {
      total += (uint32_t)__shfl_xor((int)total, o, 64);
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_part[w] = before, s_tot[w] = total, s_lo[w] = lo, s_hi[w] = hi;
  const uint32_t j0 = rlo + blockIdx.x * CB + threadIdx.x * 16;   // (rlo: a multiple of 16)
  const uint32_t m = dirty16(r, j0, rhi);
  const uint32_t c = (uint32_t)__popc(m);
  uint32_t incl = c;   // lanes of a wavefront: inclusive scan
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  if (totals && threadIdx.x == 0) {
    const uint32_t n = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    r.c->ndirty = n;
    r.c->min_dirty = n ? min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])) : 0xFFFFFFFFu;
    r.c->max_dirty = n ? max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])) : 0u;
  }
  if (!m) return;
  uint32_t at = s_part[0] + s_part[1] + s_part[2] + s_part[3] + incl - c;
  for (int q = 0; q < w; ++q) at += s_w[q];
  // (round 3) a big bucket is marked in the list (bit 31: the narrow kernels and their wavefront slots pass it over -- its index is
  // beyond every `hi` -- and k_update takes the mark off) and entered in the big list right here, so that k_eval_big does not have
  // to wait for the narrow kernel of the pass to find it: the two run side by side
  for (uint32_t mm = m; mm && at < LIST_CAP; mm &= mm - 1, ++at) {
    const uint32_t j = j0 + (uint32_t)__builtin_ctz(mm);
    const bool big = (r.bflags[j] & F_BIG) != 0;
    r.dlist[at] = j | (big ? 0x80000000u : 0u);
    if (big) {   // (listed by POSITION: k_eval_big only takes the part of the list that the pass's narrow kernel and k_update cover)
      const uint32_t bat = atomicAdd(&r.c->nbig, 1u);
      if (bat < LIST_CAP) r.blist[bat] = at | 0x80000000u;
    }
  }
}

// Tail sweeps.  A pair whose alignment is rejected is not entered in the seen-pair table, so the next bucket holding both reads
// aligns it again from its own anchors -- and is rejected again, and so on through the ~30 buckets the two reads share: one
// sweep (one lone 0.33 ms alignment) per hand-over, which is what the last ~15 sweeps of a 4.5 Gbase set consist of.  Once
// the sweeps are small, a bucket that files an alignment therefore also files the one every other registered reader of that
// pair would ask for (its rows for the two reads, its anchors): the results are in the memo when those buckets come to it.
// A speculative request is just an alignment whose result the memo holds; at worst it is never asked for.
// file the alignment of entries (row, par) of a bucket whose records start at s0, unless the memo knows it already
__device__ __forceinline__ void file_entries(const R &r, uint32_t s0, uint32_t row, uint32_t par) {
  const Ent e0 = entry_of(r.y0[s0 + row]), e1 = entry_of(r.y0[s0 + par]);
  if (e0.pos1 < e1.pos1 || e0.rid == e1.rid) return;
  const uint32_t dir0 = r.dir[s0 + row], dir1 = r.dir[s0 + par], q_off = e0.pos1 - e1.pos1;
  if (q_off >= (1u << 30)) return;
  const unsigned long long a = (unsigned long long)e0.rid << 32 | e1.rid;
  const uint32_t bk = q_off << 2 | dir0 << 1 | dir1;
  uint32_t i = (uint32_t)mix64(a ^ mix64(bk)) & r.mmask;
  for (int probes = 0; probes < 1024; ++probes) {
    unsigned long long cur = r.mt[i].a;
    if (cur == 0) {
      cur = atomicCAS(&r.mt[i].a, 0ULL, a);
      if (cur == 0) {  // new: a request of its own
        r.mt[i].b = bk + 1;
        const uint32_t my = atomicAdd(&r.c->nreq, 1u);
        if (my >= r.req_cap) {
          atomicOr(&r.c->overflow, OV_REQS);
          return;
        }
        pgx_align_key key;
        key.rid0 = e0.rid, key.rid1 = e1.rid, key.q_off = q_off, key.dir0 = (uint8_t)dir0, key.dir1 = (uint8_t)dir1, key.pad[0] = key.pad[1] = 0;
        r.rq_key[my] = key;
        r.mt[i].req = my;
        return;
      }
    }
    if (cur == a && *(volatile uint32_t *)&r.mt[i].b == bk + 1) return;  // known already
    i = (i + 1) & r.mmask;
  }
}
__device__ __forceinline__ void file_for_reader(const R &r, uint32_t C, uint32_t rid_a, uint32_t rid_b) {
  const uint32_t b = r.bid[C], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
  int ia = -1, ib = -1;
  bool twice = false;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t rid = (uint32_t)(r.y0[s0 + i] >> 32);
    if (rid == rid_a) twice |= ia >= 0, ia = (int)i;
    else if (rid == rid_b) twice |= ib >= 0, ib = (int)i;
  }
  if (ia < 0 || ib < 0 || twice) return;
  file_entries(r, s0, (uint32_t)min(ia, ib), (uint32_t)max(ia, ib));  // (the row is the entry with the smaller index)
}

// ---- file the alignments the converged lists still need ---------------------------------------------------------------
// (Only buckets that are not dirty right now are filed -- the others are about to be evaluated again.  Filing while the sweep
// is still running is always safe: a request is just an alignment whose result the memo will hold; at worst it is never
// asked for again.)
__global__ __launch_bounds__(256) void k_file(R r, uint32_t limit) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < limit && (r.bflags[j] & F_UNFILED) && !r.dirty[j];
  __shared__ uint32_t s_tot[4], s_base;
  if (!__syncthreads_or(active)) return;   // (block-uniform)
  uint32_t cnt = 0;
  if (active)
    for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next) cnt += (r.items[it - 1].info & I_UNFILED) ? 1u : 0u;
  // request numbers: ONE atomic per block.  The request counter is one address, and an L2 channel serves same-address atomics one
  // wavefront-instruction at a time (~12 ns): with an add per wavefront the first sweep's launch -- 9.4 M requests at c4s, 46 M in a
  // human-scale chunk -- lasted exactly requests / 64 x 12 ns (2.0 ms, 8.7 ms)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += t;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
  if (lane == 63) s_tot[wv] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t all = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    s_base = all ? atomicAdd(&r.c->nreq, all) : 0u;
  }
  __syncthreads();
  uint32_t base = s_base;
  for (int w = 0; w < wv; ++w) base += s_tot[w];
  if (!__ballot(active)) return;
  // (from here on every lane of the wavefront stays in step: the fan-out below is done by all of them together)
  bool run = active && cnt != 0;
  if (active && !cnt) r.bflags[j] &= (uint8_t)~F_UNFILED;
  if (!__ballot(run)) return;
  if ((unsigned long long)base + total > r.req_cap) {
    if (run) atomicOr(&r.c->overflow, OV_REQS);
    return;
  }
  uint32_t my = base + incl - cnt;
  const uint32_t b = run ? r.bid[j] : 0u, s0 = run ? r.bstart[b] : 0u;
  uint32_t it = run ? r.ihead[j] : NIL;
  const bool tail = r.tail != 0;
  const uint32_t nn = run ? r.bstart[b + 1] - s0 : 0u;
  for (;;) {
    // ---- this lane's next unfiled item ----
    bool fan = false;             // the item filed a NEW alignment (tail mode): its pair's other readers are looked at below
    bool ah = false;              // the item was filed in tail mode: its row's next partners are filed ahead below
    uint32_t f_slot = 0, f_a = 0, f_b = 0, ah_ai = 0, ah_pi = 0;
    while (it != NIL && !fan && !ah) {
      Item &im = r.items[it - 1];
      const uint32_t nxt = im.next;
      if (im.info & I_UNFILED) {
        const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF;
        const Ent e0 = entry_of(r.y0[s0 + ai]), e1 = entry_of(r.y0[s0 + pi]);
        const uint32_t dir0 = r.dir[s0 + ai], dir1 = r.dir[s0 + pi], q_off = e0.pos1 - e1.pos1;
        const unsigned long long a = (unsigned long long)e0.rid << 32 | e1.rid;
        const uint32_t bk = q_off << 2 | dir0 << 1 | dir1;
        pgx_align_key key;
        key.rid0 = e0.rid, key.rid1 = e1.rid, key.q_off = q_off, key.dir0 = (uint8_t)dir0, key.dir1 = (uint8_t)dir1, key.pad[0] = key.pad[1] = 0;
        // find or insert.  Another lane may be inserting the same key right now: its `b` may still read 0, in which case this
        // lane files a duplicate in another slot (same alignment, same result -- harmless).
        uint32_t i = (uint32_t)mix64(a ^ mix64(bk)) & r.mmask, found = NONE;
        bool fresh = false;
        for (int probes = 0; probes < 1024; ++probes) {
          unsigned long long cur = r.mt[i].a;
          if (cur == 0) {
            cur = atomicCAS(&r.mt[i].a, 0ULL, a);
            if (cur == 0) {
              r.mt[i].b = bk + 1;
              found = i, fresh = true;
              break;
            }
          }
          if (cur == a && *(volatile uint32_t *)&r.mt[i].b == bk + 1) {
            found = i;
            break;
          }
          i = (i + 1) & r.mmask;
        }
        if (found == NONE) {   // (this bucket stays unfiled; the overflow bit sends the walk to larger tables)
          atomicOr(&r.c->overflow, OV_MEMO);
          run = false, it = NIL;
          break;
        }
        r.rq_key[my] = key;  // (a request slot whose key was already filed by someone else just repeats that alignment)
        if (fresh) r.mt[found].req = my;
        ++my;
        im.mslot = found;
        im.info &= ~I_UNFILED;
        if (tail) ah = true, ah_ai = ai, ah_pi = pi;   // ... and the row's next partners (below)
        if (tail && fresh) fan = true, f_slot = im.pslot, f_a = e0.rid, f_b = e1.rid;
      }
      it = nxt;
    }
    const uint64_t fm = __ballot(fan), am = __ballot(ah);
    if (!fm && !am && !__ballot(it != NIL)) break;
    // ---- tail mode: the row's next partners -- if this candidate is rejected the row goes on to them (a row of a repeat-rich bucket can
    // have dozens of candidates, each rejection otherwise costing a sweep) -- a partner per lane.  (Through round 3 the filing lane walked
    // its r.tail partners itself, a dependent memo probe each, item after item: the tail sweeps' k_file launches took up to 7 ms at c4s.)
    for (uint64_t mm = am; mm; mm &= mm - 1) {
      const int L = __builtin_ctzll(mm);
      const uint32_t sL = (uint32_t)__shfl((int)s0, L, 64), aL = (uint32_t)__shfl((int)ah_ai, L, 64), pL = (uint32_t)__shfl((int)ah_pi, L, 64);
      const uint32_t nL = (uint32_t)__shfl((int)nn, L, 64);
      for (uint32_t p = pL + 1 + (uint32_t)lane; p < nL && p <= pL + r.tail; p += 64) file_entries(r, sL, aL, p);
    }
    // ---- tail mode: the alignment every OTHER reader of a newly requested pair would ask for (file_for_reader) -- by the whole
    // wavefront, a reader bucket per lane.  (Round 2 left this to the filing lane alone: a pair of a repeat-rich set has dozens of
    // readers of up to 128 entries each, scanned one after the other -- k_file was 21 ms of a c4s step and 42 ms of c5s'.)
    for (uint64_t mm = fm; mm; mm &= mm - 1) {
      const int L = __builtin_ctzll(mm);
      const uint32_t ps = (uint32_t)__shfl((int)f_slot, L, 64), ra = (uint32_t)__shfl((int)f_a, L, 64), rb2 = (uint32_t)__shfl((int)f_b, L, 64);
      const uint32_t jj = (uint32_t)__shfl((int)j, L, 64);
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[ps >> r.cshift]);
      const uint32_t c = min(w[0], NIN);
      for (uint32_t q = (uint32_t)lane; q < c; q += 64) {
        const uint32_t rb = w[2 + q];
        if (rb != 0 && rb - 1 != jj && rb - 1 < r.nb) file_for_reader(r, rb - 1, ra, rb2);
      }
      if (c >= NIN) {   // the overflow list: every lane walks it (one broadcast load per node), node i goes to lane i % 64
        uint32_t idx = 0;
        for (uint32_t nd = w[1]; nd != NIL; nd = r.rn[nd - 1].next, ++idx)
          if ((idx & 63u) == (uint32_t)lane) {
            const uint32_t rb = r.rn[nd - 1].bucket;
            if (rb != jj && rb < r.nb) file_for_reader(r, rb, ra, rb2);
          }
      }
    }
  }
  if (run) r.bflags[j] &= (uint8_t)~F_UNFILED;
}

// ---- check the guesses against the results ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_settle(R r) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= r.nb || !(r.bflags[j] & F_GUESS)) return;
  const uint32_t b = r.bid[j], s0 = r.bstart[b];
  bool bad = false, remain = false;
  for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next) {
    Item &im = r.items[it - 1];
    if (!(im.info & I_GUESS)) continue;
    const uint32_t req = im.mslot != NONE ? r.mt[im.mslot].req : NONE;
    if (req >= r.settled) {
      remain = true;
      continue;
    }
    const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF, gtype = (im.info >> 16) & 3;
    const Ent e0 = entry_of(r.y0[s0 + ai]), e1 = entry_of(r.y0[s0 + pi]);
    uint32_t type;
    const bool acc = classify(r.rq_res[req], r.rlen[e0.rid], r.rlen[e1.rid], e0.pos1 - e1.pos1, &type);
    if (!acc || type != gtype) bad = true;
#ifdef PGX_SETTLE_STATS
    if (!acc || type != gtype) {
      const uint32_t rl0 = r.rlen[e0.rid], rl1 = r.rlen[e1.rid], qo = e0.pos1 - e1.pos1;
      const uint32_t ol = min(rl0 - qo, rl1);
      const pgx_match mm = r.rq_res[req];
      int cat = acc ? 3 : (ol <= 520 ? 4 : (mm.q_end == 0 && mm.t_end == 0 ? 5 : 6));
      atomicAdd(&r.spread[((j >> 6) % SPREAD) * 8 + cat], 1ULL);
    }
#endif
    else im.info &= ~I_GUESS;
  }
  if (bad) r.dirty[j] = 1;
  if (!remain) r.bflags[j] &= (uint8_t)~F_GUESS;
}

// ---- the ovlp_t records, bucket by bucket in visit order, each bucket's in evaluation order ---------------------------
__global__ __launch_bounds__(256) void k_emit(R r, const uint32_t *__restrict__ off, pgx_ovlp *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long lk = 0, sk = 0, ck = 0;
  if (j < r.nb) {
    lk = r.lookups[j], sk = r.skips[j];
    const uint32_t num = r.inum[j];
    if (num) {
      const uint32_t b = r.bid[j], s0 = r.bstart[b];
      uint32_t k = 0;
      for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next, ++k) {  // the list runs newest first
        const Item im = r.items[it - 1];
        const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF;
        const uint64_t ya = r.y0[s0 + ai], yb = r.y0[s0 + pi];
        pgx_ovlp o;
        o.y0 = ya, o.y1 = yb;
        o.rl0 = r.rlen[(uint32_t)(ya >> 32)], o.rl1 = r.rlen[(uint32_t)(yb >> 32)];
        o.strand0 = r.dir[s0 + ai], o.strand1 = r.dir[s0 + pi], o.ovlp_type = (uint8_t)((im.info >> 16) & 3), o.pad0 = 0;
        o.match = r.rq_res[r.mt[im.mslot].req];
        o.pad1 = 0;
        out[(size_t)off[j] + (num - 1 - k)] = o;
        ck += record_checksum(o, (uint64_t)off[j] + (num - 1 - k));
      }
    }
  }
  // totals: one atomic pair per wavefront
  for (int o = 32; o; o >>= 1) {
    lk += (unsigned long long)__shfl_xor((int)(lk >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)lk, o, 64);
    sk += (unsigned long long)__shfl_xor((int)(sk >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)sk, o, 64);
    ck += (unsigned long long)__shfl_xor((int)(ck >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)ck, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (lk | sk | ck)) {
    unsigned long long *line = r.spread + ((j >> 6) % SPREAD) * 8;
    atomicAdd(line + 1, lk);
    atomicAdd(line + 2, sk);
#if !defined(PGX_BIG_STATS) && !defined(PGX_SETTLE_STATS)   // (the statistics builds count in the same words)
    atomicAdd(line + 7, ck);
#endif
  }
}

// read pairs the walk has entered in the pair table (what the next stage's table is sized by: dev_replay)
__global__ __launch_bounds__(256) void k_count_pairs(const PHot *__restrict__ ph, uint32_t cap, unsigned long long *__restrict__ out) {
  uint32_t c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) c += ph[i].key != 0;
  for (int o = 32; o; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
uint32_t pow2_at_least(size_t x) {
  size_t c = 1024;
  while (c < x) c <<= 1;
  return (uint32_t)c;
}
unsigned cdiv256(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

namespace {
// one attempt with the given table sizes (multiples of the defaults); returns 0, or the OV_* bits of what overflowed
uint32_t replay_attempt(const pgx_seqdb *db, const DevicePairs &dp, const uint32_t *visit_bids, const uint32_t *d_bids, size_t nb, size_t n_entries,
                        uint32_t bestn, int band, bool predict, const std::function<pgx_ovlp *(size_t)> &alloc_out,
                        size_t *n_out, pgx_overlap_stats *st, bool trace, const double *mult, double *usage) {
  const double t0 = now_ms();
  hipStream_t s = ctx().stream;
  const size_t ne = std::max<size_t>(n_entries, 1024);
  MemTag mem_tag("replay.other");
  R r;
  memset(&r, 0, sizeof(r));
  r.nb = (uint32_t)nb;
  DevBuf<uint32_t> bid(d_bids ? 0 : nb);   // (d_bids: the visit list was assembled on the device, dev_place_bids)
  if (!d_bids) bid.upload(visit_bids, nb);
  r.bid = d_bids ? d_bids : bid.p, r.bstart = dp.bstart.p, r.y0 = dp.y0.p, r.dir = dp.dir.p, r.rlen = db->d_rlen.p;
  const uint32_t pcap = pow2_at_least((size_t)(ne * mult[3])), mcap = pow2_at_least((size_t)(ne * mult[4]));
  DevBuf<PHot> ph;
  DevBuf<PCold> pc;
  DevBuf<MSlot> mt;
  {
    MemTag t1("replay.pair_table_hot");
    ph.alloc(pcap);
  }
  // The hot table wants a load of <= 0.4 (dev_replay) and the reader lists are 16 x as large per slot (34 GB for a full-size configs[3]
  // chunk).  Where that is more than a third of the free device memory, 2 (4, 8) neighbouring hot slots share one list: all sharing costs is
  // that a change of one pair also marks the other's later readers dirty -- a spurious re-evaluation, never a missed one (measured at c4:
  // + 19 % evaluations, + 1.3 % step time for 17 GB less).  PGX_REPLAY_COLD_SHIFT forces the shift (tests: 0 .. 3).
  r.cshift = 0;
  if (getenv("PGX_REPLAY_COLD_SHIFT")) {
    r.cshift = (uint32_t)std::min(3, std::max(0, atoi(getenv("PGX_REPLAY_COLD_SHIFT"))));
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      while (r.cshift < 3 && ((size_t)pcap >> r.cshift) * sizeof(PCold) > (free_b + dev_cache_free_bytes()) / 3) ++r.cshift;   // (the last stage's tables are in the cache)
  }
  const size_t ccap = ((size_t)pcap >> r.cshift) + 1;
  {
    MemTag t2("replay.pair_table_readers");
    pc.alloc(ccap);
  }
  {
    MemTag t3("replay.memo_table");
    mt.alloc(mcap);
  }
  r.ph = ph.p, r.pc = pc.p, r.pmask = pcap - 1, r.mt = mt.p, r.mmask = mcap - 1;
  r.item_cap = (uint32_t)std::min<size_t>((size_t)(((size_t)(ne * 6) + nb * (size_t)64) * mult[0]) + (size_t)(nb / GPW + 2 + SPARSE_CAP + 8 + BIG_WG * BIG_NW) * ICH + (1u << 20), 0x7FFFFFF0u);
  r.rn_cap = (uint32_t)std::min<size_t>((size_t)(ne * 8 * mult[1]) + (1u << 20), 0x7FFFFFF0u);
  r.req_cap = (uint32_t)std::min<size_t>((size_t)(ne * 1 * mult[2]) + 65536, 0x7FFFFFF0u);
  DevBuf<Item> items;
  DevBuf<RNode> rn;
  DevBuf<pgx_align_key> rq_key;
  DevBuf<pgx_match> rq_res;
  {
    MemTag t4("replay.items");
    items.alloc(r.item_cap);
  }
  {
    MemTag t5("replay.reader_nodes");
    rn.alloc(r.rn_cap);
  }
  {
    MemTag t6("replay.requests");
    rq_key.alloc(r.req_cap), rq_res.alloc(r.req_cap);
  }
  r.items = items.p, r.rn = rn.p, r.rq_key = rq_key.p, r.rq_res = rq_res.p;
  DevBuf<uint8_t> bytes(nb * 5);
  DevBuf<uint32_t> words(nb * 5);
  r.dirty = bytes.p, r.evaluated = bytes.p + nb, r.parity = bytes.p + 2 * nb, r.bflags = bytes.p + 3 * nb, r.ever = bytes.p + 4 * nb;
  r.ihead = words.p, r.inum = words.p + nb, r.ohead = words.p + 2 * nb, r.lookups = words.p + 3 * nb, r.skips = words.p + 4 * nb;
  // big buckets (>= big_min entries, no read twice) are evaluated by a workgroup each: k_eval_big beside every evaluation launch
  // (measured at C4 scale: the buckets that hold a read twice -- 7 k of 2.3 M, up to 128 entries, one partner at a time in the narrow
  // kernels -- were what every sparse pass waited for; big buckets WITHOUT a repeated read are rare (40 of 2.3 M beyond 48 entries:
  // the multiplicity cut-off removes the repeat families' shimmers) and stay with the narrow kernels by default)
  // (round 4: 48 -- with k_eval_big's walk a third shorter the long buckets WITHOUT a repeated read are better off there too: k_eval_rows 38 -> 22 ms
  //  per c4s step, hidden behind k_eval_big as it is: 360 -> 354 ms; from 24 entries on k_eval_big doubles, 411 ms.  Sixteen wavefronts = eight
  //  rows a step (-DPGX_BIG_NW=16): k_eval_big 56 -> 186 ms.)
  r.big_min = getenv("PGX_REPLAY_BIG") ? (uint32_t)std::max(0, atoi(getenv("PGX_REPLAY_BIG"))) : 48u;
  r.dup_min = getenv("PGX_REPLAY_DUP") ? (uint32_t)std::max(0, atoi(getenv("PGX_REPLAY_DUP"))) : 12u;
  DevBuf<uint4> wcur(nb + 2 + SPARSE_CAP + 1 + (size_t)BIG_WG * BIG_NW);  // (one slot per wavefront of k_eval: GPW buckets each; list mode; k_eval_big)
  r.wcur = wcur.p, r.wlist0 = (uint32_t)(nb + 2), r.wbig0 = (uint32_t)(nb + 2 + SPARSE_CAP + 1);
  DevBuf<uint32_t> dlist(LIST_CAP), blist(LIST_CAP + 64);
  r.dlist = dlist.p, r.blist = blist.p;
  DevBuf<Counters> dc(1);
  r.c = dc.p;
  DevBuf<unsigned long long> spread(SPREAD * 8);
  r.spread = spread.p;
  PGX_HIP(hipMemsetAsync(spread.p, 0, SPREAD * 8 * sizeof(unsigned long long), s));
  {
    const int mq = END_FUZZ * 2 - 8, mt = END_FUZZ * 2 - 8;
    r.predict = predict ? std::max(mq, 1) : 0, r.predict2 = mt;
  }
  r.bestn = bestn, r.settled = 0;
  const bool timed_misc = getenv("PGX_REPLAY_TIMING") && atoi(getenv("PGX_REPLAY_TIMING")) != 0;   // "replay_misc" / "replay_emit" in pgx_timing_get
  std::optional<KernelTimer> tm_setup;
  if (timed_misc) tm_setup.emplace("replay_misc", nb);
  PGX_HIP(hipMemsetAsync(ph.p, 0, (size_t)pcap * sizeof(PHot), s));
  PGX_HIP(hipMemsetAsync(pc.p, 0, ccap * sizeof(PCold), s));
  PGX_HIP(hipMemsetAsync(mt.p, 0, (size_t)mcap * sizeof(MSlot), s));
  PGX_HIP(hipMemsetAsync(bytes.p, 0, nb * 5, s));
  PGX_HIP(hipMemsetAsync(words.p, 0, nb * 5 * sizeof(uint32_t), s));
  PGX_HIP(hipMemsetAsync(dc.p, 0, sizeof(Counters), s));
  hipLaunchKernelGGL(k_init_slots, dim3(cdiv256(wcur.n)), dim3(256), 0, s, wcur.p, (uint32_t)wcur.n, (uint32_t)(nb / GPW + 2), r.wlist0, dc.p);
  if (trace) {
    DevBuf<uint32_t> hist(32);
    PGX_HIP(hipMemsetAsync(hist.p, 0, 32 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_setup, dim3(cdiv256(nb)), dim3(256), 0, s, r, hist.p);
    uint32_t h[32];
    hist.download(h, 32);
    sync();
    fprintf(stderr, "[pgx]   buckets by entries / 8 (last class: >= 120):");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", h[i]);
    fprintf(stderr, "\n[pgx]   ... of those holding a read twice (one partner at a time):");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", h[16 + i]);
    fprintf(stderr, "\n");
  } else {
    hipLaunchKernelGGL(k_setup, dim3(cdiv256(nb)), dim3(256), 0, s, r, (uint32_t *)nullptr);
  }
  tm_setup.reset();

  bool use_big = r.big_min || r.dup_min;   // (decided below, once k_setup has counted the big buckets)
  static Counters *hc = nullptr;  // pinned mirror of the device counters
  static unsigned long long *hs = nullptr;  // pinned mirror of the spread totals
  static uint32_t *reset3 = nullptr;  // {ndirty, min_dirty, max_dirty} before a count (pinned, constant)
  if (!hc) {
    PGX_HIP(hipHostMalloc((void **)&hc, sizeof(Counters), hipHostMallocDefault));
    PGX_HIP(hipHostMalloc((void **)&hs, SPREAD * 8 * sizeof(unsigned long long), hipHostMallocDefault));
    PGX_HIP(hipHostMalloc((void **)&reset3, 3 * sizeof(uint32_t), hipHostMallocDefault));
    reset3[0] = 0, reset3[1] = 0xFFFFFFFFu, reset3[2] = 0;
  }
  auto fetch = [&](bool totals) {  // counters (and, for the trace / the statistics, the spread totals) to the host; synchronises
    PGX_HIP(hipMemcpyAsync(hc, dc.p, sizeof(Counters), hipMemcpyDeviceToHost, s));
    if (totals) PGX_HIP(hipMemcpyAsync(hs, spread.p, SPREAD * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    sync();
    if (totals) {
      hc->evals = hc->lookups = hc->skips = 0;
      hc->records = 0;   // (re-used as the stream checksum k_emit adds up in word 7 of every line)
      for (uint32_t i = 0; i < SPREAD; ++i) hc->evals += hs[i * 8], hc->lookups += hs[i * 8 + 1], hc->skips += hs[i * 8 + 2], hc->records += hs[i * 8 + 7];
#ifdef PGX_BIG_STATS
      {
        unsigned long long b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
        for (uint32_t i = 0; i < SPREAD; ++i) b3 += hs[i * 8 + 3], b4 += hs[i * 8 + 4], b5 += hs[i * 8 + 5], b6 += hs[i * 8 + 6], b7 += hs[i * 8 + 7];
        fprintf(stderr, "[pgx]   k_eval_big so far: %u evaluations, %llu steps, %llu rows committed; steps cut at a duplicate %llu, ended by a containment %llu, all rows %llu; longest %u steps, %u evaluations of >= 64 steps (entries %u)\n",
                hc->big_evals, b3, b4, b5, b6, b7, hc->big_max_steps, hc->big_long, hc->big_long_n);
      }
#endif
#ifdef PGX_SETTLE_STATS
      unsigned long long c3 = 0, c4 = 0, c5 = 0, c6 = 0;
      for (uint32_t i = 0; i < SPREAD; ++i) c3 += hs[i * 8 + 3], c4 += hs[i * 8 + 4], c5 += hs[i * 8 + 5], c6 += hs[i * 8 + 6];
      fprintf(stderr, "[pgx]   wrong guesses so far: type %llu, rejected short overlap %llu, rejected no match %llu, rejected other %llu\n", c3, c4, c5, c6);
#endif
    }
  };
  if (use_big) {   // no big bucket in this set (uniform-random genomes): none of the ~100 passes of a step launches k_eval_big
    fetch(false);
    use_big = hc->nbig_total != 0;
  }
  const uint32_t nblk = (uint32_t)((nb + CB - 1) / CB);
  DevBuf<uint32_t> cblk((size_t)nblk * 3);
  auto launch_count = [&](uint32_t lo = 0, uint32_t hi = 0xFFFFFFFFu) {  // count + list of the dirty buckets of [lo, hi)
    hi = std::min<uint32_t>(hi, (uint32_t)nb);
    const uint32_t nbl = std::max<uint32_t>(1, (hi - lo + CB - 1) / CB);
    std::optional<KernelTimer> tmc;
    if (timed_misc) tmc.emplace("replay_misc", 0);
    hipLaunchKernelGGL(k_count_a, dim3(nbl), dim3(256), 0, s, r, lo, hi, cblk.p);
    hipLaunchKernelGGL(k_count_b, dim3(nbl), dim3(256), 0, s, r, lo, hi, cblk.p, nbl);
  };
  auto read_counters = [&](bool count_dirty) {
    if (count_dirty) {  // reset the three dirty statistics, keep the rest
      launch_count();
    }
    fetch(true);
    return hc->overflow == 0;
  };
  const size_t window = getenv("PGX_REPLAY_WIN") ? (size_t)atoll(getenv("PGX_REPLAY_WIN")) : (size_t)262144;
  // The schedule's constants.  Every one of them was an environment knob through round 3; tools/knob_sweep.sh (profiles/r04f_knob_sweep_c4s.txt)
  // moved each over its plausible range on the repeat-rich 9-Gbase set: 391-405 ms per step whatever the setting (only k_eval_big behind
  // instead of beside the narrow kernel is worse, 417 ms), so they are constants now; PGX_REPLAY_WIN / _K stay because the parity
  // tests randomise the schedule with them.
  const size_t win0 = 16384, win1 = 131072;   // the first pass ramps its window from win0 up to win1
  // dense rounds only while more than 1 / dense_den of the buckets is dirty.  (Through round 4 also from 65,536 dirty buckets on: the second sweep of a
  // full-size c4 chunk -- 275 k wrong guesses among 2.7 M buckets -- then went window by window, 41 passes each as long as its longest big bucket, 52 ms;
  // as sparse passes from the list: 7.06 -> 6.97 s per step, + 0.3 % evaluations.)
  const size_t dense_min = (size_t)1 << 40;
  const size_t tail_max = 4000;               // tail mode (file_for_reader, look-ahead) once a sweep asks for at most this many alignments, or 1/256 of the first sweep's
  const uint32_t ahead = 24u;                 // tail mode: partners of a row filed ahead
  // (round 6, tried and removed: k_settle filing, right where it finds a REJECTED guess, the alignment every other reader of that pair would ask
  //  for and the row's next 0 / 2 / 6 partners, aligned before the next sweep -- bit-exact, and no sweep fewer: 18-19 sweeps and 6.21-6.33 s per
  //  full-size c4 step against 18 and 6.13 s without, profiles/r06b_settle_fan_c4.txt.  The sweeps behind the second are not hand-overs of one
  //  pair between its readers -- tail mode's file_for_reader already covers those -- but a dependency chain through DIFFERENT pairs.)
  const bool use_win_list = true;
  // evaluate / update iterations per window of a dense round: 2 (through round 4: 3; at full-size c4 6.89 -> 6.81 s per step with 2 and 6.90 with 1,
  // c3 / c4s / c5s unchanged; the third iteration of a window mostly re-runs k_eval_big's longest bucket for a handful of dirty buckets that the
  // sparse passes behind the round pick up anyway)
  const int inner = getenv("PGX_REPLAY_K") ? atoi(getenv("PGX_REPLAY_K")) : 2;
  const bool wide = true;          // sparse passes: a wavefront per bucket, four rows per step
  const size_t dense_den = 3;      // dense rounds while more than 1/dense_den of the buckets is dirty
  const bool wide_dense = false;   // (the dense rounds keep 16 lanes per bucket: measured)
  // sparse passes per host round trip.  Round 2 measured 2 .. 8 within 1 % of each other and kept 4; with the GPU no longer waiting for the
  // host elsewhere (round 3) the round trips and the k_file pass that precedes each one show: 8 instead of 4 = replay kernels 38.0 -> 35.5 ms
  // and the step 113.4 -> 111.2 ms at c3, c4s 437 -> 429 ms, c5s 651 -> 639 (6), the E. coli-size set unchanged
  const bool big_side = true;   // k_eval_big beside the narrow kernel of a pass, on a second stream
  static hipStream_t side_stream = nullptr;
  static hipEvent_t side_ev[2] = {nullptr, nullptr};
  if (!side_stream) {
    PGX_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    for (auto &e : side_ev) PGX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    static ShutdownHook h_([] {   // (a later pgx_init may choose another device)
      if (side_stream) (void)hipStreamDestroy(side_stream), side_stream = nullptr;
      for (auto &e : side_ev)
        if (e) (void)hipEventDestroy(e), e = nullptr;
    });
  }
  const int big_every = 1;   // sparse passes per k_eval_big launch
  const int chain = 8;       // sparse passes per host round trip
  static const bool deep = getenv("PGX_TRACE") && atoi(getenv("PGX_TRACE")) >= 2;  // per-kernel wall times (synchronises after every launch)
  const bool timed = getenv("PGX_REPLAY_TIMING") && atoi(getenv("PGX_REPLAY_TIMING")) != 0;  // "replay" in pgx_timing_get
  double td = 0, t_eval = 0, t_upd = 0;
  size_t first_req = 0;  // requests [first_req, ...) belong to the running sweep
  unsigned sweeps = 0, rounds_total = 0;
  uint32_t d_lo = 0, d_hi = (uint32_t)nb;  // range holding the dirty buckets
  size_t n_dirty = nb;                     // as of the last fetch
  bool have_list = false;                  // the device holds a dirty list (a count has run since the last evaluations)
  bool known = true;                       // n_dirty / d_lo / d_hi are current
  double align_ms = 0;
  size_t first_batch = 0;   // alignments the first sweep asked for
  auto count_dirty = [&](bool reset = false) {  // no host round trip: ndirty, the range and the list stay on the device
    (void)reset;
    launch_count();
    have_list = true;
  };
  for (;;) {
    ++sweeps;
    const double p0 = now_ms();
    unsigned rounds = 0;
    for (;;) {  // until no bucket is dirty
      ++rounds;
      if (known && (n_dirty > dense_min || n_dirty * dense_den > nb)) {  // many of the buckets: a group of 16 lanes each, in bucket order
        // dense: window by window, in order (a window's buckets mostly depend on earlier windows)
        const size_t a_lo = d_lo & ~(size_t)63;  // (aligned: a bucket always belongs to the same wavefront slot)
        const size_t win = n_dirty > window / 4 ? window : (size_t)(d_hi - a_lo);
        // the very first pass ramps the window up (win0, 2 win0, ... window): with an empty pair table every bucket of a window
        // believes it owns all its pairs, and the early windows -- where nothing is "seen" yet -- re-evaluate nearly all of their
        // buckets; smaller windows there mean fewer wasted evaluations (2.20 M -> 1.7 M at 4.5 Gbases), later ones are launch-bound
        const bool first_pass = sweeps == 1 && rounds == 1;
        const size_t cap = first_pass ? std::min(win1, win) : win;
        size_t step = first_pass ? std::min(win0, cap) : win;
        for (size_t lo = a_lo, nxt; lo < d_hi; lo = nxt, step = std::min(step * 2, cap)) {
          nxt = lo + step;
          const uint32_t hi = (uint32_t)std::min<size_t>(d_hi, nxt);
          for (int k = 0; k < inner; ++k) {
            std::optional<KernelTimer> tm;  // PGX_REPLAY_TIMING=1: "replay_dense" = k_eval, "replay_rows" = k_eval_rows, "replay_update" = k_update
            if (timed) tm.emplace(wide_dense ? "replay_rows" : "replay_dense", k == 0 ? hi - lo : 0);  // (units: buckets of the window, once)
            if (deep) sync(), td = now_ms();
            // all but the very first evaluation of a window find only a part of its buckets dirty: they run from the window's
            // list (wavefronts full of live buckets) instead of over every bucket of the window
            const bool from_list = use_win_list && !(first_pass && k == 0) && !wide_dense && hi - lo <= LIST_CAP;
            if (from_list) launch_count((uint32_t)lo, hi);
            const bool side = use_big && from_list && big_side && !timed && !deep;   // (the window's big buckets are in the count's list: beside k_eval)
            if (side) {
              PGX_HIP(hipEventRecord(side_ev[0], s));
              PGX_HIP(hipStreamWaitEvent(side_stream, side_ev[0], 0));
              hipLaunchKernelGGL(k_eval_big, dim3((unsigned)std::min<size_t>(BIG_WG, hi - lo)), dim3(64 * BIG_NW), 0, side_stream, r, (uint32_t)LIST_CAP,
                                 (uint32_t)nb, DEV_LIST_WIN);
              PGX_HIP(hipEventRecord(side_ev[1], side_stream));
            }
            if (wide_dense) hipLaunchKernelGGL((k_eval_rows<64, 16>), dim3(cdiv256((size_t)(hi - lo) * 64)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            else if (from_list) hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST_WIN);
            else hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            tm.reset();
            if (side) {
              PGX_HIP(hipStreamWaitEvent(s, side_ev[1], 0));
            } else if (use_big) {
              if (timed) tm.emplace("replay_big", 0);
              const unsigned wgs = (unsigned)std::min<size_t>(BIG_WG, hi - lo);
              if (from_list) hipLaunchKernelGGL(k_eval_big, dim3(wgs), dim3(64 * BIG_NW), 0, s, r, (uint32_t)LIST_CAP, (uint32_t)nb, DEV_LIST_WIN);
              else hipLaunchKernelGGL(k_eval_big, dim3(wgs), dim3(64 * BIG_NW), 0, s, r, (uint32_t)lo, hi, 0u);
              tm.reset();
            }
            if (deep) sync(), t_eval += now_ms() - td, td = now_ms();
            if (timed) tm.emplace("replay_update", 0);
            if (from_list) hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST_WIN);
            else hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            tm.reset();
            if (deep) sync(), t_upd += now_ms() - td, fprintf(stderr, "[pgx]     iteration: eval %.3f ms, update %.3f ms\n", t_eval, t_upd), t_eval = t_upd = 0;
          }
        }
        count_dirty();
      }
      // sparse: `chain` passes straight from the list the last count left on the device -- the kernels read its length there,
      // so the host is not in the loop (a pass with nothing to do costs two empty launches); what a pass dirties is listed by
      // the count behind it
      if (!have_list) count_dirty(true);
      {
        const size_t est = known && n_dirty <= SPARSE_CAP / 4 ? std::max<size_t>(n_dirty * 4, 1024) : (size_t)SPARSE_CAP;
        const unsigned groups = (unsigned)std::min<size_t>(est, SPARSE_CAP);
        for (int c = 0; c < chain; ++c) {
          std::optional<KernelTimer> tm;
          const bool big_now = use_big && (c % big_every == big_every - 1 || c == chain - 1);
          // the big buckets of the pass (listed by the count, k_count_b) beside the narrow kernel, on a second stream: a launch of
          // k_eval_big lasts as long as its longest bucket (~0.4 ms of dependent probes), whatever else the GPU could be doing
          const bool side = big_now && big_side && !timed && !deep;
          if (side) {
            PGX_HIP(hipEventRecord(side_ev[0], s));
            PGX_HIP(hipStreamWaitEvent(side_stream, side_ev[0], 0));
            hipLaunchKernelGGL(k_eval_big, dim3(std::min<unsigned>(BIG_WG, groups)), dim3(64 * BIG_NW), 0, side_stream, r, groups, (uint32_t)nb, DEV_LIST);
            PGX_HIP(hipEventRecord(side_ev[1], side_stream));
          }
          if (timed) tm.emplace(wide ? "replay_rows" : "replay_dense", 0);
          if (wide) hipLaunchKernelGGL((k_eval_rows<64, 16>), dim3(cdiv256((size_t)groups * 64)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          else hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)groups * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          tm.reset();
          if (side) {
            PGX_HIP(hipStreamWaitEvent(s, side_ev[1], 0));
          } else if (big_now) {
            if (timed) tm.emplace("replay_big", 0);
            hipLaunchKernelGGL(k_eval_big, dim3(std::min<unsigned>(BIG_WG, groups)), dim3(64 * BIG_NW), 0, s, r, groups, (uint32_t)nb, DEV_LIST);
            tm.reset();
          }
          if (timed) tm.emplace("replay_update", 0);
          hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)groups * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          tm.reset();
          count_dirty();
        }
      }
      // file what the clean buckets need (always safe), then one round trip for everything: dirty count, range, requests
      {
        std::optional<KernelTimer> tmf;
        if (timed_misc) tmf.emplace("replay_misc", 0);
        hipLaunchKernelGGL(k_file, dim3(cdiv256(nb)), dim3(256), 0, s, r, (uint32_t)nb);
      }
      fetch(trace);
      if (hc->overflow) goto overflowed;
      n_dirty = hc->ndirty, known = true;
      r.memo_used = hc->nreq != 0;
      d_lo = n_dirty ? hc->min_dirty : 0, d_hi = n_dirty ? hc->max_dirty + 1 : 0;
      if (trace)
        fprintf(stderr, "[pgx]   round %u: %zu dirty left in [%u, %u), %llu evaluations, t = +%.2f ms\n", rounds, n_dirty, d_lo, d_hi,
                (unsigned long long)hc->evals, now_ms() - t0);
      if (!n_dirty) break;
      if (rounds > 20000) {  // (cannot happen: the lowest unstable bucket rises every pass)
        hc->overflow |= OV_PASSES;
        goto overflowed;
      }
    }
    rounds_total += rounds;
    const size_t nreq = hc->nreq;
    if (trace)
      fprintf(stderr, "[pgx] device sweep %u: %u rounds, %llu evaluations so far, %.2f ms, %zu requests\n", sweeps, rounds,
              (unsigned long long)hc->evals, now_ms() - p0, nreq - first_req);
    if (nreq == first_req) break;
    const double a0 = now_ms();
    const size_t batch = nreq - first_req;
    if (sweeps == 1) first_batch = batch;
    r.tail = tail_max && batch <= std::max(tail_max, first_batch / 256) ? ahead : 0u;   // (the NEXT sweep's k_file)
    dev_align(db, r.rq_key + first_req, batch, band, r.rq_res + first_req, sweeps > 2 ? 2 : sweeps > 1 ? 1 : 0);
    r.settled = (uint32_t)nreq;
    first_req = nreq;
    {
      std::optional<KernelTimer> tms;
      if (timed_misc) tms.emplace("replay_misc", 0);
      hipLaunchKernelGGL(k_settle, dim3(cdiv256(nb)), dim3(256), 0, s, r);
    }
    count_dirty();
    if (batch > 100000) {  // a big batch: worth a round trip to know how many guesses were wrong (dense or sparse next)
      fetch(false);
      if (hc->overflow) goto overflowed;
      n_dirty = hc->ndirty, known = true;
      d_lo = n_dirty ? hc->min_dirty : 0, d_hi = n_dirty ? hc->max_dirty + 1 : 0;
      if (trace) fprintf(stderr, "[pgx]   alignments + settle %.2f ms, %zu buckets guessed wrong\n", now_ms() - a0, n_dirty);
      if (!n_dirty) break;
    } else {
      known = false;  // (a small batch: the wrong guesses are handled by the sparse passes of the next round)
      n_dirty = std::min<size_t>(batch, SPARSE_CAP / 8);
      if (trace) fprintf(stderr, "[pgx]   %zu alignments + settle enqueued in %.2f ms\n", batch, now_ms() - a0);
    }
    align_ms += now_ms() - a0;
  }
  {
    const double e0 = now_ms();
    DevBuf<uint32_t> off(nb + 1);
    size_t tb = 0;
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, r.inum, off.p, (int)nb, s));
    DevBuf<uint8_t> tmp(tb + 256);
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, r.inum, off.p, (int)nb, s));
    uint32_t last_off = 0, last_num = 0;
    PGX_HIP(hipMemcpyAsync(&last_off, off.p + nb - 1, 4, hipMemcpyDeviceToHost, s));
    PGX_HIP(hipMemcpyAsync(&last_num, r.inum + nb - 1, 4, hipMemcpyDeviceToHost, s));
    sync();
    const size_t nrec = (size_t)last_off + last_num;
    // (the table census for the next stage's sizes runs HERE, ahead of the record copy: a memory-bound kernel beside the copy's blit kernel
    //  crawls -- 40 ms instead of 0.5 for these 2 GB, profiles/r05e_chunk_timeline_c4.txt)
    DevBuf<unsigned long long> d_keys(1);
    PGX_HIP(hipMemsetAsync(d_keys.p, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_count_pairs, dim3((unsigned)std::min<size_t>(cdiv256(pcap), 4096)), dim3(256), 0, s, ph.p, pcap, d_keys.p);
    results_wait();   // (pgx_results_async: the previous stage's record copy -- long finished -- gives its device buffer back first)
    pgx_ovlp *host = alloc_out(nrec);
    DevBuf<pgx_ovlp> d_out;
    {
      MemTag t7("replay.records_out");
      d_out.alloc(std::max<size_t>(nrec, 1));
    }
    {
      std::optional<KernelTimer> tme;
      if (timed_misc) tme.emplace("replay_emit", nrec);   // the records written and brought to the host (pinned destination)
      hipLaunchKernelGGL(k_emit, dim3(cdiv256(nb)), dim3(256), 0, s, r, off.p, d_out.p);
      // pgx_results_async: the copy runs on its own stream behind k_emit and this call returns without it -- the 2.9 GB of a human-scale
      // chunk (55 ms over PCIe) overlap the NEXT chunk's join and first sweep; the caller waits (pgx_results_wait) before it reads
      if (nrec && record_sink()) record_sink()->take(std::move(d_out), nrec);   // (served commands: device -> output file, pgx_served.cpp)
      else if (nrec && results_async() && !timed_misc) results_copy_async(host, std::move(d_out), nrec);
      else if (nrec) PGX_HIP(hipMemcpyAsync(host, d_out.p, nrec * sizeof(pgx_ovlp), hipMemcpyDeviceToHost, s));
    }
    unsigned long long n_keys = 0;
    d_keys.download(&n_keys, 1);
    if (!read_counters(false)) goto overflowed;
    // what this stage used, per bucket entry: read pairs, requests (= memo entries), items, reader nodes
    usage[0] = (double)n_keys / ne, usage[1] = (double)hc->nreq / ne, usage[2] = (double)hc->item_top / ne, usage[3] = (double)hc->rnode_top / ne;
    if (trace)
      fprintf(stderr, "[pgx]   tables: %llu read pairs in %u slots (load %.2f), %u alignments in %u memo slots (load %.2f), items %u of %u, reader nodes %u of %u, requests %u of %u\n",
              n_keys, pcap, (double)n_keys / pcap, hc->nreq, mcap, (double)hc->nreq / mcap, hc->item_top, r.item_cap, hc->rnode_top, r.rn_cap, hc->nreq, r.req_cap);
    *n_out = nrec;
    if (st) {
      st->n_align_needed = hc->lookups, st->n_seen_skip = hc->skips, st->n_align_gpu = first_req;
      st->rounds = sweeps;
      st->n_evaluations = hc->evals;
#if !defined(PGX_BIG_STATS) && !defined(PGX_SETTLE_STATS)
      st->stream_checksum = hc->records;
#endif
    }
    if (trace)
      fprintf(stderr, "[pgx] device replay: %u sweeps, %u rounds, %llu evaluations, %zu records; emit %.2f ms; alignments %.2f ms; total %.2f ms\n",
              sweeps, rounds_total, (unsigned long long)hc->evals, nrec, now_ms() - e0, align_ms, now_ms() - t0);
  }
  return 0;
overflowed:
  if (trace || !(hc->overflow & (OV_ITEMS | OV_NODES | OV_REQS | OV_PAIRS | OV_MEMO)))
    fprintf(stderr, "[pgx] note: the device replay gave up (code %u: items %u of %u, reader nodes %u of %u, requests %u of %u)\n", hc->overflow,
            hc->item_top, r.item_cap, hc->rnode_top, r.rn_cap, hc->nreq, r.req_cap);
  return hc->overflow ? hc->overflow : OV_PASSES;
}
}  // namespace

// the visit list on the device from the per-group slices the host's table replay left (pgx_overlap.cpp::build_visit, ids-only form)
namespace {
__global__ void k_place_bids(const uint32_t *__restrict__ ids_all, const uint32_t *__restrict__ psrc, const uint32_t *__restrict__ pcnt,
                             const uint64_t *__restrict__ pdst, uint32_t n_groups, uint32_t *__restrict__ bid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups) return;
  const uint32_t src = psrc[i], n = pcnt[i];
  const uint64_t dst = pdst[i];
  for (uint32_t k = 0; k < n; ++k) bid[dst + k] = ids_all[src + k];
}
}  // namespace
void dev_place_bids(const uint32_t *ids_all, size_t n_ids, const uint32_t *psrc, const uint32_t *pcnt, const uint64_t *pdst,
                    size_t n_groups, size_t nb, DevBuf<uint32_t> &bid) {
  bid.alloc(std::max<size_t>(nb, 1));
  if (!n_groups || !nb) return;
  DevBuf<uint32_t> d_ids(n_ids), d_src(n_groups), d_cnt(n_groups);
  DevBuf<uint64_t> d_dst(n_groups);
  d_ids.upload(ids_all, n_ids), d_src.upload(psrc, n_groups), d_cnt.upload(pcnt, n_groups), d_dst.upload(pdst, n_groups);
  hipLaunchKernelGGL(k_place_bids, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, ctx().stream, d_ids.p, d_src.p, d_cnt.p, d_dst.p,
                     (uint32_t)n_groups, bid.p);
  sync();   // (the upload sources are the caller's host arrays; the temporaries go back to the block cache)
}

namespace {
double g_learned[4] = {0, 0, 0, 0};
ShutdownHook h_learn([] { replay_forget_sizes(); });
}  // namespace
void replay_forget_sizes() {
  for (double &m : g_learned) m = 0;
}

bool dev_replay(const pgx_seqdb *db, const DevicePairs &dp, const uint32_t *visit_bids, const uint32_t *d_bids, size_t nb, size_t n_entries,
                uint32_t bestn, int band, bool predict, uint32_t ovlp_upper, const std::function<pgx_ovlp *(size_t)> &alloc_out,
                size_t *n_out, pgx_overlap_stats *st, bool trace) {
  *n_out = 0;
  // what the encodings hold (anything else goes to the host replay)
  if (ovlp_upper > 128 || nb >= (1u << 29) - 2 || n_entries >= (1ULL << 31) || !dp.valid) return false;
  if (nb == 0) {
    alloc_out(0);
    return true;
  }
  {  // the device tables must fit beside what is already resident (otherwise the host replay, which only needs host memory)
    const size_t ne = std::max<size_t>(n_entries, 1024);
    const size_t need = (size_t)pow2_at_least(ne) * (sizeof(PHot) + sizeof(PCold) + sizeof(MSlot)) + ne * (6 * sizeof(Item) + 8 * sizeof(RNode) + sizeof(pgx_align_key) + sizeof(pgx_match)) +
                        nb * (size_t)(64 * sizeof(Item) + 64) + (256u << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b) {
      dev_cache_trim();  // (blocks the cache holds for re-use count as used)
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b) {
        fprintf(stderr, "[pgx] note: the device replay's tables (%.1f GB) do not fit the free device memory (%.1f GB); the host replay takes over\n",
                need / 1e9, free_b / 1e9);
        return false;
      }
    }
  }
  // Table sizes are multiples of the defaults (items 6 x, reader nodes 8 x, requests 1 x the bucket entries; pair and memo table: the power of
  // two from the entries up).  Two things were found at full-size configs[3] (round 5, profiles/r05a_chunk_timeline_c4.txt):
  //  * every one of the 8 chunks of a step ran two attempts in vain -- requests 1.72 x the entries, then the memo table -- and an
  //    overflowing attempt costs a whole first sweep (the request / memo overflow is only seen when k_file runs): 205 of 1,103 ms;
  //  * the open-addressing tables want a LOW load: with the pair table at 0.75 (64 M slots for 50 M read pairs) and the memo table at 0.71
  //    the evaluations of a chunk took 73 ms longer than at 0.37 / 0.36 (linear probing: ~8 probes per miss instead of ~2).
  // So a stage measures what it used per bucket entry -- read pairs, requests, items, reader nodes -- and the NEXT stage of the process (the
  // chunks of a job are alike) sizes its tables by that: hash tables for a load of at most 0.4, arenas with 20 % to spare.  The first stage of
  // a process starts from the defaults and, where they overflow, repeats with x 4 hash tables / x 2 arenas.
  double *learned = g_learned;   // read pairs, requests, items, reader nodes per bucket entry (0: not known)
  double mult[5] = {1, 1, 1, 1, 1};
  if (learned[0] > 0) {
    // (the arenas may also SHRINK to what the last stage used + 25-100 %: 8 reader nodes per entry are reserved by default and a c4 chunk links
    //  25 thousand of its 213 million; a stage that needs more than that repeats once and the next one knows)
    const double items_default = 6.0 + 64.0 * (double)nb / std::max<size_t>(n_entries, 1024);
    mult[0] = std::max(0.2, learned[2] * 1.25 / items_default), mult[1] = std::max(0.02, learned[3] * 2 / 8), mult[2] = std::max(1.0, learned[1] * 1.2);
    mult[3] = std::max(1.0, learned[0] / 0.4), mult[4] = std::max(1.0, learned[1] / 0.4);
  }
  if (getenv("PGX_REPLAY_PAIRS_X")) mult[3] = atof(getenv("PGX_REPLAY_PAIRS_X"));
  if (getenv("PGX_REPLAY_MEMO_X")) mult[4] = atof(getenv("PGX_REPLAY_MEMO_X"));
  for (int attempt = 0; attempt < 4; ++attempt) {
    double usage[4] = {0, 0, 0, 0};
    uint32_t ov;
    try {
      ov = replay_attempt(db, dp, visit_bids, d_bids, nb, n_entries, bestn, band, predict, alloc_out, n_out, st, trace, mult, usage);
    } catch (const Fail &f) {
      //  tables sized by what an EARLIER stage used -- possibly another job's in a long-lived server -- may not fit where the
      // defaults would: once more with the defaults and the memory forgotten, else the host replay (which only needs host memory)
      if (f.code != PGX_ENOMEM && !(f.code == PGX_EHIP && strstr(pgx_last_error(), "hipMalloc"))) throw;   // (only a failed allocation)
      (void)hipGetLastError();
      const bool had_learned = learned[0] > 0;
      replay_forget_sizes();
      dev_cache_trim();
      if (!had_learned || attempt > 0) {
        fprintf(stderr, "[pgx] note: the device replay's tables could not be allocated (%s); the host replay takes over\n", pgx_last_error());
        return false;
      }
      for (double &m : mult) m = 1;
      if (trace) fprintf(stderr, "[pgx]   the tables sized by the last stage's use do not fit (%s): once more with the default sizes\n", pgx_last_error());
      continue;
    }
    if (!ov) {
      for (int k = 0; k < 4; ++k) learned[k] = std::max(learned[k], usage[k]);
      if (st) st->replay_attempts = (uint32_t)attempt + 1;
      return true;
    }
    if (ov & (OV_QOFF | OV_PASSES)) break;  // not a matter of table sizes
    // unusual data (repeat-rich sets): the same walk again with larger tables.  Requests and memo entries are the same alignments: when the
    // request array was too small the memo table of the same size class is too, and k_file stopped before it could say so -- grow both
    for (int k = 0; k < 3; ++k)
      if (ov & (1u << k)) mult[k] = mult[k] < 1 ? 1.0 : mult[k] * 2;   // (an arena that was shrunk to the last stage's use goes back to the default first)
    if (ov & OV_PAIRS) mult[3] *= 4;
    if (ov & (OV_MEMO | OV_REQS)) mult[4] *= 4;
    if (trace) fprintf(stderr, "[pgx]   next attempt with items x %g, reader nodes x %g, requests x %g, pair table x %g, memo table x %g\n", mult[0], mult[1], mult[2], mult[3], mult[4]);
  }
  fprintf(stderr, "[pgx] note: the device replay's tables overflowed; the host replay takes over\n");
  return false;
}

}  // namespace pgx
