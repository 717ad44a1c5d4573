// pgx_khash_dev.hip -- klib-khash's slot layout for a sequence of DISTINCT keys, computed on the device (round 3).
//
// The overlap stage visits the first-key groups in the slot order of the reference's outer khash table
// (/root/reference/src/shmr_overlap.c:206-215; table behaviour: src/khash.h:232-336, hash :373, load factor :180).  The host form
// (DistinctSlotTable, pgx_khash.h) replays the puts and the resizes one after the other on one thread: 6.6 ms for the 0.75 M first
// keys of a 4.5-Gbase chunk -- hidden behind the join -- but 126 ms for the 4.4 M of the `-l 1` stress shape, which the GPU then
// waits for.  Both halves of the process have a parallel form:
//
//  * BETWEEN two resizes nothing is deleted and every put takes the first free slot of its probe sequence (home, +1, +3, +6, ...):
//    the layout is that of a PRIORITY insertion, priority = insertion index (a slot ends up with the earliest-inserted key among
//    those whose sequence reaches it before finding a place) -- history-independent, hence computable in any order: every key walks
//    its sequence, takes a slot from a later key with a compare-and-swap and then carries that key on from where it sat.
//  * A RESIZE (kh_resize, khash.h:258-284) walks the old slots in index order, places each element at the first slot of its new
//    sequence that no PLACED element holds, and when an element that has not been moved yet sits there, evicts it and places it
//    next.  So the new layout is again a priority insertion, the priorities being the placement order: rank (j, 0) for the element
//    the scan finds at old slot j, rank (root, depth + 1) for an element evicted by one of rank (root, depth).  The ranks depend on
//    where elements land, which depends on the ranks: a fixed point.  Start from "nobody is evicted" (rank = old slot), insert by
//    rank, re-derive every rank from who landed on whose old slot before the scan got there, repeat until no rank changes.  The
//    sequential process satisfies the fixed-point equations and, by induction over the placement order, is their only solution, so a
//    converged state IS khash's layout.  Evictions that change anything are rare (an element must land beyond its own old slot on
//    one the scan has not reached): 1-10 iterations in practice; beyond DEV_KHASH_MAX_ITER the caller falls back to the host form.
//
// STATUS: exact (tests/test_gpu_khash.py: against the oracle's literal replay) and NOT the stage's default (PGX_DEV_OUTER_MIN = number of
// first keys from which it is used; unset: never).  Measured on the c5s shape (4.39 M first keys, 8.4 M slots): 11 s against the host
// thread's 0.126 s.  The reference's keys are hash << 8 | span with span = 16 throughout, so khash's integer hash gives ALL keys
// 1 / 256 of the slots as homes -- ~130 keys per home, probe chains of hundreds to thousands of steps.  The host form does not walk
// them (a skip count per home: the next put of that home resumes where the last one ended); the lanes here do, 14-19 times per resize.
// The parallel counterpart of the skip count is known -- an element may start at step r, r = the number of higher-priority elements
// with the same home (they all sit earlier in the same sequence) -- and needs a sort by (home, priority) per pass; with synthetic
// keys whose spans vary (chains of ~10) the device form is on a par with the host (4.4 M keys: 327 vs 296 ms; 1.5 M: 40 vs 50 ms).
//
// Slots hold id + 1 (0 = empty); `step[id]` = number of probe increments the element made to get where it is, written by the lane
// that carries the element BEFORE the compare-and-swap that publishes it.
#include <chrono>

#include "pgx_internal.h"
#include "pgx_khash.h"

namespace pgx {

namespace {
constexpr int DEV_KHASH_MAX_ITER = 64;
constexpr uint32_t PROBE_BUDGET = 1u << 22;   // probes one lane may spend in one launch (a degenerate key set: give up, the host form has the chain-skipping trick)

__device__ __forceinline__ uint32_t kh32(uint64_t k) { return (uint32_t)(k >> 33 ^ k ^ k << 11); }

__global__ void k_kh_hash(const uint64_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ hv) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hv[i] = kh32(keys[i]);
}

// Priority insertion of the elements listed in `who` (RANKED: ids + 1 from the old table's slots, 0 = none) or of ids [lo, hi)
// into tab[0 .. m].  Priority: RANKED ? (rank[id], id) : id -- smaller first.  Elements already in the table of an un-RANKED
// launch have smaller ids than every new one, i.e. they are never displaced.
template <bool RANKED>
__global__ __launch_bounds__(256) void k_kh_insert(const uint32_t *__restrict__ hv, uint32_t *__restrict__ tab, uint32_t m,
                                                   uint32_t *__restrict__ step, const uint32_t *__restrict__ who, uint32_t lo, uint32_t hi,
                                                   const unsigned long long *__restrict__ rank, uint32_t *__restrict__ flags) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x;
  if (RANKED) {
    if (t >= hi) return;
    const uint32_t w = who[t];
    if (!w) return;
    x = w - 1;
  } else {
    x = lo + t;
    if (x >= hi) return;
  }
  uint32_t st = 0, p = hv[x] & m, budget = PROBE_BUDGET;
  unsigned long long rx = RANKED ? rank[x] : 0ull;
  for (;;) {
    if (--budget == 0) {
      atomicOr(flags + 1, 1u);   // (sticky: read with the next round trip)
      return;
    }
    uint32_t cur = __hip_atomic_load(&tab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool mine;   // does x take this slot from its occupant?
    if (cur == 0) {
      mine = true;
    } else {
      const uint32_t c = cur - 1;
      if (RANKED) {
        const unsigned long long rc = rank[c];
        mine = rx < rc || (rx == rc && x < c);
      } else {
        mine = x < c;
      }
    }
    if (!mine) {
      ++st;
      p = (p + st) & m;
      continue;
    }
    step[x] = st;
    __threadfence();   // (the step is visible before the slot says the element is there)
    const uint32_t old = atomicCAS(&tab[p], cur, x + 1);
    if (old != cur) continue;   // somebody else changed the slot: look at it again
    if (cur == 0) return;
    // carry the displaced element on from here
    __threadfence();
    x = cur - 1;
    st = __hip_atomic_load(&step[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (RANKED) rx = rank[x];
    ++st;
    p = (p + st) & m;
  }
}

// the old table's view of a resize: rank (slot, 0) for every element, to start with
__global__ void k_kh_rank0(const uint32_t *__restrict__ oldtab, uint32_t S, unsigned long long *__restrict__ rank) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const uint32_t y = oldtab[j];
  if (y) rank[y - 1] = (unsigned long long)j << 32;
}
// the ranks that follow from a layout: the element of old slot j was evicted iff the element that landed on slot j was placed before
// the scan reached j.  flags[0]: some rank changed (flags[1]: a lane ran out of its probe budget).
__global__ void k_kh_rerank(const uint32_t *__restrict__ oldtab, const uint32_t *__restrict__ newtab, uint32_t S,
                            const unsigned long long *__restrict__ rank, unsigned long long *__restrict__ nrank, uint32_t *__restrict__ flags) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const uint32_t y = oldtab[j];
  if (!y) return;
  const uint32_t w = newtab[j];
  unsigned long long nr = (unsigned long long)j << 32;
  if (w && w != y) {
    const unsigned long long rw = rank[w - 1];
    if (rw < nr) nr = rw + 1;
  }
  if (nr != rank[y - 1]) atomicOr(flags, 1u);
  nrank[y - 1] = nr;
}
__global__ void k_kh_slots64(const uint32_t *__restrict__ tab, const uint32_t *__restrict__ hv, uint32_t S, uint64_t *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= S) return;
  const uint32_t y = tab[j];   // (the word DistinctSlotTable keeps: hash | state | id)
  out[j] = y ? ((uint64_t)hv[y - 1] << 32) | (1ull << 30) | (uint64_t)(y - 1) : 0ull;
}
inline uint32_t upper_of(uint32_t nn) { return (uint32_t)(nn * 0.77 + 0.5); }
inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// The first keys -- up to a table of SEED_SLOTS -- are replayed on the HOST and the table uploaded: a small, full table wraps its
// probe sequences around many times, evictions interact, and the fixed point of such a resize can take as many iterations as the
// table has elements (measured: 128 -> 256 slots 30 to > 64 iterations; 65,536 -> 131,072 four; beyond that one or two), while
// the sequential replay of 50 k keys is 0.3 ms.
constexpr uint32_t SEED_SLOTS = 1u << 16;
}  // namespace

// d_keys: n distinct keys in insertion order (device).  touch: a put of a present key follows the last insertion (it only runs the
// load check, khash.h:298-306).  Leaves the table as 8-byte words (0 = empty, else hash32 << 32 | 1 << 30 | id) in `slots`,
// *n_slots entries.  Returns false when the computation gave up (no convergence, a degenerate probe chain): the host form decides.
bool dev_khash_slots(const uint64_t *d_keys, size_t n_keys, bool touch, DevBuf<uint64_t> &slots, uint32_t *n_slots) {
  *n_slots = 0;
  if (n_keys == 0 || n_keys >= (1u << 30)) return false;
  const uint32_t n = (uint32_t)n_keys;
  hipStream_t st = ctx().stream;
  KernelTimer tm("visit", 0);
  const bool trace = getenv("PGX_TRACE") != nullptr;
  const double t_begin = trace ? (sync(), wall_ms()) : 0.0;
  uint32_t cap = 4;
  while (upper_of(cap) <= n && cap < (1u << 31)) cap <<= 1;
  if (cap >= (1u << 31)) return false;
  cap *= 2;   // (a trailing touch may resize once more)
  DevBuf<uint32_t> hv(n), step(n), tabA(cap), tabB(cap), flags(2);
  PGX_HIP(hipMemsetAsync(flags.p, 0, 2 * sizeof(uint32_t), st));
  DevBuf<unsigned long long> rankA(n), rankB(n);
  hipLaunchKernelGGL(k_kh_hash, dim3((n + 255) / 256), dim3(256), 0, st, d_keys, n, hv.p);
  uint32_t *tab = tabA.p, *other = tabB.p;
  unsigned long long *rank = rankA.p, *nrank = rankB.p;
  uint32_t S = 4, count = 0;
  {
    const uint32_t seed = std::min(n, upper_of(SEED_SLOTS));
    std::vector<uint64_t> hk(seed);
    PGX_HIP(hipMemcpyAsync(hk.data(), d_keys, (size_t)seed * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    sync();
    DistinctSlotTable t;
    for (uint32_t i = 0; i < seed; ++i) t.put_new(hk[i], i);
    std::vector<uint32_t> ht(t.nb);
    for (uint32_t s0 = 0; s0 < t.nb; ++s0) ht[s0] = t.is_used(s0) ? t.id_at(s0) + 1 : 0u;
    S = t.nb, count = seed;
    PGX_HIP(hipMemcpyAsync(tab, ht.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    sync();   // (ht goes out of scope)
  }
  uint32_t hflags[2] = {0, 0};
  auto resize = [&]() -> bool {   // S -> 2 S with `count` elements in `tab`
    const uint32_t N = 2 * S;
    hipLaunchKernelGGL(k_kh_rank0, dim3((S + 255) / 256), dim3(256), 0, st, tab, S, rank);
    for (int it = 0;; ++it) {
      if (it >= DEV_KHASH_MAX_ITER) {
        if (trace) fprintf(stderr, "[pgx]   device khash: resize %u -> %u not converged after %d iterations\n", S, N, it);
        return false;
      }
      PGX_HIP(hipMemsetAsync(other, 0, (size_t)N * sizeof(uint32_t), st));
      PGX_HIP(hipMemsetAsync(flags.p, 0, sizeof(uint32_t), st));
      hipLaunchKernelGGL((k_kh_insert<true>), dim3((S + 255) / 256), dim3(256), 0, st, hv.p, other, N - 1, step.p, tab, 0u, S, rank, flags.p);
      hipLaunchKernelGGL(k_kh_rerank, dim3((S + 255) / 256), dim3(256), 0, st, tab, other, S, rank, nrank, flags.p);
      PGX_HIP(hipMemcpyAsync(hflags, flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      sync();
      if (hflags[1]) {
        if (trace) fprintf(stderr, "[pgx]   device khash: a lane ran out of its probe budget (resize %u -> %u)\n", S, N);
        return false;
      }
      if (!hflags[0]) {
        if (trace && (it > 1 || S >= (1u << 18))) fprintf(stderr, "[pgx]   device khash: resize %u -> %u converged after %d iteration(s)\n", S, N, it + 1);
        break;
      }
      std::swap(rank, nrank);
    }
    std::swap(tab, other);
    S = N;
    return true;
  };
  while (count < n) {
    const uint32_t hi = std::min(n, upper_of(S));
    if (hi > count) {
      hipLaunchKernelGGL((k_kh_insert<false>), dim3((hi - count + 255) / 256), dim3(256), 0, st, hv.p, tab, S - 1, step.p,
                         (const uint32_t *)nullptr, count, hi, (const unsigned long long *)nullptr, flags.p);
      count = hi;
      if (count == n) {   // (the last batch: no resize behind it reads the flags)
        PGX_HIP(hipMemcpyAsync(hflags, flags.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        sync();
        if (hflags[1]) return false;
      }
    }
    if (count < n && !resize()) return false;
  }
  if (touch && count >= upper_of(S) && !resize()) return false;
  slots.alloc(S);
  hipLaunchKernelGGL(k_kh_slots64, dim3((S + 255) / 256), dim3(256), 0, st, tab, hv.p, S, slots.p);
  PGX_HIP(hipGetLastError());
  sync();   // (the temporaries go back to the block cache)
  if (trace) fprintf(stderr, "[pgx]   device khash: %u keys -> %u slots in %.2f ms\n", n, S, wall_ms() - t_begin);
  *n_slots = S;
  return true;
}

}  // namespace pgx
