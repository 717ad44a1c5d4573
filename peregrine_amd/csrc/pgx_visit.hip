// pgx_visit.hip -- the bucket visit order of the overlap stage, built on the device (round 3).
//
// shmr_overlap walks a two-level klib-khash map in slot order (/root/reference/src/shmr_overlap.c:206-215 over the map build_map
// fills, src/shmr_utils.c:295-404): the OUTER table is keyed by the first shimmer of a pair, every outer entry owns an INNER table
// keyed by the second.  The slot layout of a khash table is a function of the order in which its distinct keys were first
// inserted (src/khash.h:232-336; see pgx_khash.h), so the walk is reproduced by replaying the insertions:
//   * the outer table -- one table, 0.7 M keys at 4.5 Gbases, inherently sequential -- stays on the host (DistinctSlotTable), started
//     while the join's sorts still run;
//   * the inner tables -- 0.7 M independent little tables, 5 buckets on average -- were replayed by 48 host threads from tables the
//     join had to download first (2 ms of copies + 5.2 ms of replay + 1.8 ms of slot scan with the GPU idle, and 4x that when eight
//     ranks share the host's cores).  Here a LANE replays the table of a group of up to 48 buckets (64 slots of 16 bits in LDS), a
//     WAVEFRONT the table of a larger one (4,096 slots, 64 probe positions per step), each leaving the group's visited bucket
//     ids in slot order; when the host's outer table arrives (8 bytes per slot, pinned) a scan over its slots puts the groups'
//     lists in their final places.  Groups beyond 3,153 buckets send the stage back to the host path (pairs_fetch_tables).
// Slot word: hash bits | 2-bit state | id, as in DistinctSlotTable: the keys of a table are distinct and never compared, a resize
// needs only the low bits of the hash, and "occupied" alternates between 1 and 2 from one resize to the next, which tells
// "not moved yet" from "placed" during a rehash (khash.h:258-284 does that with two flag arrays).
#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {

namespace {
__device__ __forceinline__ uint32_t kh32(uint64_t k) { return (uint32_t)(k >> 33 ^ k ^ k << 11); }               // khash.h:373
__device__ __forceinline__ uint32_t upper_of(uint32_t nn) { return (uint32_t)(nn * 0.77 + 0.5); }                // khash.h:180, :257

constexpr int LB = 256;        // groups (lanes) per block of the lane kernel
constexpr int LANE_SLOTS = 64, WAVE_SLOTS = 4096;

// a lane's 64 slots, dword-interleaved over the block (slot pair s, s^1 of lane t in dword (s/2) * LB + t: no bank conflicts
// between lanes that are at the same slot, whatever the others do)
struct LaneTab {
  uint16_t *base;
  __device__ __forceinline__ uint32_t get(uint32_t s) const { return base[(s >> 1) * (2 * LB) + (s & 1)]; }
  __device__ __forceinline__ void set(uint32_t s, uint32_t v) const { base[(s >> 1) * (2 * LB) + (s & 1)] = (uint16_t)v; }
};

// the insertions of one group, by one lane: keys k1[0..n) in first-insertion order, then (trail) one more put of a present key,
// which only runs the load check (khash.h:298-306).  Returns the final number of slots.
template <int IB, int HB, typename Tab>
__device__ __forceinline__ uint32_t replay_group(const Tab &tab, const uint64_t *__restrict__ k1, uint32_t n, bool trail) {
  constexpr int HS = IB + 2;
  constexpr uint32_t ST = 3u << IB, HM = (1u << HB) - 1;
  uint32_t nb = 0, size = 0, upper = 0, live = 1;
  for (uint32_t i = 0; i <= n; ++i) {
    if (i == n && !trail) break;
    if (size >= upper) {  // kh_resize to twice the slots: every element moves, an element found in the way is carried on
      const uint32_t nn = nb ? nb * 2 : 4;
      for (uint32_t s = nb; s < nn; ++s) tab.set(s, 0);
      const uint32_t m = nn - 1, old = live << IB, nw = (live ^ 3u) << IB;
      for (uint32_t j = 0; j < nb; ++j) {
        uint32_t e = tab.get(j);
        if ((e & ST) != old) continue;
        tab.set(j, 0);
        for (;;) {
          uint32_t p = (e >> HS) & m, step = 0, prev;
          while (((prev = tab.get(p)) & ST) == nw) p = (p + (++step)) & m;
          tab.set(p, (e & ~ST) | nw);
          if ((prev & ST) != old) break;
          e = prev;
        }
      }
      nb = nn, upper = upper_of(nn), live ^= 3u;
    }
    if (i == n) break;
    const uint32_t hv = kh32(k1[i]), m = nb - 1;
    uint32_t p = hv & m, step = 0;
    while (tab.get(p)) p = (p + (++step)) & m;
    tab.set(p, ((hv & HM) << HS) | (live << IB) | i);
    ++size;
  }
  return nb;
}

__global__ __launch_bounds__(LB) void k_inner_lane(const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gbucket,
                                                   const uint8_t *__restrict__ gtrail, const uint64_t *__restrict__ k1,
                                                   const uint32_t *__restrict__ bsz, const uint32_t *__restrict__ bord, uint32_t ng,
                                                   uint32_t upper, uint32_t *__restrict__ ids_all, uint32_t *__restrict__ gnb,
                                                   unsigned long long *__restrict__ tot) {
  __shared__ uint16_t lds[LANE_SLOTS * LB];
  const uint32_t g = blockIdx.x * LB + threadIdx.x;
  unsigned long long ne = 0;
  if (g < ng) {
    const uint32_t b0 = gbucket[g], n = gbucket[g + 1] - b0;
    if (n <= VISIT_LANE_MAX) {
      uint32_t cnt = 0;
      if (gstart[g + 1] - gstart[g] > 2) {   // (else no bucket of this group can hold more than 2 records: nothing to visit)
        const LaneTab tab{lds + 2 * threadIdx.x};
        const uint32_t nb = replay_group<6, 6>(tab, k1 + b0, n, gtrail[g] != 0);
        for (uint32_t s = 0; s < nb; ++s) {
          const uint32_t e = tab.get(s);
          if (!e) continue;
          const uint32_t id = e & 63u, bn = bsz[b0 + id];
          if (bn > 2 && bn <= upper) ids_all[b0 + cnt++] = bord[b0 + id], ne += bn;   // shmr_overlap.c:216
        }
      }
      gnb[g] = cnt;
    }
  }
  for (int o = 32; o; o >>= 1) ne += __shfl_xor(ne, o, 64);
  if ((threadIdx.x & 63) == 0 && ne) atomicAdd(tot, ne);
}

// a wavefront per large group: the same replay with the probe sequence examined 64 positions at a time
__global__ __launch_bounds__(64) void k_inner_wave(const uint32_t *__restrict__ big, const uint32_t *__restrict__ gbucket,
                                                   const uint8_t *__restrict__ gtrail, const uint64_t *__restrict__ k1,
                                                   const uint32_t *__restrict__ bsz, const uint32_t *__restrict__ bord, uint32_t upper,
                                                   uint32_t *__restrict__ ids_all, uint32_t *__restrict__ gnb,
                                                   unsigned long long *__restrict__ tot) {
  constexpr int IB = 12, HS = 14;
  constexpr uint32_t ST = 3u << IB, HM = WAVE_SLOTS - 1, IDM = (1u << IB) - 1;
  __shared__ uint32_t tab[WAVE_SLOTS];
  const int lane = threadIdx.x;
  const uint32_t g = big[blockIdx.x];
  const uint32_t b0 = gbucket[g], n = gbucket[g + 1] - b0;
  const bool trail = gtrail[g] != 0;
  // first position of home's probe sequence (home, +1, +3, +6, ...) whose slot is not `busy`; state == 0: busy = any occupant
  auto first_free = [&](uint32_t home, uint32_t m, uint32_t state) {
    for (uint32_t s0 = 0;; s0 += 64) {
      const uint32_t st = s0 + (uint32_t)lane;
      const uint32_t p = (home + ((st * (st + 1)) >> 1)) & m;
      const uint32_t v = tab[p];
      const uint64_t fr = __ballot(state ? (v & ST) != state : v == 0);
      if (fr) return (uint32_t)__shfl((int)p, __builtin_ctzll(fr), 64);
    }
  };
  uint32_t nb = 0, size = 0, upper_b = 0, live = 1;
  for (uint32_t i0 = 0; i0 <= n; i0 += 64) {
    const uint32_t mine = i0 + lane < n ? kh32(k1[b0 + i0 + lane]) : 0u;
    for (uint32_t l = 0; l < 64 && i0 + l <= n; ++l) {
      const uint32_t i = i0 + l;
      if (i == n && !trail) break;
      if (size >= upper_b) {
        const uint32_t nn = nb ? nb * 2 : 4;
        for (uint32_t s = nb + lane; s < nn; s += 64) tab[s] = 0;
        __syncthreads();
        const uint32_t m = nn - 1, old = live << IB, nw = (live ^ 3u) << IB;
        for (uint32_t j0 = 0; j0 < nb; j0 += 64) {
          uint32_t from = 0;   // lanes below `from` are done in this batch
          for (;;) {
            const uint32_t v = j0 + lane < nb ? tab[j0 + lane] : 0u;
            const uint64_t todo = __ballot((v & ST) == old && (uint32_t)lane >= from);
            if (!todo) break;
            const int jl = __builtin_ctzll(todo);
            uint32_t e = (uint32_t)__shfl((int)v, jl, 64);
            from = (uint32_t)jl + 1;
            if (lane == 0) tab[j0 + jl] = 0;
            __syncthreads();
            for (;;) {
              const uint32_t p = first_free((e >> HS) & m, m, nw);
              const uint32_t prev = tab[p];
              __syncthreads();
              if (lane == 0) tab[p] = (e & ~ST) | nw;
              __syncthreads();
              if ((prev & ST) != old) break;
              e = prev;
            }
          }
        }
        nb = nn, upper_b = upper_of(nn), live ^= 3u;
      }
      if (i == n) break;
      const uint32_t hv = (uint32_t)__shfl((int)mine, (int)l, 64);
      const uint32_t p = first_free(hv & (nb - 1), nb - 1, 0u);
      if (lane == 0) tab[p] = ((hv & HM) << HS) | (live << IB) | i;
      __syncthreads();
      ++size;
    }
  }
  // the visited buckets in slot order
  uint32_t cnt = 0;
  unsigned long long ne = 0;
  for (uint32_t s0 = 0; s0 < nb; s0 += 64) {
    const uint32_t e = s0 + lane < nb ? tab[s0 + lane] : 0u;
    const uint32_t id = e & IDM;
    const uint32_t bn = e ? bsz[b0 + id] : 0u;
    const bool keep = bn > 2 && bn <= upper;
    const uint64_t km = __ballot(keep);
    if (keep) {
      const uint32_t idx = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
      ids_all[b0 + idx] = bord[b0 + id];
      ne += bn;
    }
    cnt += (uint32_t)__builtin_popcountll(km);
  }
  for (int o = 32; o; o >>= 1) ne += __shfl_xor(ne, o, 64);
  if (lane == 0) {
    gnb[g] = cnt;
    if (ne) atomicAdd(tot, ne);
  }
}

// the outer table's slots in order: how many visited buckets the group of every occupied slot contributes, and (after the scan)
// their ids into their final places
__global__ void k_outer_counts(const uint64_t *__restrict__ slots, uint32_t n_slots, const uint32_t *__restrict__ gord,
                               const uint32_t *__restrict__ gnb, uint32_t *__restrict__ cnt) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint64_t e = slots[s];
  cnt[s] = e ? gnb[gord[(uint32_t)e & 0x3FFFFFFFu]] : 0u;
}
__global__ void k_outer_place(const uint64_t *__restrict__ slots, uint32_t n_slots, const uint32_t *__restrict__ gord,
                              const uint32_t *__restrict__ gbucket, const uint32_t *__restrict__ ids_all, const uint32_t *__restrict__ off,
                              uint32_t *__restrict__ bid) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t d = off[s], c = off[s + 1] - d;
  if (!c) return;
  const uint32_t src = gbucket[gord[(uint32_t)slots[s] & 0x3FFFFFFFu]];
  for (uint32_t k = 0; k < c; ++k) bid[d + k] = ids_all[src + k];
}
}  // namespace

void dev_visit_inner(const DevicePairs &dp, uint32_t ovlp_upper, DevVisit &v) {
  PGX_REQUIRE(dp.tables && dp.max_group_buckets <= VISIT_WAVE_MAX, PGX_ESTATE, "dev_visit_inner: no device tables, or a group too large");
  const uint32_t ng = (uint32_t)dp.n_groups;
  hipStream_t st = ctx().stream;
  KernelTimer tm("visit", dp.n_buckets);
  v.ids_all.alloc(std::max<size_t>(dp.n_buckets, 1)), v.gnb.alloc(std::max<size_t>(ng, 1)), v.tot.alloc(1);
  PGX_HIP(hipMemsetAsync(v.tot.p, 0, sizeof(unsigned long long), st));
  if (!ng) return;
  hipLaunchKernelGGL(k_inner_lane, dim3((ng + LB - 1) / LB), dim3(LB), 0, st, dp.gstart.p, dp.gbucket.p, dp.gtrail.p, dp.bkey1_ord.p,
                     dp.bn_ord.p, dp.bord.p, ng, ovlp_upper, v.ids_all.p, v.gnb.p, v.tot.p);
  if (dp.n_big_groups)
    hipLaunchKernelGGL(k_inner_wave, dim3(dp.n_big_groups), dim3(64), 0, st, dp.big_groups.p, dp.gbucket.p, dp.gtrail.p, dp.bkey1_ord.p,
                       dp.bn_ord.p, dp.bord.p, ovlp_upper, v.ids_all.p, v.gnb.p, v.tot.p);
  PGX_HIP(hipGetLastError());
}

void dev_visit_place(const DevicePairs &dp, DevVisit &v, const uint64_t *slots, uint32_t n_slots, DevBuf<uint32_t> &bid, size_t *n_buckets,
                     size_t *n_entries) {
  hipStream_t st = ctx().stream;
  *n_buckets = *n_entries = 0;
  bid.alloc(std::max<size_t>(dp.n_buckets, 1));
  if (!n_slots || !dp.n_groups) return;
  uint32_t total = 0;
  unsigned long long ne = 0;
  {
    KernelTimer tm("visit", 0);
    DevBuf<uint64_t> d_own(n_slots);
    DevBuf<uint32_t> cnt(n_slots), off((size_t)n_slots + 1);
    PGX_HIP(hipMemcpyAsync(d_own.p, slots, (size_t)n_slots * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const uint64_t *d_slots_p = d_own.p;
    hipLaunchKernelGGL(k_outer_counts, dim3((n_slots + 255) / 256), dim3(256), 0, st, d_slots_p, n_slots, dp.gord.p, v.gnb.p, cnt.p);
    PGX_HIP(hipMemsetAsync(off.p, 0, sizeof(uint32_t), st));
    size_t bytes = 0;
    PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, cnt.p, off.p + 1, (int)n_slots, st));
    DevBuf<uint8_t> tmp(bytes + 256);
    PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp.p, bytes, cnt.p, off.p + 1, (int)n_slots, st));
    hipLaunchKernelGGL(k_outer_place, dim3((n_slots + 255) / 256), dim3(256), 0, st, d_slots_p, n_slots, dp.gord.p, dp.gbucket.p,
                       v.ids_all.p, off.p, bid.p);
    PGX_HIP(hipGetLastError());
    PGX_HIP(hipMemcpyAsync(&total, off.p + n_slots, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    PGX_HIP(hipMemcpyAsync(&ne, v.tot.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  }
  sync();   // (the temporaries go back to the block cache: stream-ordered reuse)
  *n_buckets = total, *n_entries = (size_t)ne;
}

}  // namespace pgx
