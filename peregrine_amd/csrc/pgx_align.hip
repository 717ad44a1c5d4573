// pgx_align.hip -- banded O(ND) furthest-reaching confirmation (ovlp_match, /root/reference/src/DWmatch.c:66-204),
// EIGHT (default) or four candidate alignments per wavefront.
//
// A candidate keeps on average 3.6 diagonals alive (<= 8 in 98.7 % of the steps, at most band+1 = 101), so one wavefront
// per candidate leaves most lanes idle -- and the kernel is bound by VALU issue (profiles/r01_pmc_align.txt), so lane
// utilisation is throughput.  Here a wavefront is eight independent 8-lane groups (GL = 8; or four 16-lane groups, the
// description below uses 16); each group runs the reference's d-loop for its own
// candidate, in lock-step with the other three, and pulls the next candidate from a device-wide counter the moment it
// finishes (persistent groups: no tail inside the wave).  Lane j of a group owns diagonal k = min_k + 2*(base+j) of the
// current step; wider bands take several rounds of 16.  V lives in a per-group LDS ring indexed by k (only the previous
// step's values are ever read, so 2*band+8 slots never alias live data; only V[1] needs to start at 0).
// Per step: start point from V[k-1], V[k+1]; an 8-code probe per lane (off-diagonal fronts stop there); long snakes are
// extended by the whole group, 128 codes per iteration (8 per lane with 16 lanes, 16 per lane with 8), with coalesced loads; the order-dependent side results (first
// diagonal reaching an end, first extension > 16, first occurrence of the strictly longest extension) are resolved
// lowest-k-first with ballots restricted to the group; band update by ballot of U >= best - band.
#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {
namespace {

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
struct U128 {
  uint64_t lo, hi;
};
__device__ __forceinline__ U128 load_u128_unaligned(const uint8_t *p) {  // one global_load_dwordx4 (the seqdb has 1 KiB of tail padding)
  U128 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}
__device__ __forceinline__ int match8(uint64_t qa, uint64_t ta, int qs, int ts) {
  const uint64_t diff = ((qa >> qs) ^ (ta >> ts)) & 0x0F0F0F0F0F0F0F0FULL;
  return diff ? (__builtin_ctzll(diff) >> 3) : 8;
}
// (the HIP __ballot takes an int: the compiler materialises 0/1 and compares again; the builtin takes the predicate's lane mask as is)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
template <int CTRL>
__device__ __forceinline__ int dpp_max(int v) {
  return max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true));   // (bound_ctrl: folds into one v_max_i32_dpp)
}
// maximum over the GL (8 or 16) lanes of a group, result in every lane: xor-butterfly with quad_perm / row_half_mirror /
// row_mirror
template <int GL>
__device__ __forceinline__ int group_max_i32(int v) {
  v = dpp_max<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_max<0x4E>(v);   // quad_perm [2,3,0,1]
  if (GL >= 8) v = dpp_max<0x141>(v);  // row_half_mirror: lanes i <-> 7-i of every 8
  if (GL == 16) v = dpp_max<0x140>(v);  // row_mirror: lanes i <-> 15-i
  return v;
}
template <int GL>
__device__ __forceinline__ uint32_t group_bits(uint64_t wave_mask, int gbase) {
  return (uint32_t)(wave_mask >> gbase) & ((1u << GL) - 1u);
}
template <int CTRL>
__device__ __forceinline__ int dpp_min(int v) {
  return min(v, __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true));   // (one v_min_i32_dpp)
}
// minimum over the 8 lanes of a group, result in every lane (same butterfly as group_max_i32<8>)
__device__ __forceinline__ int group_min8_i32(int v) {
  v = dpp_min<0xB1>(v);
  v = dpp_min<0x4E>(v);
  return dpp_min<0x141>(v);
}
// v_ffbl_b32 as the hardware defines it: the position of the lowest set bit, 0xFFFFFFFF for 0 (__builtin_ctz leaves 0 undefined and the
// guarded form costs a compare and a select)
__device__ __forceinline__ uint32_t ffbl_or_ones(uint32_t v) {
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
// value of lane l of the caller's group: gb4 = byte address of the group's lane 0 (ds_bpermute addresses lanes by 4 l).  __shfl computes
// ((l & 63) | (self & ~63)) << 2 in front of every call -- three instructions of a kernel that is bound by their issue
__device__ __forceinline__ int group_lane(int v, int gb4, int l) { return __builtin_amdgcn_ds_bpermute(gb4 + (l << 2), v); }

// ---- one candidate per wavefront: all per-candidate state is wave-uniform (scalar registers, readlane instead of
// cross-lane permutes, no predicated bookkeeping), so a d-step costs roughly half the instructions of the grouped kernel.
// A lone alignment is a chain of 300-600 dependent steps bound by instruction issue, hence this form for SMALL launches
// (the replay's tail rounds), where latency is everything and idle lanes cost nothing.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_step(int v) {  // v = max(v, v from the DPP-selected lane); lanes without a source keep v
  return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ int wave_max_i32(int v) {  // DPP tree: 6 VALU ops, result broadcast from lane 63
  v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row maximum
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave maximum
  return __builtin_amdgcn_readlane(v, 63);
}


}  // namespace

// 16 codes of two 2-bit packs compared (pgx_pack.hip): q / t address the dword that holds base 0, xq / yt = position from there on
__device__ __forceinline__ int match16p(const uint8_t *q, uint32_t xq, const uint8_t *t, uint32_t yt) {
  uint2 qd, td;
  __builtin_memcpy(&qd, q + (xq >> 4) * 4, 8), __builtin_memcpy(&td, t + (yt >> 4) * 4, 8);
  const uint32_t df = __builtin_amdgcn_alignbit(qd.y, qd.x, (xq & 15) << 1) ^ __builtin_amdgcn_alignbit(td.y, td.x, (yt & 15) << 1);
  return (int)min(ffbl_or_ones(df) >> 1, 16u);
}
// PACKED (round 6): seq = the 2-bit packs, roff = d_poff (dword index of a read's forward strand; its reverse complement follows): a probe
// compares 16 codes, a snake iteration 64 x 16.  The run a diagonal is extended by is the same whatever the piece size, so the results are
// those of the byte form; candidates on reads without 2-bit codes never come here (dev_align).
template <bool PACKED>
__device__ __forceinline__ void align_one_per_wave(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ roff,
                                                   const uint32_t *__restrict__ rlen, const pgx_align_key *__restrict__ keys, uint32_t a, int band,
                                                   int ring, pgx_match *__restrict__ out) {
  extern __shared__ int32_t V[];
  constexpr int PB = PACKED ? 16 : 8;   // codes of a probe / of a lane's piece of a snake iteration
  const int lane = threadIdx.x;
  const pgx_align_key key = keys[a];
  const uint32_t len0 = rlen[key.rid0], len1 = rlen[key.rid1];
  const uint8_t *q = PACKED ? seq + (roff[key.rid0] + (key.dir0 ? (len0 + 15u) >> 4 : 0u) + (key.q_off >> 4)) * 4 : seq + roff[key.rid0] + key.q_off;
  const uint8_t *t = PACKED ? seq + (roff[key.rid1] + (key.dir1 ? (len1 + 15u) >> 4 : 0u)) * 4 : seq + roff[key.rid1];
  const uint32_t qo = PACKED ? key.q_off & 15u : 0u;
  const int q_len = (int)(len0 - key.q_off);
  const int t_len = (int)len1;
  const int qs = key.dir0 ? 4 : 0, ts = key.dir1 ? 4 : 0;
  const int max_d = (int)(0.3 * (double)(q_len + t_len));  // DWmatch.c:96, one IEEE double multiply
  const int band_size = band * 2;
  const int mask = ring - 1;
  for (int i = lane; i < ring; i += 64) V[i] = 0;
  __syncthreads();

  int best_m = -1, min_k = 0, max_k = 0;
  uint32_t longest = 0;
  bool started = false, matched = false;
  int q_bgn = 0, t_bgn = 0, q_m_end = 0, t_m_end = 0, q_end = 0, t_end = 0, dist = 0;

  for (int d = 0; d < max_d; ++d) {
    if (max_k - min_k > band_size) break;
    const int nk = max_k >= min_k ? ((max_k - min_k) >> 1) + 1 : 0;
    int x = 0, y = 0;
    for (int base = 0; base < nk && !matched; base += 64) {
      const int j = base + lane;
      const bool active = j < nk;
      const int k = min_k + 2 * j;
      int x1 = 0, y1 = 0;
      x = 0, y = 0;
      bool more = false;
      if (active) {
        const int va = V[(k - 1) & mask], vb = V[(k + 1) & mask];
        x = (k == min_k || (k != max_k && va < vb)) ? vb : va + 1;
        y = x - k;
        x1 = x, y1 = y;
        // probe: the first 8 codes.  Off-diagonal fronts almost always stop here.
        const int rem = min(q_len - x, t_len - y);
        if (rem > 0) {
          int m = PACKED ? match16p(q, qo + (uint32_t)x, t, (uint32_t)y) : match8(load_u64_unaligned(q + x), load_u64_unaligned(t + y), qs, ts);
          m = min(m, rem);
          x += m, y += m;
          more = (m == PB) && (rem > PB);
        }
      }
      // long snakes (normally one per step): the whole wavefront extends one diagonal, 512 codes per iteration
      uint64_t mm = ballot64(more);
      while (mm) {
        const int L = __builtin_ctzll(mm);
        const int xs = __builtin_amdgcn_readlane(x, L), ys = __builtin_amdgcn_readlane(y, L);
        const int rem = min(q_len - xs, t_len - ys);  // > 0 by construction
        const int off = lane * PB;
        int m = 0;
        if (off < rem)
          m = min(PACKED ? match16p(q, qo + (uint32_t)(xs + off), t, (uint32_t)(ys + off))
                         : match8(load_u64_unaligned(q + xs + off), load_u64_unaligned(t + ys + off), qs, ts), rem - off);
        const uint64_t stop = ballot64(m < PB);
        int ext;
        if (stop) {
          const int f = __builtin_ctzll(stop);
          ext = PB * f + __builtin_amdgcn_readlane(m, f);
        } else {
          ext = 64 * PB;
        }
        if (lane == L) x += ext, y += ext;
        if (stop || ext >= rem) mm &= mm - 1;  // this diagonal is done (mismatch found or an end reached)
      }
      const int ext = x - x1;
      const bool hit = active && (x >= q_len || y >= t_len);
      const uint64_t hitmask = ballot64(hit);
      const int hl = hitmask ? __builtin_ctzll(hitmask) : 64;
      const bool valid = active && lane <= hl;
      if (!started) {
        const uint64_t m = ballot64(valid && ext > 16);
        if (m) {
          const int l = __builtin_ctzll(m);
          q_bgn = __builtin_amdgcn_readlane(x1, l), t_bgn = __builtin_amdgcn_readlane(y1, l);
          started = true;
        }
      }
      if (ballot64(valid && (uint32_t)ext > longest)) {
        const int mx = wave_max_i32(valid ? ext : -1);
        const int l = __builtin_ctzll(ballot64(valid && ext == mx));
        longest = (uint32_t)mx;
        q_m_end = __builtin_amdgcn_readlane(x, l), t_m_end = __builtin_amdgcn_readlane(y, l);
      }
      if (valid) V[k & mask] = x;
      best_m = max(best_m, wave_max_i32(valid ? x + y : -1));
      if (hitmask) {
        matched = true;
        q_end = __builtin_amdgcn_readlane(x, hl), t_end = __builtin_amdgcn_readlane(y, hl);
      }
    }
    __syncthreads();
    if (matched) {
      dist = d;
      break;
    }
    // band update (DWmatch.c:166-183)
    int new_min = max_k, new_max = min_k;
    const int thr = best_m - band;
    for (int base = 0; base < nk; base += 64) {
      const int j = base + lane;
      const int k2 = min_k + 2 * j;
      int u;
      if (nk <= 64) u = x + y;  // still in registers
      else u = j < nk ? 2 * V[k2 & mask] - k2 : 0;
      const uint64_t m = ballot64(j < nk && u >= thr);
      if (m) {
        new_min = min(new_min, min_k + 2 * (base + __builtin_ctzll(m)));
        new_max = max(new_max, min_k + 2 * (base + 63 - __builtin_clzll(m)));
      }
    }
    max_k = new_max + 1;
    min_k = new_min - 1;
  }
  if (lane == 0) {
    pgx_match r;
    if (matched) {
      r.q_bgn = q_bgn, r.t_bgn = t_bgn, r.q_end = q_end, r.t_end = t_end, r.dist = dist;
      r.m_size = (q_end - q_bgn + t_end - t_bgn + 2 * dist) / 2;
    } else {
      r.q_bgn = 0, r.t_bgn = 0, r.q_end = 0, r.t_end = 0, r.dist = 0, r.m_size = 0;
    }
    r.q_m_end = q_m_end, r.t_m_end = t_m_end;
    out[a] = r;
  }
}
template <bool PACKED>
__global__ __launch_bounds__(64) void k_align1(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ roff,
                                              const uint32_t *__restrict__ rlen,
                                              const pgx_align_key *__restrict__ keys, uint32_t n, int band,
                                              int ring, pgx_match *__restrict__ out) {
  if (blockIdx.x < n) align_one_per_wave<PACKED>(seq, roff, rlen, keys, blockIdx.x, band, ring, out);
}
// the same over a device-resident list of candidates (the ones a grouped launch handed on: reads with ambiguous bases, and the
// STRAGGLERS -- round 3: at C4 scale ~600 of 9 M candidates of a launch run through low-complexity sequence with a band of up to 100
// diagonals for thousands of steps; 8 lanes take 13 rounds per step for them, and the launch waited ~100 ms for them alone
// (profiles/r03c_align_iterations.txt).  A whole wavefront takes such a band in two rounds.)
template <bool PACKED>
__global__ __launch_bounds__(64) void k_align1_list(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ roff,
                                                   const uint32_t *__restrict__ rlen, const pgx_align_key *__restrict__ keys,
                                                   const uint32_t *__restrict__ list_n, const uint32_t *__restrict__ list, int band, int ring,
                                                   pgx_match *__restrict__ out) {
  const uint32_t cnt = *list_n;
  for (uint32_t i = blockIdx.x; i < cnt; i += gridDim.x) {
    __syncthreads();   // (the previous candidate's V ring is done with)
    align_one_per_wave<PACKED>(seq, roff, rlen, keys, list[i], band, ring, out);
  }
}


// GL = lanes per candidate: 16 (four candidates per wavefront) or 8 (eight).
// VT = storage type of the V ring: uint16_t when no read is longer than 65,535 bases (x <= q_len fits), which halves the LDS of
// a workgroup and lets 32 instead of 20 eight-candidate wavefronts share a CU; int32_t otherwise.
template <int GL, typename VT>
__global__ __launch_bounds__(64) void k_align4(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ roff,
                                               const uint32_t *__restrict__ rlen,
                                               const pgx_align_key *__restrict__ keys, uint32_t n, int band, int ring,
                                               pgx_match *__restrict__ out, uint32_t *__restrict__ counter) {
  extern __shared__ int32_t Vall[];
  const int lane = threadIdx.x, gl = lane & (GL - 1), gbase = lane & ~(GL - 1);
  VT *V = reinterpret_cast<VT *>(Vall) + (lane / GL) * ring;
  const int mask = ring - 1, band_size = band * 2;

  uint32_t c_next = 0, c_end = 0;   // the wavefront's chunk of the work counter
  // per-candidate state, uniform within a 16-lane group
  bool alive = false, exhausted = false;
  uint32_t a = 0;
  const uint8_t *q = seq, *t = seq;
  int q_len = 0, t_len = 0, qs = 0, ts = 0, max_d = 0, d = 0;
  int best_m = -1, min_k = 0, max_k = 0;
  uint32_t longest = 0;
  bool started = false, matched = false;
  int q_bgn = 0, t_bgn = 0, q_m_end = 0, t_m_end = 0, q_end = 0, t_end = 0;

  for (;;) {
    // ---- idle groups pull the next candidate ---------------------------------------------------------------
    // (the work counter in chunks of 8 per wavefront, as in k_align_ph below: one address, ~12 ns per same-address atomic)
    const bool fetching = !alive && !exhausted;
    const uint64_t fw = ballot64(fetching);
    uint32_t na = 0;
    if (fw) {
      constexpr uint32_t CHUNK = 8;
      uint64_t need = 0;
      for (int g = 0; g < 64; g += GL) need |= 1ULL << g;   // the first lanes of the groups
      need &= fw;
      const uint32_t cnt = (uint32_t)__builtin_popcountll(need);
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      const uint32_t avail = __builtin_amdgcn_readfirstlane(c_end - c_next);
      na = c_next + r;
      if (avail < cnt) {
        uint32_t got = 0;
        if (lane == 0) got = atomicAdd(counter, CHUNK);
        got = __builtin_amdgcn_readfirstlane(got);
        if (r >= avail) na = got + (r - avail);
        c_next = got + (cnt - avail), c_end = got + CHUNK;
      } else {
        c_next += cnt;
      }
    }
    if (fetching) {
      na = (uint32_t)__shfl((int)na, gbase, 64);
      if (na >= n) {
        exhausted = true;
      } else {
        a = na;
        const pgx_align_key key = keys[a];
        q = seq + roff[key.rid0] + key.q_off;
        t = seq + roff[key.rid1];
        q_len = (int)(rlen[key.rid0] - key.q_off);
        t_len = (int)rlen[key.rid1];
        qs = key.dir0 ? 4 : 0, ts = key.dir1 ? 4 : 0;
        max_d = (int)(0.3 * (double)(q_len + t_len));  // DWmatch.c:96, one IEEE double multiply
        d = 0, best_m = -1, min_k = 0, max_k = 0, longest = 0;
        started = matched = false;
        q_bgn = t_bgn = q_m_end = t_m_end = q_end = t_end = 0;
        if (gl == 0) V[1 & mask] = (VT)0;  // the only slot read before it is written (d = 0 reads V[k+1] = V[1])
        alive = true;
      }
    }
    if (!ballot64(alive)) break;
    __syncthreads();

    // ---- end of the d-loop without a match (DWmatch.c:118-122,196-199) ---------------------------------------
    bool stepping = alive;
    if (alive && (d >= max_d || max_k - min_k > band_size)) {
      if (gl == 0) {
        pgx_match r;
        r.m_size = 0, r.dist = 0, r.q_bgn = 0, r.q_end = 0, r.t_bgn = 0, r.t_end = 0;
        r.t_m_end = t_m_end, r.q_m_end = q_m_end;
        out[a] = r;
      }
      alive = false, stepping = false;
    }

    // ---- one step: all diagonals of the current band, 16 per round -------------------------------------------
    const int nk = max_k >= min_k ? ((max_k - min_k) >> 1) + 1 : 0;
    int x = 0, y = 0;
    for (int base = 0;; base += GL) {
      const bool inround = stepping && !matched && base < nk;
      if (!ballot64(inround)) break;
      const int j = base + gl;
      const bool active = inround && j < nk;
      const int k = min_k + 2 * j;
      int x1 = 0, y1 = 0;
      bool more = false;
      if (inround) x = 0, y = 0;
      if (active) {
        const int va = (int)V[(k - 1) & mask], vb = (int)V[(k + 1) & mask];
        x = (k == min_k || (k != max_k && va < vb)) ? vb : va + 1;
        y = x - k;
        x1 = x, y1 = y;
        const int rem = min(q_len - x, t_len - y);
        if (rem > 0) {  // probe: the first 8 codes
          int m = match8(load_u64_unaligned(q + x), load_u64_unaligned(t + y), qs, ts);
          m = min(m, rem);
          x += m, y += m;
          more = (m == 8) && (rem > 8);
        }
      }
      // long snakes: the group extends one diagonal at a time, 128 codes per iteration
      uint64_t mw = ballot64(more);
      while (mw) {
        const uint32_t gm = group_bits<GL>(mw, gbase);
        const bool has = gm != 0;
        const int L = has ? __builtin_ctz(gm) : 0;
        const int xs = __shfl(x, gbase + L, 64), ys = __shfl(y, gbase + L, 64);
        const int rem = min(q_len - xs, t_len - ys);
        // codes per lane and iteration: 8 with 16-lane groups, 16 with 8-lane groups (128 per group either way)
        constexpr int SL = GL == 16 ? 8 : 16;
        const int off = gl * SL;
        int m = SL;
        if (has) {
          m = 0;
          if (off < rem) {
            m = match8(load_u64_unaligned(q + xs + off), load_u64_unaligned(t + ys + off), qs, ts);
            if (SL == 16 && m == 8 && off + 8 < rem)
              m += match8(load_u64_unaligned(q + xs + off + 8), load_u64_unaligned(t + ys + off + 8), qs, ts);
            m = min(m, rem - off);
          }
        }
        const uint32_t sg = group_bits<GL>(ballot64(has && m < SL), gbase);
        int ext = GL * SL;
        if (sg) {
          const int f = __builtin_ctz(sg);
          ext = SL * f + __shfl(m, gbase + f, 64);
        }
        if (has && gl == L) {
          x += ext, y += ext;
          if (sg || ext >= rem) more = false;  // mismatch found or an end reached: this diagonal is done
        }
        mw = ballot64(more);
      }
      const int ext = x - x1;
      const bool hit = active && (x >= q_len || y >= t_len);
      const uint64_t hitw = ballot64(hit);  // rare (once per candidate): everything that depends on it sits behind the branch
      int hl = GL;
      if (hitw) {
        const uint32_t hitm = group_bits<GL>(hitw, gbase);
        if (hitm) hl = __builtin_ctz(hitm);
      }
      const bool valid = active && gl <= hl;
      {  // first extension > 16 fixes q_bgn/t_bgn once (DWmatch.c:142-146)
        const uint64_t sw = ballot64(valid && ext > 16 && !started);
        if (sw) {
          const uint32_t m = group_bits<GL>(sw, gbase);
          const int l = m ? __builtin_ctz(m) : 0;
          const int bx = __shfl(x1, gbase + l, 64), by = __shfl(y1, gbase + l, 64);
          if (m) q_bgn = bx, t_bgn = by, started = true;
        }
      }
      if (ballot64(valid && (uint32_t)ext > longest)) {  // strictly longer extension (DWmatch.c:148-152)
        const int mx = group_max_i32<GL>(valid ? ext : -1);
        const uint32_t m = group_bits<GL>(ballot64(valid && ext == mx), gbase);
        const int l = m ? __builtin_ctz(m) : 0;
        const int ex = __shfl(x, gbase + l, 64), ey = __shfl(y, gbase + l, 64);
        if (inround && mx >= 0 && (uint32_t)mx > longest) longest = (uint32_t)mx, q_m_end = ex, t_m_end = ey;
      }
      if (valid) V[k & mask] = (VT)x;
      {
        const int s = group_max_i32<GL>(valid ? x + y : -1);
        if (inround) best_m = max(best_m, s);
      }
      if (hitw) {
        const uint32_t hitm = group_bits<GL>(hitw, gbase);
        const int ex = __shfl(x, gbase + (hl & (GL - 1)), 64), ey = __shfl(y, gbase + (hl & (GL - 1)), 64);
        if (inround && hitm) matched = true, q_end = ex, t_end = ey;
      }
    }
    __syncthreads();

    if (stepping && matched) {  // DWmatch.c:185-194
      if (gl == 0) {
        pgx_match r;
        r.q_bgn = q_bgn, r.t_bgn = t_bgn, r.q_end = q_end, r.t_end = t_end, r.dist = d;
        r.m_size = (q_end - q_bgn + t_end - t_bgn + 2 * d) / 2;
        r.t_m_end = t_m_end, r.q_m_end = q_m_end;
        out[a] = r;
      }
      alive = false, stepping = false;
    }
    // ---- band update (DWmatch.c:166-183) -------------------------------------------------------------------------
    int new_min = max_k, new_max = min_k;
    const int thr = best_m - band;
    for (int base = 0;; base += GL) {
      const bool inround = stepping && base < nk;
      if (!ballot64(inround)) break;
      const int j = base + gl;
      const int k2 = min_k + 2 * j;
      int u = 0;
      if (inround && j < nk) u = (nk <= GL) ? x + y : 2 * (int)V[k2 & mask] - k2;
      const uint32_t m = group_bits<GL>(ballot64(inround && j < nk && u >= thr), gbase);
      if (m) {
        new_min = min(new_min, min_k + 2 * (base + __builtin_ctz(m)));
        new_max = max(new_max, min_k + 2 * (base + 31 - __builtin_clz(m)));
      }
    }
    if (stepping) max_k = new_max + 1, min_k = new_min - 1, ++d;
  }
}


// ---- the same d-loop as a per-group PHASE MACHINE (round 2) ------------------------------------------------------------------
// k_align4 steps all groups of a wavefront through one d-step per iteration: a step takes as many rounds of GL diagonals, and
// its snake loop as many iterations, as the NEEDIEST group of the wavefront -- which is why narrower groups did not pay there
// (with sixteen 4-lane groups some group needs a second round in nearly every step).  Here every group carries its own phase
//   FETCH -> STEP (loop conditions of DWmatch.c:118-122) -> ROUND (start points + 8-code probe of GL diagonals) -> SNAKE (one
//   64- or 128-code extension of the group's lowest unfinished diagonal per iteration) -> END (the order-dependent side results
//   of the round, V, best_m, the first diagonal that reaches an end) -> next ROUND, or BAND (one round of the band scan per
//   iteration, DWmatch.c:166-183) -> STEP of d + 1
// and an iteration of the wavefront runs every phase body once, each under its groups' predicate; a group passes through
// ROUND, SNAKE, END and BAND within ONE iteration when its step has a single round and a single extension (the common case),
// and only the groups that need more take more iterations.  Lane utilisation is what the kernel is bound by (VALU issue):
// sixteen 4-lane groups keep 3.6 live diagonals on 4 lanes instead of 8.
enum { PH_FETCH = 0, PH_STEP = 1, PH_ROUND = 2, PH_SNAKE = 3, PH_END = 4, PH_BAND = 5, PH_DONE = 6 };
// The V rings are exchanged between the lanes of ONE wavefront (the block is a wavefront): its LDS instructions execute in program
// order, so all the two exchange points of an iteration need is that the compiler keeps that order -- not __syncthreads()'s
// s_waitcnt vmcnt(0) lgkmcnt(0), which also drains the sequence loads in flight.  -DPGX_PH_BARRIER: the round-2 form.
#ifdef PGX_PH_BARRIER
#define PH_SYNC() __syncthreads()
#else
#define PH_SYNC() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
#endif
// PACKED (round 3): the reads come from the 2-bit packs of the seqdb (pgx_pack.hip: k_pack2, one pass per read database, both
// strands) instead of its bytes: the probe compares 16 bases with two funnel shifts, an XOR and a find-first-bit (8 codes, two
// 64-bit shifts, XOR, AND and a 64-bit count before), an extension step 32 bases per lane = 256 per group with four funnel shifts
// (128 with four 64-bit shift / XOR / AND / count sequences before), and fewer steps need an extension at all (a probe of 16 ends
// 32 % of the main diagonals' matches, one of 8 only 18 %).  seq = the two packs (pack1 = seq + pack_stride dwords); candidates
// that meet a read with ambiguous bases (nflag) are handed on to the byte-wise launch through esc_list.
// ROUND 6 -- a WORKGROUP of up to 16 wavefronts shares one run of the request list.  The wavefronts still run on their own (no barrier
// inside the loop; every V ring belongs to one 8-lane group), but they take their chunks of 8 candidates from a SEGMENT of `seg`
// consecutive requests that the workgroup claimed from the device-wide counter, through one packed LDS word {segment start, taken}.
// With the requests of a launch in the order of the packs' layout (order list: dev_align sorts them by the query read's rank) the
// 128 candidates a workgroup has in flight are neighbours in the layout -- one or two loci -- and a CU (two workgroups) touches a few
// dozen translation ranges of the 47 GB of packs instead of 512 (pgx_pack.hip; rounds 2-5: a wavefront per workgroup, chunks of 8
// straight from the device-wide counter: every wavefront of a CU at another locus).
// PACKED: seq = the packs, roff = d_poff (dword index of a read's forward strand; its reverse complement follows at + ceil(len / 16)).
template <int GL, typename VT, bool PACKED>
__global__ __launch_bounds__(1024, 8) void k_align_ph(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ roff,
                                                      const uint32_t *__restrict__ rlen, const pgx_align_key *__restrict__ keys,
                                                      uint32_t n, int band, int ring, pgx_match *__restrict__ out,
                                                      uint32_t *__restrict__ counter, const uint32_t *__restrict__ redo_n,
                                                      const uint32_t *__restrict__ redo_list, uint32_t *__restrict__ esc_n,
                                                      uint32_t *__restrict__ esc_list, const uint32_t *__restrict__ nflag, uint32_t seg,
                                                      uint32_t iter_limit) {   // iter_limit != 0: a candidate still running after that many
                                                                               // wavefront iterations is handed on (esc_list) at its next step
  // redo_list != nullptr: the candidates are keys[redo_list[0 .. *redo_n)] (an order list, or the ones another launch handed on);
  // esc_list != nullptr: stragglers and candidates on reads without 2-bit codes are appended there instead of being finished
  extern __shared__ int32_t Vall[];
  __shared__ unsigned long long s_state;   // {start of the workgroup's segment : 32 | requests of it handed out : 32}
  __shared__ uint32_t s_lock;              // held by the wavefront that is claiming the next segment
  static_assert(GL == 8, "the group operations below are written for 8-lane groups (half a DPP row)");
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1), gbase = lane & ~(GL - 1), gb4 = gbase << 2;
  VT *V = reinterpret_cast<VT *>(Vall) + ((threadIdx.x >> 6) * (64 / GL) + lane / GL) * ring;
  if (threadIdx.x == 0) s_state = (unsigned long long)seg, s_lock = 0u;   // (an exhausted segment: the first fetch claims one)
  __syncthreads();
  const int mask = ring - 1, band_size = band * 2;
  constexpr int SL = PACKED ? 32 : (GL == 16 ? 8 : 16);  // codes per lane and snake iteration
  constexpr int PROBE = PACKED ? 16 : 8;                   // codes of a probe
  if (redo_list) n = *redo_n;
  const uint32_t iter_budget = esc_list && iter_limit ? iter_limit : 0xFFFFFFFFu;
  uint32_t qo = 0, to = 0;   // PACKED: position of base 0 inside the dword q / t point at (q, t then address dwords of a pack)
  uint32_t iters = 0;        // wavefront iterations since the group fetched its candidate
  uint32_t c_next = 0, c_end = 0;   // the wavefront's chunk of the work counter (uniform over the wavefront)

  // per-candidate state, uniform within a group
  int phase = PH_FETCH;
  uint32_t a = 0;
  const uint8_t *q = seq, *t = seq;
  int q_len = 0, t_len = 0, qs = 0, ts = 0, max_d = 0, d = 0;
  int best_m = -1, min_k = 0, max_k = 0, nk = 0, base = 0, bbase = 0, new_min = 0, new_max = 0;
  uint32_t longest = 0;
  bool started = false;
  int q_bgn = 0, t_bgn = 0, q_m_end = 0, t_m_end = 0;
  // per-lane state of the running round
  int x = 0, y = 0, x1 = 0, y1 = 0, k = 0;
  bool active = false;
  uint32_t gm = 0;   // SNAKE: the lanes of the group whose diagonal still has codes to compare (bit gl; uniform within the group)

#ifdef PGX_ALIGN_STATS
  uint32_t my_iters = 0;
  int my_maxnk = 0;
  bool has_cand = false;
#endif
  for (;;) {
#ifdef PGX_ALIGN_STATS
    if (phase == PH_FETCH && has_cand && gl == 0 && !redo_list) {   // (stats build: counter[1..32] / [33..64]: log2 histogram of the wavefront iterations a candidate took / their sums)
      const uint32_t b = my_iters ? 31u - (uint32_t)__builtin_clz(my_iters) : 0u;
      atomicAdd(counter + 1 + min(b, 31u), 1u);
#ifdef PGX_ALIGN_STATS_NK
      atomicAdd(counter + 33 + min((uint32_t)my_maxnk, 31u), 1u);   // (instead of the sums: candidates by the most diagonals a step of theirs had)
#else
      atomicAdd(counter + 33 + min(b, 31u), my_iters);
#endif
    }
    if (phase == PH_FETCH) has_cand = false, my_iters = 0, my_maxnk = 0;
    my_maxnk = max(my_maxnk, phase == PH_ROUND || phase == PH_STEP ? nk : 0);
    ++my_iters;
#endif
    // ---- FETCH: idle groups pull the next candidate --------------------------------------------------------------
    ++iters;
    // The work counter is ONE address: an L2 channel serves same-address atomics one wavefront-instruction at a time, ~12 ns each
    // (measured on k_keep) -- with an add per candidate a launch of 4.66 M candidates cannot finish in under 56 ms whatever
    // the kernel does, and that is exactly where rounds 3 and 4 stood (84 M alignments/s at c3 for every form of the kernel; 24 more
    // VALU or 40 more SALU instructions per iteration changed nothing).  A wavefront therefore takes CHUNK candidates per add and hands
    // them to its groups as they come free.  What a wavefront holds back at the end of a launch is other wavefronts' idle time: alignment
    // kernels per c3 / c4s step with CHUNK 8: 45.5 / 123.8 ms, 16: 46.0 / 127.3, 64: 49.0 / 157.4 (c4s's last candidates are its longest);
    // the last 32 candidates per wavefront taken one by one again: 48.5 / 128.5 (the tail is then the old floor).
    constexpr uint32_t RETRY = 0xFFFFFFFFu;   // "no chunk this time" (never a request number: n < 2^31)
    const bool fetching = phase == PH_FETCH;
    const uint64_t fw = ballot64(fetching);
    uint32_t na = 0;
    if (fw) {   // (wave-uniform; ~3 % of the iterations)
      constexpr uint32_t CHUNK = 8;
      const uint64_t need = fw & 0x0101010101010101ULL;   // the first lanes of the fetching groups
      const uint32_t cnt = (uint32_t)__builtin_popcountll(need);
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));   // such lanes below this one
      const uint32_t avail = __builtin_amdgcn_readfirstlane(c_end - c_next);
      na = c_next + r;
      if (avail < cnt) {   // a new chunk; what is left of the old one goes out first
        uint32_t got = RETRY;
        if (lane == 0) {
          unsigned long long o = atomicAdd(&s_state, (unsigned long long)CHUNK);   // (one LDS atomic: start and count of one and the same segment)
          if ((uint32_t)o < seg) {
            got = (uint32_t)(o >> 32) + (uint32_t)o;
          } else if (atomicCAS(&s_lock, 0u, 1u) == 0u) {   // the segment is used up and nobody is fetching the next one yet
            o = atomicAdd(&s_state, (unsigned long long)CHUNK);   // (under the lock: a wavefront may have installed one since the look above)
            if ((uint32_t)o < seg) {
              got = (uint32_t)(o >> 32) + (uint32_t)o;
            } else {
              got = atomicAdd(counter, seg);
              atomicExch(&s_state, ((unsigned long long)got << 32) | CHUNK);
            }
            atomicExch(&s_lock, 0u);   // (LDS operations of a wavefront execute in order: the segment is in place before the lock opens)
          }
          // else: another wavefront of the workgroup is fetching the next segment; this one's idle groups ask again in the next iteration
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got == RETRY) {
          if (r >= avail) na = RETRY;
          c_next = c_end = 0;
        } else {
          if (r >= avail) na = got + (r - avail);
          c_next = got + (cnt - avail), c_end = got + CHUNK;
        }
      } else {
        c_next += cnt;
      }
    }
    if (fetching) {
      iters = 0;
      na = (uint32_t)group_lane((int)na, gb4, 0);
      if (na == RETRY) {
        // (stays in FETCH)
      } else if (na >= n) {
        phase = PH_DONE;
      } else {
#ifdef PGX_ALIGN_STATS
        has_cand = true;
#endif
        a = redo_list ? redo_list[na] : na;
        const pgx_align_key key = keys[a];
        const uint32_t len0 = rlen[key.rid0], len1 = rlen[key.rid1];
        if (PACKED) {   // (pgx_pack.hip: [forward | reverse complement] of a read, each strand from a dword on)
          q = seq + (roff[key.rid0] + (key.dir0 ? (len0 + 15u) >> 4 : 0u) + (key.q_off >> 4)) * 4;
          t = seq + (roff[key.rid1] + (key.dir1 ? (len1 + 15u) >> 4 : 0u)) * 4;
          qo = key.q_off & 15u, to = 0;
        } else {
          q = seq + roff[key.rid0] + key.q_off;
          t = seq + roff[key.rid1];
        }
        q_len = (int)(len0 - key.q_off);
        t_len = (int)len1;
        qs = key.dir0 ? 4 : 0, ts = key.dir1 ? 4 : 0;
        max_d = (int)(0.3 * (double)(q_len + t_len));  // DWmatch.c:96, one IEEE double multiply
        d = 0, best_m = -1, min_k = 0, max_k = 0, longest = 0;
        started = false;
        q_bgn = t_bgn = q_m_end = t_m_end = 0;
        if (gl == 0) V[1 & mask] = (VT)0;  // the only slot read before it is written (d = 0 reads V[k+1] = V[1])
        phase = PH_STEP;
        if (PACKED && ((nflag[key.rid0] | nflag[key.rid1]) & 1u)) {   // an ambiguous base has no 2-bit code: the byte-wise launch takes it
          if (gl == 0) esc_list[n + atomicAdd(esc_n + 1, 1u)] = a;      // (second list of the escalation block: [4 + n ..), count at [1])
          phase = PH_FETCH;
        }
      }
    }
    if (!ballot64(phase != PH_DONE)) break;
    PH_SYNC();

    // ---- STEP: the loop conditions of a new d (DWmatch.c:118-122,196-199) -------------------------------------------
    // (the V ring holds every live diagonal of a band that passes the width test below: ring >= 2 band + 8 by dev_align, so the only
    //  hand-on left is the iteration budget -- round 2's narrow-ring launches are gone)
    if (iters > iter_budget && phase == PH_STEP && !(d >= max_d || max_k - min_k > band_size)) {   // a straggler: k_align1_list takes it, a wavefront of its own
      if (gl == 0) esc_list[atomicAdd(esc_n, 1u)] = a;
      phase = PH_FETCH;
    }
    if (phase == PH_STEP) {
      const int width = max_k - min_k;
      if (d >= max_d || width > band_size) {   // the end of the d-loop without a match
        if (gl == 0) {
          pgx_match r;
          r.m_size = 0, r.dist = 0, r.q_bgn = 0, r.q_end = 0, r.t_bgn = 0, r.t_end = 0;
          r.t_m_end = t_m_end, r.q_m_end = q_m_end;
          out[a] = r;
        }
        phase = PH_FETCH;
      } else if (width < 0) {   // a degenerate band: an empty k-loop, then the band update over nothing (followed literally)
        nk = 0, base = 0, bbase = 0, new_min = max_k, new_max = min_k, phase = PH_BAND;
      } else {
        nk = (width >> 1) + 1, base = 0, phase = PH_ROUND;
      }
    }

    // ---- ROUND: start points and the 8-code probe of GL diagonals ----------------------------------------------------
    int probe_full = 0;   // >= PROBE: the lane's probe matched throughout and its diagonal has more codes to compare
    if (phase == PH_ROUND) {
      const int j = base + gl;
      active = j < nk;
      k = min_k + 2 * j;
      // (x, y, x1, y1 of the lanes beyond the band keep whatever they held: every use below is behind `active`)
      if (active) {
        const int va = (int)V[(k - 1) & mask], vb = (int)V[(k + 1) & mask];
        x = (k == min_k || (k != max_k && va < vb)) ? vb : va + 1;
        y = x - k;
        x1 = x, y1 = y;
        const int rem = min(q_len - x, t_len - y);
        if (rem > 0) {
          int m;
          if (PACKED) {
            const uint32_t xq = qo + (uint32_t)x, yt = to + (uint32_t)y;
            uint2 qd, td;
            __builtin_memcpy(&qd, q + (xq >> 4) * 4, 8), __builtin_memcpy(&td, t + (yt >> 4) * 4, 8);
            const uint32_t df = __builtin_amdgcn_alignbit(qd.y, qd.x, (xq & 15) << 1) ^ __builtin_amdgcn_alignbit(td.y, td.x, (yt & 15) << 1);
            m = (int)min(ffbl_or_ones(df) >> 1, 16u);   // (no test of df: the instruction answers ~0 for 0)
          } else {
            m = match8(load_u64_unaligned(q + x), load_u64_unaligned(t + y), qs, ts);
          }
          probe_full = (int)min((uint32_t)m, (uint32_t)(rem - 1));   // (m <= PROBE: >= PROBE iff m == PROBE and rem > PROBE; rem >= 1, so unsigned
          m = (int)min((uint32_t)m, (uint32_t)rem);                  //  minima like the one with 16 above: three v_min_u32 in a row)
          x += m, y += m;
        }
      }
    }
    {
      // (a ballot of ONE integer compare is one v_cmp; of a bool the compiler materialises 0 / 1 and compares again.  Lanes outside
      //  ROUND hold 0.)  The group's mask is STATE from here on: SNAKE clears a bit per finished diagonal, no ballot of its own.
      asm volatile("" : "+v"(probe_full));   // (kept a VGPR value: as a bool the compiler carries it as a lane mask and the ballot costs a select + a compare on top)
      const uint32_t g = group_bits<GL>(ballot64(probe_full >= PROBE), gbase);
      if (phase == PH_ROUND) gm = g, phase = g ? PH_SNAKE : PH_END;
    }

    // ---- SNAKE: one extension of the group's lowest unfinished diagonal ----------------------------------------------
    {
      const bool sn = phase == PH_SNAKE;
      if (ballot64(sn)) {
        const int L = __builtin_ctz(gm | 0x100u);   // (groups outside SNAKE: lane 8 = the next group's lane 0, looked at and dropped)
        const int xs = group_lane(x, gb4, L), ys = group_lane(y, gb4, L);
        const int rem = min(q_len - xs, t_len - ys);
        const int off = gl * SL;
        // e = length of the run if it ends inside this lane's piece (a mismatch, or the end of a sequence), GL * SL if it does not:
        // the run's length is the minimum over the group (the pieces before the first such lane match throughout)
        int e = GL * SL;
        if (sn) {
          e = off;   // (a piece beyond the end of a sequence: the run ends at its start at the latest)
          if (off < rem) {
            int m;
            if (PACKED) {   // 32 bases: three dwords of either pack (one 16-byte load at a dword address), two funnel shifts each
              const uint32_t xq = qo + (uint32_t)(xs + off), yt = to + (uint32_t)(ys + off);
              uint4 qd, td;
              __builtin_memcpy(&qd, q + (xq >> 4) * 4, 16), __builtin_memcpy(&td, t + (yt >> 4) * 4, 16);
              const uint32_t qsh = (xq & 15) << 1, tsh = (yt & 15) << 1;
              const uint32_t d0 = __builtin_amdgcn_alignbit(qd.y, qd.x, qsh) ^ __builtin_amdgcn_alignbit(td.y, td.x, tsh);
              const uint32_t d1 = __builtin_amdgcn_alignbit(qd.z, qd.y, qsh) ^ __builtin_amdgcn_alignbit(td.z, td.y, tsh);
              // (both halves at once: written as d0 ? .. : d1 ? .. the compiler loads the third dwords only behind the test of d0 -- a
              //  second, dependent trip to the cache in most extensions, in a kernel whose wavefronts wait 60 % of their time)
              m = (int)min(min(ffbl_or_ones(d0), ffbl_or_ones(d1) | 32u) >> 1, 32u);   // (both halves looked at, no test of either)
            } else {  // 16 codes with one 16-byte load per sequence (half the vector-memory instructions of two 8-byte ones)
              const U128 qa = load_u128_unaligned(q + xs + off), ta = load_u128_unaligned(t + ys + off);
              m = match8(qa.lo, ta.lo, qs, ts);
              if (m == 8) m += match8(qa.hi, ta.hi, qs, ts);
            }
            m = (int)min((uint32_t)m, (uint32_t)(rem - off));   // (off < rem)
            e = m < SL ? off + m : GL * SL;
          }
        }
        const int ext = group_min8_i32(e);
        if (sn) {
          if (gl == L) x += ext, y += ext;
          if (ext < GL * SL || ext >= rem) gm &= gm - 1;  // mismatch found or an end reached: this diagonal is done
          if (!gm) phase = PH_END;
        }
      }
    }

    // ---- END of the round: the order-dependent side results, lowest k first (DWmatch.c:142-164) ------------------------
    {
      const bool e = phase == PH_END;
      const int ext = x - x1;
      const bool hit = e && active && (x >= q_len || y >= t_len);
      const uint64_t hitw = ballot64(hit);
      int hl = GL;
      uint32_t hitm = 0;
      if (hitw) {   // (once per candidate)
        hitm = group_bits<GL>(hitw, gbase);
        if (hitm) hl = __builtin_ctz(hitm);
        asm volatile("" : "+v"(hl));   // (keeps the branch: folded into selects the four instructions run in every iteration)
      }
      const bool valid = e && active && gl <= hl;
      {  // first extension > 16 fixes q_bgn/t_bgn once (DWmatch.c:142-146)
        const uint64_t sw = ballot64((valid && !started ? ext : 0) > 16);   // (a select + ONE compare: the compiler turns a ballot of
                                                                           // combined predicates into select 0/1 + compare on top of them)
        if (sw) {
          const uint32_t m = group_bits<GL>(sw, gbase);
          const int l = m ? __builtin_ctz(m) : 0;
          const int bx = group_lane(x1, gb4, l), by = group_lane(y1, gb4, l);
          if (m) q_bgn = bx, t_bgn = by, started = true;
        }
      }
      const int ev = valid ? ext : -1;
      if (ballot64(ev > (int)longest)) {  // strictly longer extension (DWmatch.c:148-152)
        const int mx = group_max_i32<GL>(ev);
        const uint32_t m = group_bits<GL>(ballot64(valid && ext == mx), gbase);
        const int l = m ? __builtin_ctz(m) : 0;
        const int ex = group_lane(x, gb4, l), ey = group_lane(y, gb4, l);
        if (e && mx >= 0 && (uint32_t)mx > longest) longest = (uint32_t)mx, q_m_end = ex, t_m_end = ey;
      }
      if (valid) V[k & mask] = (VT)x;
      const int u = x + y;
      {
        const int s = group_max_i32<GL>(valid ? u : -1);
        if (e) best_m = max(best_m, s);
      }
      bool matched = false;
      if (hitw) {
        const int ex = group_lane(x, gb4, hl & (GL - 1)), ey = group_lane(y, gb4, hl & (GL - 1));
        if (e && hitm) {  // DWmatch.c:185-194
          matched = true;
          if (gl == 0) {
            pgx_match r;
            r.q_bgn = q_bgn, r.t_bgn = t_bgn, r.q_end = ex, r.t_end = ey, r.dist = d;
            r.m_size = (ex - q_bgn + ey - t_bgn + 2 * d) / 2;
            r.t_m_end = t_m_end, r.q_m_end = q_m_end;
            out[a] = r;
          }
        }
      }
      // The band update (DWmatch.c:166-183) of a step whose diagonals all sat in this one round -- 98.7 % of the steps -- right here, from
      // the registers: U = x + y of the group's lanes, the qualifying diagonals as a mask, its lowest and highest bit.  (hit lanes: the
      // candidate is finished, the update is not looked at.)  Wider bands go through BAND below, a round of GL diagonals per iteration.
      const uint32_t qm = group_bits<GL>(ballot64((e && active ? u : INT32_MIN) >= best_m - band), gbase);
      if (e) {
        if (matched) {
          phase = PH_FETCH;
        } else if (nk <= GL) {
          const int lo = qm ? min_k + 2 * __builtin_ctz(qm) : max_k, hi = qm ? min_k + 2 * (31 - __builtin_clz(qm)) : min_k;
          max_k = hi + 1, min_k = lo - 1, ++d, phase = PH_STEP;
        } else {
          base += GL;
          phase = PH_ROUND;
          if (base >= nk) bbase = 0, new_min = max_k, new_max = min_k, phase = PH_BAND;
        }
      }
    }
    PH_SYNC();

    // ---- BAND: one round of the band update (DWmatch.c:166-183) ------------------------------------------------------
    {
      const bool bnd = phase == PH_BAND;
      if (ballot64(bnd)) {
        const int thr = best_m - band;
        const int j = bbase + gl;
        const int k2 = min_k + 2 * j;
        int u = 0;
        if (bnd && j < nk) u = 2 * (int)V[k2 & mask] - k2;   // (nk > GL: U of the earlier rounds' diagonals is not in the registers)
        const uint32_t m = group_bits<GL>(ballot64((bnd && j < nk ? u : INT32_MIN) >= thr), gbase);
        if (bnd) {
          if (m) {
            new_min = min(new_min, min_k + 2 * (bbase + __builtin_ctz(m)));
            new_max = max(new_max, min_k + 2 * (bbase + 31 - __builtin_clz(m)));
          }
          bbase += GL;
          if (bbase >= nk) max_k = new_max + 1, min_k = new_min - 1, ++d, phase = PH_STEP;
        }
      }
    }
  }
}

// sort keys of an order list: the rank of the query read in the packs' layout, the request number as the value; *n_out = n
__global__ void k_order_keys(const pgx_align_key *__restrict__ keys, uint32_t n, const uint32_t *__restrict__ prank, uint32_t *__restrict__ k,
                             uint32_t *__restrict__ v, uint32_t *__restrict__ n_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_out = n;
  if (i < n) k[i] = prank[keys[i].rid0], v[i] = i;
}

// The 2-bit packs ahead of the first large launch: run_overlap calls this while the GPU would otherwise wait for the host's outer
// table, so the first stage's k_pack2 (1.6 ms at 4.5 Gbases) is off the critical path; later stages find the packs in place
// (pgx_pack.hip: they are kept with the database).
void dev_align_prepare(const pgx_seqdb *db) {
  const char *pm = getenv("PGX_ALIGN_PACKED_MIN");
  if (db->max_rlen > 65535u || (pm && atol(pm) < 0)) return;
  (void)seq_packs(db);
}

// Dispatch (each form bit-exact against the oracle: tests/test_gpu_parity.py::test_align_variants_vs_oracle):
//   n <= small     : k_align1, a wavefront per candidate -- the replay's tail rounds, where latency is everything
//   large launches : k_align_ph<8, u16, packed> over the 2-bit packs + k_align1_list for what it hands on (reads with bytes that have no
//                    2-bit code, stragglers past the iteration budget); without packs (no HBM for them, PGX_ALIGN_PACKED_MIN < 0)
//                    k_align_ph<8, u16> on the seqdb bytes; with a read beyond 65,535 bases (16-bit V ring too narrow) k_align4<8, int32>
void dev_align(const pgx_seqdb *db, const pgx_align_key *d_keys, size_t n, int band, pgx_match *d_out, int tail_batch) {
  if (n == 0) return;
  // (the knobs are read per call: the parity tests walk every kernel variant inside one process)
  // launches up to this many alignments take a wavefront per candidate (k_align1).  On uniform candidates the crossover is ~14 k
  // (tools/alignlat.py: 13,000) -- but the mid-size launches of a stage are its TAIL sweeps (tail_batch: every request batch of the
  // device replay after the first), and in repeat-rich sets those are mostly long, wide-band alignments that the 8-lane groups of
  // k_align_ph first run to their iteration budget and then hand on: 60,000 for them = c4s 423 -> 417 ms per step, c5s 633 -> 628,
  // c3 unchanged (its second batch holds 71 k).  With the work counter's floor gone (round 4) re-swept on c4s: 13,000 -> 315.0 ms per step,
  // 30,000 -> 313.3, 60,000 -> 307.2, 120,000 -> 303.3, 250,000 and more -> 300.4-302.9 (its largest tail batch); full-size c4: 60,000 -> 8.56 s,
  // 250,000 -> 8.51, 1,000,000 -> 8.67 (the sweep-2 batches of ~0.8 M belong to the grouped kernel) -- but c3, whose SECOND batch is 71 k ordinary
  // candidates (wrong type guesses, not stragglers), loses 2 ms of 100 with them on k_align1: 250,000 from the third batch on, 60,000 for the
  // second.  PGX_ALIGN_SMALL overrides all three.
  const long small_max = getenv("PGX_ALIGN_SMALL") ? atol(getenv("PGX_ALIGN_SMALL")) : tail_batch >= 2 ? 250000 : tail_batch ? 60000 : 13000;
  KernelTimer tm((long)n <= small_max ? "align1" : "align", n);
  int ring = 64;
  while (ring < 2 * band + 8) ring <<= 1;
  hipStream_t st = ctx().stream;
  // the one-candidate-per-wavefront kernels read the 2-bit packs too (round 6) where the database has them and none of its reads lacks 2-bit codes
  // (n_flagged_reads: such candidates compare nibbles, DWmatch.c:136-137); PGX_ALIGN1_PACKED=0 keeps them on the bytes while those are there
  const bool one_packed = seq_packs_valid(db) && db->n_flagged_reads == 0 && db->max_rlen <= 65535u &&
                          (!db->d_seq.p || !(getenv("PGX_ALIGN1_PACKED") && atoi(getenv("PGX_ALIGN1_PACKED")) == 0));
  PGX_REQUIRE(db->d_seq.p || one_packed, PGX_ESTATE, "the seqdb's bytes were released (pgx_seqdb_release_bytes) and its packs are gone");
  if ((long)n <= small_max) {
    if (one_packed)
      hipLaunchKernelGGL(k_align1<true>, dim3((unsigned)n), dim3(64), ring * sizeof(int32_t), st, reinterpret_cast<const uint8_t *>(db->d_pack.p), db->d_poff.p,
                         db->d_rlen.p, d_keys, (uint32_t)n, band, ring, d_out);
    else
      hipLaunchKernelGGL(k_align1<false>, dim3((unsigned)n), dim3(64), ring * sizeof(int32_t), st, db->d_seq.p, db->d_roff.p, db->d_rlen.p, d_keys,
                         (uint32_t)n, band, ring, d_out);
    PGX_HIP(hipGetLastError());
    return;
  }
  uint32_t *counter = ws<uint32_t>("align.counter", 8);
  PGX_HIP(hipMemsetAsync(counter, 0, 8 * sizeof(uint32_t), st));
  if (db->max_rlen > 65535u) {   // x <= read length must fit the V ring's element: 32-bit rings, 20 wavefronts per CU
    const unsigned grid = (unsigned)std::min<size_t>((n + 7) / 8, (size_t)ctx().num_cu * 20);
    hipLaunchKernelGGL((k_align4<8, int32_t>), dim3(grid), dim3(64), 8 * ring * sizeof(int32_t), st, db->d_seq.p, db->d_roff.p, db->d_rlen.p,
                       d_keys, (uint32_t)n, band, ring, d_out, counter);
    PGX_HIP(hipGetLastError());
    return;
  }
  // k_align_ph: workgroups of NW wavefronts (a V ring of `ring` u16 per 8-lane group), as many as give a CU its 32 wavefronts
  const size_t lds_wave = (size_t)8 * ring * sizeof(uint16_t);
  const unsigned waves_cu = (unsigned)std::min<size_t>(32, (158u << 10) / lds_wave);   // 32 = all the wavefronts a CU holds; alignment kernels per c3 step
                                                                                       // with the chunked work counter: 16 -> 65.3 ms, 20 -> 55.9, 24 -> 50.1, 28 -> 46.3, 32 -> 43.7
  const long nw_env = getenv("PGX_ALIGN_NW") ? atol(getenv("PGX_ALIGN_NW")) : 0;
  unsigned NW = 1;   // the widest workgroup that does not cost the CU wavefronts (ring 256: 16 x 2; ring 512: 19 fit -> 16 x 1)
  for (unsigned c = 16; c >= 1; c >>= 1)
    if (c <= waves_cu && (waves_cu / c) * c > (waves_cu / NW) * NW) NW = c;
  if (waves_cu >= 16 && (waves_cu / 16) * 16 >= (waves_cu / NW) * NW) NW = 16;
  if (nw_env >= 1 && nw_env <= 16 && (unsigned)nw_env <= waves_cu) NW = (unsigned)nw_env;
  const size_t lds = lds_wave * NW;
  const unsigned wg_cu = waves_cu / NW;
  // the packs are built by the first launch of at least PGX_ALIGN_PACKED_MIN alignments (default 100,000) and serve every later launch
  // on this database; < 0: never.
  const char *pm = getenv("PGX_ALIGN_PACKED_MIN");
  const long packed_min = pm ? atol(pm) : 100000;
  const uint32_t *packs = (packed_min >= 0 && ((long)n >= packed_min || seq_packs_valid(db))) ? seq_packs(db) : nullptr;
  // segment = the run of requests a workgroup's wavefronts share (a multiple of the chunk of 8): long enough that the workgroup stays at
  // one place of the list, short enough that the last segments of a launch do not leave the other CUs idle
  const size_t n_wg = (size_t)ctx().num_cu * wg_cu;
  auto segment = [&](size_t cnt) {
    const long se = getenv("PGX_ALIGN_SEG") ? atol(getenv("PGX_ALIGN_SEG")) : 0;
    size_t sg = se > 0 ? (size_t)se : std::min<size_t>(512, std::max<size_t>(8 * NW, cnt / (n_wg * 16)));
    return (uint32_t)((sg + 7) & ~(size_t)7);
  };
  const uint32_t seg = segment(n);
  const unsigned grid = (unsigned)std::min<size_t>((n + seg - 1) / seg, n_wg);
  if (packs) {
    const uint32_t iter_limit = getenv("PGX_ALIGN_ITER_LIMIT") ? (uint32_t)atol(getenv("PGX_ALIGN_ITER_LIMIT")) : 2500u;   // (0: no hand-on of stragglers)
    // escalation block: [0] stragglers handed on, [1] candidates that touch a read without 2-bit codes, [2] the byte-wise launch's work
    // counter, [3] the number of requests (for the order list), [4 .. 4 + n) the stragglers, [4 + n .. 4 + 2 n) the others
    uint32_t *esc = ws<uint32_t>("align.esc", 2 * n + 4);
    PGX_HIP(hipMemsetAsync(esc, 0, 3 * sizeof(uint32_t), st));
    // The requests in the order of the packs' layout (round 6): sorted by the rank of the query read -- stable, so the requests of one read
    // stay in the walk's order -- when the layout follows the locus keys; results are addressed by request number, nothing downstream
    // sees the order.  15.7 M requests: ~1 ms of a 150-190 ms launch.
    const uint32_t *order = nullptr;
    const long order_min = getenv("PGX_ALIGN_ORDER_MIN") ? atol(getenv("PGX_ALIGN_ORDER_MIN")) : 200000;
    if (db->locus_ordered && order_min >= 0 && (long)n >= order_min) {
      const uint32_t nn = (uint32_t)n;
      uint32_t *k_in = ws<uint32_t>("align.ord_kin", n), *k_out = ws<uint32_t>("align.ord_kout", n), *v_in = ws<uint32_t>("align.ord_vin", n),
               *v_out = ws<uint32_t>("align.ord_vout", n);
      hipLaunchKernelGGL(k_order_keys, dim3((nn + 255) / 256), dim3(256), 0, st, d_keys, nn, db->d_prank.p, k_in, v_in, esc + 3);
      int bits = 1;
      while (bits < 32 && (db->rlen_by_rid.size() >> bits)) ++bits;
      size_t tb = 0;
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)nn, 0, bits, st));
      void *tmp = ws_raw("align.ord_tmp", tb + 256);
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k_in, k_out, v_in, v_out, (int)nn, 0, bits, st));
      order = v_out;
    }
    hipLaunchKernelGGL((k_align_ph<8, uint16_t, true>), dim3(grid), dim3(64 * NW), lds, st, reinterpret_cast<const uint8_t *>(packs), db->d_poff.p,
                       db->d_rlen.p, d_keys, (uint32_t)n, band, ring, d_out, counter, order ? esc + 3 : (const uint32_t *)nullptr, order, esc,
                       esc + 4, db->d_nflag.p, seg, iter_limit);
    // candidates on reads with ambiguous bases: the same phase machine on the seqdb bytes, eight per wavefront, from their list (round 3 gave
    // each a wavefront of its own through k_align1_list: 5 % of the reads flagged = +64 % alignment time at c3).  Only when the database
    // holds such a read at all.
    if (db->n_flagged_reads) {
      const uint32_t seg2 = segment(n / 16 + 1);
      hipLaunchKernelGGL((k_align_ph<8, uint16_t, false>), dim3((unsigned)std::min<size_t>(n / (64 * NW) + 8, n_wg)), dim3(64 * NW), lds, st,
                         db->d_seq.p, db->d_roff.p, db->d_rlen.p, d_keys, (uint32_t)n, band, ring, d_out, esc + 2, esc + 1, esc + 4 + n, esc, esc + 4,
                         (const uint32_t *)nullptr, seg2, iter_limit);
    }
    // the stragglers of either launch, a wavefront per candidate, from the list
    const dim3 g1((unsigned)std::min<size_t>(std::max<size_t>(n / 256, 256), (size_t)ctx().num_cu * 32));
    if (one_packed)
      hipLaunchKernelGGL(k_align1_list<true>, g1, dim3(64), ring * sizeof(int32_t), st, reinterpret_cast<const uint8_t *>(packs), db->d_poff.p, db->d_rlen.p,
                         d_keys, esc, esc + 4, band, ring, d_out);
    else
      hipLaunchKernelGGL(k_align1_list<false>, g1, dim3(64), ring * sizeof(int32_t), st, db->d_seq.p, db->d_roff.p, db->d_rlen.p, d_keys, esc, esc + 4, band,
                         ring, d_out);
  } else {
    hipLaunchKernelGGL((k_align_ph<8, uint16_t, false>), dim3(grid), dim3(64 * NW), lds, st, db->d_seq.p, db->d_roff.p, db->d_rlen.p, d_keys,
                       (uint32_t)n, band, ring, d_out, counter, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                       (uint32_t *)nullptr, (const uint32_t *)nullptr, seg, 0u);
  }
  PGX_HIP(hipGetLastError());
}

}  // namespace pgx
