// pgx_align_lane.hip -- banded O(ND) confirmation (ovlp_match, /root/reference/src/DWmatch.c:66-204), ONE CANDIDATE PER LANE.
//
// Round 3.  k_align_ph (pgx_align.hip) gives a candidate an 8-lane group: ~3.6 of the 8 lanes hold a live diagonal, and every
// order-dependent side result of DWmatch.c:142-164 costs group ballots, DPP reductions and cross-lane reads, all of them half-rate
// opcodes on gfx950 (profiles/r03_valu_issue.txt): 30 VALU wavefront-instructions per candidate-step, 66 M alignments/s, the SIMDs
// 93 % busy issuing.  Here a LANE runs the reference's loops for its own candidate exactly as written -- k ascending, one diagonal
// at a time -- so all bookkeeping is plain per-lane arithmetic, no lane is idle by construction, and a wavefront-instruction serves
// 64 candidates.  What makes that possible:
//   * the reads as 2-bit packs (k_pack2: one pass over the seqdb per overlap stage, both strands; the high nibbles of the seqdb
//     ARE the reverse complement stored forward, src/shmr_utils.c:44-51): a 16-byte load is 64 bases, a compare of 32 bases is
//     two funnel shifts, two XORs and a find-first-bit per sequence pair;
//   * a PRIVATE WINDOW of each sequence in LDS per lane (256 bases = 16 dwords + 2 mirrored, dword-interleaved over the lanes:
//     lane l only ever touches bank l, no conflicts whatever the lanes' positions): the diagonals of a candidate live within
//     ~60 bases of its front, so nearly every compare is served from LDS and the global traffic is one aligned 16-byte chunk per
//     64 bases of progress per sequence -- fetched THREE iterations ahead of need (the loop is unrolled three times around three
//     sets of landing registers; the compiler's vmcnt bookkeeping leaves the younger loads in flight), so no iteration waits for
//     memory in steady state;
//   * V as 16-bit values in a 32-slot private ring (reads <= 65,535 bases; a candidate whose band outgrows 32 slots, or that
//     meets a read with ambiguous bases, is handed to k_align_ph through the escalation list -- measured rare);
//   * the band update of DWmatch.c:166-183 without a second pass: new_max_k online (the last k with U >= best - band AT ITS TIME:
//     best only rises at a k that itself qualifies, so the last such k is exact), new_min_k from the U of the step's first three
//     diagonals kept in registers (a scan over the ring only when all three fall out);
//   * candidates are drawn 32 at a time per wavefront (one atomic per 32), each lane's NEXT descriptor is prefetched while it
//     works on the current one.
// Every lane is a little state machine (FETCH -> DIAG -> EXT -> ... ); an iteration of the wavefront = one 32-base compare for
// every lane that has a diagonal open, whatever d-step its candidate is in.
#include "pgx_internal.h"

namespace pgx {
namespace {

// ---- 2-bit packing of the seqdb, both strands -------------------------------------------------------------------------------
// word w of stream s (s = 0: low nibbles = the read as stored; 1: high nibbles = its reverse complement, stored forward) holds
// the codes of seqdb bytes 16 w .. 16 w + 15, base i in bits 2i..2i+1.  A byte that is not one of the four one-hot codes (an
// ambiguous base: nibble 0) packs as 0 and marks its read in nflag[] (such reads never enter the lane kernel).
__device__ __forceinline__ uint32_t pack4(uint32_t n) {   // four one-hot nibbles (one per byte) -> 8 bits
  const uint32_t c = ((n >> 1) & 0x07070707u) - ((n >> 3) & 0x01010101u);   // 1,2,4,8 -> 0,1,2,3 per byte; the masks keep the shifts from
                                                                           // leaking the next byte's low bits in (no borrow then: 4 - 1, x - 0)
  return (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xFFu;
}
__global__ __launch_bounds__(256) void k_pack2(const uint4 *__restrict__ seq, size_t nwords, uint32_t *__restrict__ p0, uint32_t *__restrict__ p1,
                                               const uint64_t *__restrict__ roff_sorted, const uint32_t *__restrict__ rid_sorted, uint32_t nreads,
                                               size_t nbytes, uint32_t *__restrict__ nflag) {
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = seq[w];
    const uint32_t M = 0x0F0F0F0Fu;
    const uint32_t l0 = v.x & M, l1 = v.y & M, l2 = v.z & M, l3 = v.w & M;
    const uint32_t h0 = (v.x >> 4) & M, h1 = (v.y >> 4) & M, h2 = (v.z >> 4) & M, h3 = (v.w >> 4) & M;
    p0[w] = pack4(l0) | (pack4(l1) << 8) | (pack4(l2) << 16) | (pack4(l3) << 24);
    p1[w] = pack4(h0) | (pack4(h1) << 8) | (pack4(h2) << 16) | (pack4(h3) << 24);
    // a zero nibble inside the database = an ambiguous base (zero bytes past the end are padding)
    auto haszero = [](uint32_t x) { return ((x - 0x01010101u) & ~x & 0x80808080u) != 0; };
    if ((haszero(l0) || haszero(l1) || haszero(l2) || haszero(l3)) && w * 16 < nbytes) {
      const uint32_t ws[4] = {l0, l1, l2, l3};
      for (int j = 0; j < 16; ++j) {
        const size_t pos = w * 16 + j;
        if (pos >= nbytes || ((ws[j >> 2] >> (8 * (j & 3))) & 0xFu) != 0) continue;
        uint32_t lo = 0, hi = nreads;   // the read that holds byte `pos`: last roff <= pos
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (roff_sorted[mid] <= pos) lo = mid; else hi = mid;
        }
        if (nreads) atomicOr(&nflag[rid_sorted[lo]], 1u);
      }
    }
  }
}

// ---- candidate descriptors ----------------------------------------------------------------------------------------------------
struct LaneDesc {           // 32 bytes
  uint32_t qg_lo, qg_hi;    // global base position of query base 0 (seqdb byte offset); bit 31 of qg_hi: strand
  uint32_t tg_lo, tg_hi;    // the same for the target
  int32_t q_len, t_len;
  int32_t max_d;            // (int)(0.3 * (q_len + t_len)), DWmatch.c:96
  uint32_t flags;           // 1: not for the lane kernel (a read with ambiguous bases)
};
__global__ __launch_bounds__(256) void k_lane_prep(const pgx_align_key *__restrict__ keys, uint32_t n, const uint64_t *__restrict__ roff,
                                                   const uint32_t *__restrict__ rlen, const uint32_t *__restrict__ nflag,
                                                   LaneDesc *__restrict__ desc) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const pgx_align_key key = keys[a];
  const uint64_t qg = roff[key.rid0] + key.q_off, tg = roff[key.rid1];
  LaneDesc d;
  d.qg_lo = (uint32_t)qg, d.qg_hi = (uint32_t)(qg >> 32) | (key.dir0 ? 0x80000000u : 0u);
  d.tg_lo = (uint32_t)tg, d.tg_hi = (uint32_t)(tg >> 32) | (key.dir1 ? 0x80000000u : 0u);
  d.q_len = (int)(rlen[key.rid0] - key.q_off), d.t_len = (int)rlen[key.rid1];
  d.max_d = (int)(0.3 * (double)(d.q_len + d.t_len));
  d.flags = (nflag[key.rid0] | nflag[key.rid1]) & 1u;
  desc[a] = d;
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------
enum { LP_FETCH = 0, LP_DIAG = 1, LP_EXT = 2, LP_SCAN = 3, LP_DONE = 4 };
#ifndef PGX_LANE_W
#define PGX_LANE_W 256
#endif
constexpr int LANE_W = PGX_LANE_W;             // bases a window holds (a power of two >= 128)
constexpr int LANE_WD = LANE_W / 16;        // ... in dwords
constexpr int LANE_QD = LANE_WD + 2;        // dwords of a sequence window: the ring + 2 mirrored
constexpr int LANE_VD = 16;                 // dwords of the V ring: 32 slots of 16 bits
constexpr int LANE_DW = 2 * LANE_QD + LANE_VD;   // 52 dwords = 208 bytes per lane, 13,312 bytes per wavefront
constexpr int LANE_RING = 32;               // V slots
constexpr int LANE_BATCH = 32;              // candidates a wavefront draws per atomic

__device__ __forceinline__ uint32_t ffbl(uint32_t v) {   // v_ffbl_b32: index of the lowest set bit, 0xFFFFFFFF for 0 (no fix-up wanted)
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}

struct Landing {   // one in-flight 64-base chunk per sequence
  uint4 q, t;
  bool pq, pt;
};

__global__ __launch_bounds__(64) void k_align_lane(const uint32_t *__restrict__ pack0, const uint32_t *__restrict__ pack1,
                                                   const LaneDesc *__restrict__ desc, uint32_t n, int band, pgx_match *__restrict__ out,
                                                   uint32_t *__restrict__ counter, uint32_t *__restrict__ esc_n, uint32_t *__restrict__ esc_list,
                                                   uint32_t *__restrict__ why) {   // why: PGX_TRACE histogram of the hand-on reasons (or null)
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  uint32_t *const Q = lds + lane, *const T = lds + LANE_QD * 64 + lane;             // dword d of the window at [d * 64]
  uint16_t *const V16 = reinterpret_cast<uint16_t *>(lds + 2 * LANE_QD * 64 + lane);  // slot s at [(s >> 1) * 128 + (s & 1)]
  const int band_size = band * 2;

  // ---- per-lane state ----
  int phase = LP_FETCH;
  uint32_t a = 0;
  const uint4 *qsrc = reinterpret_cast<const uint4 *>(pack0), *tsrc = qsrc;   // chunk 0 of the candidate's windows (16-byte chunks of 64 bases)
  int qoff = 0, toff = 0;                        // position of base 0 inside chunk 0 (0..63): xr = x + qoff
  int q_len = 0, t_len = 0, max_d = 0, d = 0;
  int best_m = -1, min_k = 0, max_k = 0, k = 0, cmax = 0;
  int x = 0, y = 0, x1 = 0;
  int ulo0 = 0, ulo1 = 0, ulo2 = 0;
  uint32_t longest = 0;
  bool started = false;
  int q_bgn = 0, t_bgn = 0, q_m_end = 0, t_m_end = 0;
  int qhi = 0, qreq = 0, thi = 0, treq = 0;      // windows: bases [..hi - 256, ..hi) are in LDS, [..hi, ..req) in flight (units: bases from chunk 0)
  int fx = 0, fy = 0;                            // the furthest x / y any diagonal has reached
  int lowx = 0, lowy = 0;                        // no diagonal still to be worked on starts below these: the windows may not move past them
  bool susp = false, resuming = false;           // a diagonal of this step is set aside (its match ran out of the pinned windows) / being finished
  int sk = 0, sx = 0, sx1 = 0;
  int slo = 0, shi = 0, nmin = 0, nmax = 0;      // LP_SCAN: the band update as a scan from both ends
  bool flo = false, fhi = false;
  int skip = 0;                                  // landings to ignore after a change of candidate
  // the next candidate, prefetched
  int nxt_state = 0;                             // 0: none requested, 1: descriptor requested / loaded, 2: no candidates left
  uint32_t nxt_a = 0;
  uint4 nd0 = {0, 0, 0, 0}, nd1 = {0, 0, 0, 0};
  // the wavefront's pool of tickets (uniform)
  uint32_t pool_next = 0, pool_left = 0;
  bool exhausted = false;

  // three sets of landing registers, used round-robin by the three copies of the loop body (fully unrolled: every index below is
  // a constant, and the state stays in plain locals -- a by-reference lambda made the compiler keep some of it in scratch)
  Landing Ls[3] = {{{0, 0, 0, 0}, {0, 0, 0, 0}, false, false}, {{0, 0, 0, 0}, {0, 0, 0, 0}, false, false}, {{0, 0, 0, 0}, {0, 0, 0, 0}, false, false}};
  for (;;) {
#pragma unroll
  for (int ui = 0; ui < 3; ++ui) {
    Landing &L = Ls[ui];
    // ---- A. a chunk requested three iterations ago lands in the window ------------------------------------------------------
    if (skip > 0) --skip, L.pq = L.pt = false;   // (requested for the lane's previous candidate)
    if (L.pq) {
      const int s = (qhi >> 4) & (LANE_WD - 1);   // 4 dwords at ring dwords s..s+3 (s is a multiple of 4)
      Q[(s + 0) * 64] = L.q.x, Q[(s + 1) * 64] = L.q.y, Q[(s + 2) * 64] = L.q.z, Q[(s + 3) * 64] = L.q.w;
      if (s == 0) Q[LANE_WD * 64] = L.q.x, Q[(LANE_WD + 1) * 64] = L.q.y;
      qhi += 64;
    }
    if (L.pt) {
      const int s = (thi >> 4) & (LANE_WD - 1);
      T[(s + 0) * 64] = L.t.x, T[(s + 1) * 64] = L.t.y, T[(s + 2) * 64] = L.t.z, T[(s + 3) * 64] = L.t.w;
      if (s == 0) T[LANE_WD * 64] = L.t.x, T[(LANE_WD + 1) * 64] = L.t.y;
      thi += 64;
    }
    L.pq = L.pt = false;

    // ---- B. lanes without a candidate take the prefetched one and ask for the next ---------------------------------------------
    const uint64_t fm = __builtin_amdgcn_ballot_w64(phase == LP_FETCH);
    if (fm) {
      if (phase == LP_FETCH && nxt_state == 1) {
        a = nxt_a;
        const uint64_t qg = ((uint64_t)(nd0.y & 0x7FFFFFFFu) << 32) | nd0.x, tg = ((uint64_t)(nd0.w & 0x7FFFFFFFu) << 32) | nd0.z;
        qsrc = reinterpret_cast<const uint4 *>((nd0.y >> 31) ? pack1 : pack0) + (qg >> 6);
        tsrc = reinterpret_cast<const uint4 *>((nd0.w >> 31) ? pack1 : pack0) + (tg >> 6);
        qoff = (int)(qg & 63), toff = (int)(tg & 63);
        q_len = (int)nd1.x, t_len = (int)nd1.y, max_d = (int)nd1.z;
        d = 0, best_m = -1, min_k = 0, max_k = 0, k = 0, cmax = 0, longest = 0, started = false;
        q_bgn = t_bgn = q_m_end = t_m_end = 0;
        qhi = qreq = thi = treq = 0, fx = fy = 0, lowx = lowy = 0;
        susp = resuming = false;
        skip = 2;               // the two chunks still in flight belong to the previous candidate
        V16[0 * 128 + 1] = 0;   // slot 1: the only slot read before it is written (d = 0 reads V[k + 1] = V[1])
        nxt_state = 0;
        phase = LP_DIAG;
        if (nd1.w & 1u) {       // a read with ambiguous bases: the byte-wise kernel takes it
          if (why) atomicAdd(&why[6], 1u);
          esc_list[atomicAdd(esc_n, 1u)] = a;
          phase = LP_FETCH;
        } else if (max_d <= 0) {   // the d-loop never runs (DWmatch.c:118): no match
          pgx_match r;
          r.m_size = 0, r.dist = 0, r.q_bgn = 0, r.q_end = 0, r.t_bgn = 0, r.t_end = 0, r.t_m_end = 0, r.q_m_end = 0;
          out[a] = r;
          phase = LP_FETCH;
        }
      }
      // tickets: from the wavefront's pool, refilled LANE_BATCH at a time
      const uint64_t wm = __builtin_amdgcn_ballot_w64(phase == LP_FETCH && nxt_state == 0);
      if (wm) {
        if (pool_left == 0 && !exhausted) {
          uint32_t base = 0;
          if (lane == __builtin_ctzll(wm)) base = atomicAdd(counter, (uint32_t)LANE_BATCH);
          base = __builtin_amdgcn_readlane(base, __builtin_ctzll(wm));
          pool_next = base, pool_left = base < n ? min((uint32_t)LANE_BATCH, n - base) : 0u;
          exhausted = pool_left == 0;
        }
        if (phase == LP_FETCH && nxt_state == 0) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wm, 0u));
          if (rank < pool_left) {
            nxt_a = pool_next + rank;
            const uint4 *dp = reinterpret_cast<const uint4 *>(desc + nxt_a);
            nd0 = dp[0], nd1 = dp[1];
            nxt_state = 1;
          } else if (exhausted) {
            nxt_state = 2;
          }
        }
        const uint32_t taken = min((uint32_t)__builtin_popcountll(wm), pool_left);
        pool_next += taken, pool_left -= taken;
      }
      if (phase == LP_FETCH && nxt_state == 2) phase = LP_DONE;
    }

    // ---- C. a new diagonal: its start point from the previous step's V (DWmatch.c:124-133) --------------------------------------
    if (phase == LP_DIAG) {
      const int sa = (k - 1) & (LANE_RING - 1), sb = (k + 1) & (LANE_RING - 1);
      const int va = (int)V16[(sa >> 1) * 128 + (sa & 1)], vb = (int)V16[(sb >> 1) * 128 + (sb & 1)];
      x = (k == min_k || (k != max_k && va < vb)) ? vb : va + 1;
      y = x - k;
      x1 = x;
      phase = LP_EXT;
    }

    // ---- D. one compare of up to 32 bases on the open diagonal (DWmatch.c:135-140) -------------------------------------------------
    bool fin = false, hand_on = false, set_aside = false;
    int reason = 0;
    if (phase == LP_EXT) {
      if (resuming) lowx = x, lowy = y;                   // only this diagonal is left of its step: the windows follow it
      const int rem = min(q_len - x, t_len - y);
      const int xr = x + qoff, yr = y + toff;
      const int nav = min(min(qhi - xr, thi - yr), 32);   // bases both windows hold from here on
      hand_on = xr < qhi - LANE_W || yr < thi - LANE_W;         // behind a window (the pins below make this an assertion): handed on
      reason = 1;
      const int qi = (xr >> 4) & (LANE_WD - 1), ti = (yr >> 4) & (LANE_WD - 1);
      const uint32_t q0 = Q[qi * 64], q1 = Q[(qi + 1) * 64], q2 = Q[(qi + 2) * 64];
      const uint32_t t0 = T[ti * 64], t1 = T[(ti + 1) * 64], t2 = T[(ti + 2) * 64];
      const int n = max(min(nav, rem), 0);
      const uint32_t qs = (uint32_t)(xr & 15) << 1, ts = (uint32_t)(yr & 15) << 1;
      const uint32_t qa = __builtin_amdgcn_alignbit(q1, q0, qs), qb = __builtin_amdgcn_alignbit(q2, q1, qs);
      const uint32_t ta = __builtin_amdgcn_alignbit(t1, t0, ts), tb = __builtin_amdgcn_alignbit(t2, t1, ts);
      const uint32_t d0 = qa ^ ta, d1 = qb ^ tb;
      // first differing base: ffbl gives -1 for 0, so the unsigned minimum falls through to the next word / to "none"
      const uint32_t f = min(ffbl(d0), ffbl(d1) | 32u) >> 1;
      const int m = (int)min(f, (uint32_t)n);
      x += m, y += m;
      fin = (m < n) || (m >= rem);   // a mismatch inside what was compared, or an end reached; else the extension goes on
      fx = max(fx, x), fy = max(fy, y);
#ifdef PGX_LANE_TRACE
      if (why && a == 0) {
        const uint32_t slot = atomicAdd(&why[7], 1u);
        if (slot < 256) {
          uint32_t *tr = why + 8 + slot * 12;
          tr[0] = d, tr[1] = k, tr[2] = x - m, tr[3] = y - m, tr[4] = m, tr[5] = n, tr[6] = qhi, tr[7] = thi, tr[8] = qa, tr[9] = ta, tr[10] = qoff, tr[11] = toff;
        }
      }
#endif
      // The windows are pinned to the lowest start point of the diagonals still to come (section G), so a long match can run out
      // of them: the furthest a window gets is the first multiple of 64 above low + 192.  The diagonal is then SET ASIDE, the rest
      // of the step is done first (those diagonals sit near the old front), and it is finished when nothing else needs the old data.
      const int qmax = ((lowx + qoff + LANE_W - 64) & ~63) + 64, tmax = ((lowy + toff + LANE_W - 64) & ~63) + 64;
      set_aside = !fin && n == 0 && (xr + m >= qmax || yr + m >= tmax);
    }
    if (set_aside) {
      if (susp || resuming || x - x1 <= 16) {
        hand_on = true, reason = 2;   // a second one in the same step, or one that has not even matched 17 bases: not worth the bookkeeping
      } else {
        susp = true, sk = k, sx = x, sx1 = x1;
        if (!started) q_bgn = x1, t_bgn = x1 - k, started = true;   // its extension is > 16 whatever its end (DWmatch.c:142-146, in k order)
        k += 2;
        phase = LP_DIAG;
        if (k > max_k) k = sk, x = sx, y = sx - sk, x1 = sx1, susp = false, resuming = true, phase = LP_EXT;   // it was the step's last
      }
    }

    // ---- E. the diagonal is complete: DWmatch.c:142-164, then the next k or the end of the step ------------------------------------
    bool rescan = false;
    if (fin && !hand_on) {
      const int ext = x - x1;
      if (!started && ext > 16) q_bgn = x1, t_bgn = x1 - k, started = true;
      if ((uint32_t)ext > longest) {
        // with a diagonal of lower k set aside, whether this one is the strictly longest depends on an extension not yet known
        if (susp) hand_on = true, reason = 3;
        else longest = (uint32_t)ext, q_m_end = x, t_m_end = y;
      }
      const int sv = k & (LANE_RING - 1);
      V16[(sv >> 1) * 128 + (sv & 1)] = (uint16_t)x;
      const int u = x + y;
      best_m = max(best_m, u);
      if (u >= best_m - band) cmax = k;
      const int idx = (k - min_k) >> 1;
      if (idx == 0) ulo0 = u;
      if (idx == 1) ulo1 = u;
      if (idx == 2) ulo2 = u;
      if (x >= q_len || y >= t_len) {   // matched (DWmatch.c:160-163,185-194)
        if (susp) {
          hand_on = true, reason = 4;   // the diagonal set aside comes first in k order and may reach an end too
        } else {
          pgx_match r;
          r.q_bgn = q_bgn, r.t_bgn = t_bgn, r.q_end = x, r.t_end = y, r.dist = d;
          r.m_size = (x - q_bgn + y - t_bgn + 2 * d) / 2;
          r.t_m_end = t_m_end, r.q_m_end = q_m_end;
          out[a] = r;
          phase = LP_FETCH;
        }
      } else if (resuming) {            // the step is complete now; its band update reads everything back (best_m was not final
        resuming = false;               // when the later diagonals were judged)
        rescan = true;
      } else {
        k += 2;
        phase = LP_DIAG;
        if (k > max_k) {
          if (susp) {                   // finish the diagonal that was set aside
            k = sk, x = sx, y = sx - sk, x1 = sx1, susp = false, resuming = true;
            phase = LP_EXT;
          } else {                      // band update (DWmatch.c:166-183)
            const int thr = best_m - band;
            const int nk = ((max_k - min_k) >> 1) + 1;
            int new_min = max_k;
            if (ulo0 >= thr) new_min = min_k;
            else if (nk >= 2 && ulo1 >= thr) new_min = min_k + 2;
            else if (nk >= 3 && ulo2 >= thr) new_min = min_k + 4;
            else if (nk > 3) rescan = true;
            if (!rescan) {
              nmin = new_min, nmax = cmax;
              k = INT32_MIN;            // "a step has ended": the boundary code below
            }
          }
        }
      }
    }
    if (hand_on) {
      if (why) atomicAdd(&why[reason], 1u);
      esc_list[atomicAdd(esc_n, 1u)] = a;
      phase = LP_FETCH;
      rescan = false;
    }

    // ---- F. the band update as a scan from both ends (only when the registers above do not settle it) -------------------------------
    if (rescan) slo = min_k, shi = max_k, flo = fhi = false, phase = LP_SCAN;
    if (phase == LP_SCAN) {
      const int thr = best_m - band;
      const int s0 = slo & (LANE_RING - 1), s1 = shi & (LANE_RING - 1);
      const int u0 = 2 * (int)V16[(s0 >> 1) * 128 + (s0 & 1)] - slo, u1 = 2 * (int)V16[(s1 >> 1) * 128 + (s1 & 1)] - shi;
      if (!flo) {
        if (u0 >= thr) nmin = slo, flo = true;
        else if (slo + 2 > max_k) nmin = max_k, flo = true;
        else slo += 2;
      }
      if (!fhi) {
        if (u1 >= thr) nmax = shi, fhi = true;
        else if (shi - 2 < min_k) nmax = min_k, fhi = true;
        else shi -= 2;
      }
      if (flo && fhi) k = INT32_MIN, phase = LP_DIAG;
    }

    // ---- the loop conditions of a new step (DWmatch.c:118-122,196-199) ---------------------------------------------------------------
    if (phase == LP_DIAG && k == INT32_MIN) {
      // every diagonal of the new step starts at a V (+ 1) of [nmin, nmax]; those with U >= best - band have x >= (best - band + k) / 2
      // and the ones in between are pulled along by their neighbours: with a margin, the lowest x / y the step can start at
      lowx = max(((best_m - band + nmin) >> 1) - 16, 0), lowy = max(((best_m - band - nmax) >> 1) - 16, 0);
      min_k = nmin - 1, max_k = nmax + 1, ++d;
      k = min_k, cmax = min_k;
      if (d >= max_d || max_k - min_k > band_size) {
        pgx_match r;
        r.m_size = 0, r.dist = 0, r.q_bgn = 0, r.q_end = 0, r.t_bgn = 0, r.t_end = 0;
        r.t_m_end = t_m_end, r.q_m_end = q_m_end;
        out[a] = r;
        phase = LP_FETCH;
      } else if (max_k - min_k + 4 > LANE_RING || max_k < min_k) {
        // the band outgrew the private V ring (or degenerated to an empty step): k_align_ph redoes the candidate from scratch
        if (why) atomicAdd(&why[5], 1u);
        esc_list[atomicAdd(esc_n, 1u)] = a;
        phase = LP_FETCH;
      }
    }

    // ---- G. request the next chunk of a window when the front comes within 96 bases (three compares) of what has been asked for,
    //         unless that chunk would overwrite bases a diagonal still to come may start at -------------------------------------------
    {
      const bool active = phase != LP_FETCH && phase != LP_DONE;
      const bool wq = active && (fx + qoff + 96 > qreq) && (qreq <= lowx + qoff + LANE_W - 64);
      const bool wt = active && (fy + toff + 96 > treq) && (treq <= lowy + toff + LANE_W - 64);
      // (issued by every lane, every iteration: a lane that needs nothing re-reads its last chunk -- a cache hit -- so that the
      // number of loads in flight is the same on every path and the compiler can leave exactly two iterations' worth outstanding)
      const int qc = wq ? (qreq >> 6) : max((qreq >> 6) - 1, 0), tc = wt ? (treq >> 6) : max((treq >> 6) - 1, 0);
      L.q = qsrc[qc];   // (a lane without a candidate reads chunk 0 of the pack: harmless, and the load stays unconditional)
      L.t = tsrc[tc];
      L.pq = wq, L.pt = wt;
      if (wq) qreq += 64;
      if (wt) treq += 64;
    }
  }
  // (the exit test sits here, once per three iterations, and nowhere inside the unrolled body: an exit edge in the middle makes the
  // compiler count the loads in flight along paths that skip them, and every wait degenerates to vmcnt(0..1))
  if (!__builtin_amdgcn_ballot_w64(phase != LP_DONE)) break;
  }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------
uint64_t &align_epoch() {
  static uint64_t e = 1;
  return e;
}

static size_t lane_pack_stride(const pgx_seqdb *db) { return (((db->nbytes + 1024) / 16) + 3) & ~(size_t)3; }
static void ensure_packed(const pgx_seqdb *db);
size_t seq_pack_stride(const pgx_seqdb *db) { return lane_pack_stride(db); }
bool seq_packs_valid(const pgx_seqdb *db) { return db->pack_epoch == align_epoch() && db->d_pack.p != nullptr; }
const uint32_t *seq_packs(const pgx_seqdb *db) {
  ensure_packed(db);
  return db->d_pack.p;
}

// 2-bit packs of the whole seqdb, once per overlap stage (align_epoch): [P0 | P1], lane_pack_stride dwords each
static void ensure_packed(const pgx_seqdb *db) {
  if (db->pack_epoch == align_epoch() && db->d_pack.p) return;
  hipStream_t st = ctx().stream;
  KernelTimer tm("align_pack", db->nbytes);
  const size_t nwords = (db->nbytes + 1024) / 16;   // (the seqdb buffer carries 1 KiB of zero padding)
  const size_t stride = lane_pack_stride(db);        // dwords per stream, a multiple of 4: both streams start 16-byte aligned
  if (db->d_pack.n < 2 * stride + 64) db->d_pack.alloc(2 * stride + 64);
  const size_t nr = db->rlen_by_rid.size();
  if (db->d_nflag.n < nr + 1) db->d_nflag.alloc(nr + 1);
  if (!db->d_roff_sorted.p && !db->rid.empty()) {   // idx order = position order (src/shmr_mkseqdb.c:111-112); sorted defensively
    std::vector<std::pair<uint64_t, uint32_t>> v(db->rid.size());
    for (size_t i = 0; i < v.size(); ++i) v[i] = {db->roff[i], db->rid[i]};
    std::sort(v.begin(), v.end());
    std::vector<uint64_t> ro(v.size());
    std::vector<uint32_t> ri(v.size());
    for (size_t i = 0; i < v.size(); ++i) ro[i] = v[i].first, ri[i] = v[i].second;
    db->d_roff_sorted.alloc(v.size()), db->d_rid_sorted.alloc(v.size());
    db->d_roff_sorted.upload(ro.data(), ro.size()), db->d_rid_sorted.upload(ri.data(), ri.size());
    sync();
  }
  PGX_HIP(hipMemsetAsync(db->d_nflag.p, 0, (nr + 1) * sizeof(uint32_t), st));
  PGX_HIP(hipMemsetAsync(db->d_pack.p + stride - 4, 0, 4 * sizeof(uint32_t), st));
  PGX_HIP(hipMemsetAsync(db->d_pack.p + 2 * stride - 4, 0, (64 + 4) * sizeof(uint32_t), st));
  const unsigned grid = (unsigned)std::min<size_t>((nwords + 255) / 256, (size_t)ctx().num_cu * 32);
  hipLaunchKernelGGL(k_pack2, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint4 *>(db->d_seq.p), nwords, db->d_pack.p, db->d_pack.p + stride,
                     db->d_roff_sorted.p, db->d_rid_sorted.p, (uint32_t)db->rid.size(), db->nbytes, db->d_nflag.p);
  PGX_HIP(hipGetLastError());
  db->pack_epoch = align_epoch();
}

// the lane-per-candidate form of dev_align; returns the device count / list of the candidates it handed on (esc[0], esc + 4 ..)
uint32_t *dev_align_lane(const pgx_seqdb *db, const pgx_align_key *d_keys, size_t n, int band, pgx_match *d_out) {
  hipStream_t st = ctx().stream;
  ensure_packed(db);
  const size_t stride = lane_pack_stride(db);
  LaneDesc *desc = ws<LaneDesc>("align.lane_desc", n);
  uint32_t *esc = ws<uint32_t>("align.esc", n + 8);   // [0] handed-on count, [1] the follow-up launch's work counter, [2] this launch's, [4..) list
  PGX_HIP(hipMemsetAsync(esc, 0, 4 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_lane_prep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_keys, (uint32_t)n, db->d_roff.p, db->d_rlen.p,
                     db->d_nflag.p, desc);
  const size_t lds = (size_t)LANE_DW * 64 * sizeof(uint32_t);
  const int per_cu_env = getenv("PGX_LANE_WAVES") ? atoi(getenv("PGX_LANE_WAVES")) : 0;
  const unsigned per_cu = per_cu_env > 0 ? (unsigned)per_cu_env : (unsigned)std::min<size_t>(16, (160u << 10) / lds);
  const unsigned grid = (unsigned)std::min<size_t>((n + 63) / 64, (size_t)ctx().num_cu * per_cu);
  uint32_t *why = nullptr;
  if (getenv("PGX_TRACE")) {
    why = ws<uint32_t>("align.why", 8 + 256 * 12);
    PGX_HIP(hipMemsetAsync(why, 0, (8 + 256 * 12) * sizeof(uint32_t), st));
  }
  hipLaunchKernelGGL(k_align_lane, dim3(grid), dim3(64), lds, st, db->d_pack.p, db->d_pack.p + stride, desc, (uint32_t)n, band, d_out, esc + 2,
                     esc, esc + 4, why);
  PGX_HIP(hipGetLastError());
  if (why) {
    uint32_t h[8];
    PGX_HIP(hipMemcpyAsync(h, why, sizeof(h), hipMemcpyDeviceToHost, st));
    sync();
#ifdef PGX_LANE_TRACE
    {
      std::vector<uint32_t> tr(8 + 256 * 12);
      PGX_HIP(hipMemcpyAsync(tr.data(), why, tr.size() * 4, hipMemcpyDeviceToHost, st));
      sync();
      for (uint32_t i = 0; i < std::min(tr[7], 256u); ++i) {
        const uint32_t *t = tr.data() + 8 + i * 12;
        fprintf(stderr, "[trace] d %d k %d x %d y %d m %d n %d qhi %d thi %d qa %08x ta %08x qoff %d toff %d\n", (int)t[0], (int)t[1], (int)t[2], (int)t[3], (int)t[4],
                (int)t[5], (int)t[6], (int)t[7], t[8], t[9], (int)t[10], (int)t[11]);
      }
    }
#endif
    fprintf(stderr, "[pgx] align lane: handed on -- behind a window %u, set-aside refused %u, longest ambiguous %u, end reached beside a set-aside %u, "
                    "band > ring / empty step %u, ambiguous bases %u\n", h[1], h[2], h[3], h[4], h[5], h[6]);
  }
  return esc;
}

}  // namespace pgx
