// pgx_sketch_fast.hip -- closed-form (w=80, k=16) minimizer sketch, one wavefront per read.
//
// Must reproduce mm_sketch (/root/reference/src/mm_sketch.c:70-151, is_hpc=0) exactly for reads without ambiguous
// bases.  Restatement used here (re-derived in DESIGN.md): drop the strand-ambiguous k-mers
// (fwd == rev), number the remaining k-mer "entries" 0..n-1 in position order; then for n >= w entry p is emitted iff
//      G(p): some window of w consecutive entries that contains p has minimum hash == hash(p)      (ties included)
// with the first-window correction  (m = rightmost smallest of entries 0..w-2):
//      entries p <= w-2, p != m, hash(p) == hash(m)  are always emitted;   m is emitted iff hash(w-1) > hash(m);
// and for n < w exactly one entry, the rightmost smallest, is emitted.  Output is in entry (= position) order.
//
// Mapping to the machine (per 1 KiB tile of one read):
//  phase A  lane = one aligned 16-byte load -> 2-bit forward / reverse-complement packs of its 16 bases -> LDS (F,R).
//           Then 16 steps; in step j lane l owns base 64 j + l, so its offset inside the 16-base block and both
//           v_alignbit funnel-shift amounts are per-lane constants and the LDS reads use immediate offsets.
//           canonical k-mer = min(fwd, rev); 32-bit invertible hash (12 VALU ops); entries are appended to a linear LDS
//           buffer in position order (ballot + mbcnt compaction; lane-linear when nothing is dropped).
//  phase B  lane = chunk of 16 consecutive ENTRIES (LDS chunk stride 20 dwords => conflict-free ds_read_b128):
//           B1 window minimum WM for the window ending at each entry = min(suffix-min of chunk q-5, minima of chunks
//              q-4..q-1, prefix-min of chunk q)  (chunked van Herk, 80 = 5 x 16);
//           B2 G(p) <=> max(WM over the 80 windows containing p) >= hash(p), the same trick with maxima.
//           Windows that are not full or end beyond the last entry are 0 (a hash of 0 is the minimum of every window,
//           so 0 never yields a false positive when a full window exists).
//  The buffer is indexed by entry number, so strand-ambiguous k-mers and tile boundaries need no special cases in B.
//  Interior tiles/rounds run check-free code; the first five chunks and the last tile of a read take the SPECIAL paths.
#include "pgx_internal.h"

namespace pgx {

namespace {
constexpr int K = 16;    // k-mer size of the closed-form kernel (32-bit k-mers / hash)
// window sizes handled by the closed form: w = 16 A with A in {4, 5, 6, 8} (the window is exactly A chunks of 16 entries)
constexpr int CH = 16;    // entries per chunk
constexpr int CST = 20;   // dwords per chunk in LDS (16 + 4 pad: conflict-free ds_read_b128; unpadded measured 1.5x slower)
// chunks in the LDS buffer for window w = 16 A: <= A+2 carried + 65 new + A+1 zero pad (+1 slack)
constexpr int nb_for(int A) { return 2 * A + 70; }
constexpr int TILE = 1024;
constexpr uint32_t INF = 0xFFFFFFFFu;

template <int S>
__device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t b) {  // (a << S) + b as ONE v_lshl_add_u32
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(S), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t mul_lo(uint32_t a, uint32_t k) {  // (asm: the compiler would turn the multiply back into shift-adds)
  uint32_t r;
  asm("v_mul_lo_u32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(k));
  return r;
}
__device__ __forceinline__ uint32_t mix32(uint32_t key) {  // src/mm_sketch.c:23-32 with mask = 2^32-1 (k = 16)
  key = lshl_add<21>(key, ~key);                  // ~key + (key << 21)
  key ^= key >> 24;
#ifdef PGX_HASH_SHIFTS
  key = lshl_add<8>(key, lshl_add<3>(key, key));  // key + (key<<3) + (key<<8)
  key ^= key >> 14;
  key = lshl_add<4>(key, lshl_add<2>(key, key));  // key + (key<<2) + (key<<4)
#else
  // round 3: ONE v_mul_lo_u32 each.  profiles/r03_valu_issue.txt: on gfx950 v_mul_lo_u32 issues at the same rate as v_lshl_add_u32
  // (0.207 / 0.243 wavefront-instructions per cycle and SIMD at 4 / 8 waves against 0.207 / 0.276) -- not the quarter-rate
  // instruction the round-1 comment assumed -- so the two shift-adds of either step are one instruction too many.
  key = mul_lo(key, 265u);                        // key + (key<<3) + (key<<8)
  key ^= key >> 14;
  key = mul_lo(key, 21u);                         // key + (key<<2) + (key<<4)
#endif
  key ^= key >> 28;
  key = lshl_add<31>(key, key);
  return key;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// inclusive scan over the wavefront in six v_add_u32_dpp: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8; lanes without a
// source add 0), then row 0's / row 2's total into rows 1 / 3 (row_bcast:15) and lane 31's into rows 2, 3 (row_bcast:31)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_step(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ int wave_incl_scan_dpp(int v) {
  v = dpp_add_step<0x111, 0xf>(v);
  v = dpp_add_step<0x112, 0xf>(v);
  v = dpp_add_step<0x114, 0xf>(v);
  v = dpp_add_step<0x118, 0xf>(v);
  v = dpp_add_step<0x142, 0xa>(v);
  v = dpp_add_step<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ void lds_read16(const uint32_t *p, uint32_t (&v)[16]) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint4 t = q[i];
    v[4 * i] = t.x, v[4 * i + 1] = t.y, v[4 * i + 2] = t.z, v[4 * i + 3] = t.w;
  }
}
__device__ __forceinline__ void lds_write16(uint32_t *p, const uint32_t (&v)[16]) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}
__device__ __forceinline__ int chunk_addr(int e) { return (int)__umul24((uint32_t)(e >> 4), CST) + (e & 15); }

template <int NB>
struct LdsT {
  uint32_t H[NB * CST];   // hash of entry e at (e/16 - qbase) * CST + e % 16
  uint32_t Wm[NB * CST];  // window minimum for the window ENDING at entry e
  uint16_t P[NB * CST];   // (lastPos & 0x7fff) << 1 | strand
  uint32_t C[NB];         // per chunk: min hash
  uint32_t M[NB];         // per chunk: max of Wm
  uint32_t F[66], R[66];  // 2-bit packs of the tile's 16-base blocks; [0] = last block of the previous tile
};

// Streaming mm_reduce (src/shmr_reduce.c:53-90) of one read, fused behind the sketch: level 0 consumes the emitted L0
// minimizers, level 1 consumes level 0's output.  Each level keeps the last rs-1 elements ("carry") in front of the newly
// staged ones; hash = x>>8 (32 bits at k = 16), y = low word of mm128.y (position<<1 | strand).
constexpr int RCARRY = 16, RSTAGE = 128, RBUF = RCARRY + RSTAGE;
struct RedLds {
  uint32_t h[2][RBUF];
  uint32_t y[2][RBUF];
};
struct RedState {
  int ncarry, nnew;   // elements in the buffer: [0, ncarry) carry, [ncarry, ncarry+nnew) staged
  int cnt;            // elements of this read consumed before the staged ones (offset of the first staged element)
  uint32_t lastw;     // y of the previous window's winner
};

// Process every staged element of level `lv`; winners go to `sink(hash, y)` in order.  rs <= RCARRY + 1.
template <typename Sink>
__device__ __forceinline__ void reduce_flush(RedLds &r, RedState &st, int lv, int rs, int lane, Sink sink) {
  uint32_t *H = r.h[lv], *Y = r.y[lv];
  for (int base = 0; base < st.nnew; base += 64) {
    const int tl = base + lane;             // index among the staged elements
    const int gidx = st.cnt + tl;            // offset of the element within the read's list
    const bool valid = tl < st.nnew && gidx >= rs - 1;
    uint32_t bh = 0, by = 0;
    bool slow = valid;
    if (rs == 6) {  // the default reduction factor: minimum by min3, the winner's index from an equality mask; a tie inside a
                    // window (the same k-mer twice among six consecutive minimizers) takes the general loop below
      uint32_t eq = 0;
      const int p0 = st.ncarry + tl - 5;
      if (valid) {
        const uint32_t h0 = H[p0], h1 = H[p0 + 1], h2 = H[p0 + 2], h3 = H[p0 + 3], h4 = H[p0 + 4], h5 = H[p0 + 5];
        bh = min(min(min(h0, h1), h2), min(min(h3, h4), h5));
        eq = (h0 == bh ? 32u : 0u) | (h1 == bh ? 16u : 0u) | (h2 == bh ? 8u : 0u) | (h3 == bh ? 4u : 0u) | (h4 == bh ? 2u : 0u) |
             (h5 == bh ? 1u : 0u);
      }
      slow = valid && (eq & (eq - 1)) != 0;
      if (valid && !slow) by = Y[p0 + (__builtin_clz(eq) - 26)];   // bit 5 = element 0
    }
    if (slow) {
      int p = st.ncarry + tl - (rs - 1);     // buffer position of the window's first element (>= 0 by construction)
      int sl = (gidx + 1) % rs;              // its ring slot: (gidx - rs + 1) % rs
      bh = H[p], by = Y[p];
      int bsl = sl;
      for (int j = 1; j < rs; ++j) {
        ++p;
        if (++sl == rs) sl = 0;
        const uint32_t hh = H[p];
        if (hh < bh || (hh == bh && sl < bsl)) bh = hh, by = Y[p], bsl = sl;  // ties -> lowest slot index
      }
    }
    uint32_t prevy = (uint32_t)__shfl_up((int)by, 1, 64);
    if (lane == 0) prevy = st.lastw;
    const bool emit = valid && (gidx == rs - 1 || by != prevy);
    const uint64_t vm = __ballot(valid);
    if (vm) st.lastw = (uint32_t)__builtin_amdgcn_readlane((int)by, 63 - __builtin_clzll(vm));
    const uint64_t em = __ballot(emit);
    const int idx = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u));
    sink(emit, idx, __builtin_popcountll(em), bh, by);
  }
  __syncthreads();
  // new carry = the last min(rs-1, total) elements
  const int total = st.ncarry + st.nnew;
  const int keep = total < rs - 1 ? total : rs - 1;
  uint32_t th = 0, ty = 0;
  if (lane < keep) th = H[total - keep + lane], ty = Y[total - keep + lane];
  __syncthreads();
  if (lane < keep) H[lane] = th, Y[lane] = ty;
  st.cnt += st.nnew;
  st.ncarry = keep;
  st.nnew = 0;
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// phase A step loop.  EDGE = the tile touches the first k-1 bases or the end of the read.
// ---------------------------------------------------------------------------------------------------------
template <bool EDGE, typename Lds>
__device__ __forceinline__ int phase_a_steps(Lds &s, int lane, int t, int lead, int len, int ebuf /* E - 16*qbase */) {
  const int o16 = lane & 15, g = lane >> 4;
  const int shF = 2 * (15 - o16), shR = (2 * (o16 + 1)) & 31;
  const bool last16 = o16 == 15;
  const uint32_t *pf = &s.F[g], *pr = &s.R[g];
  int run = 0;  // entries appended so far in this tile (wave uniform)
  const int ibase = t * TILE + lane - lead;
  // while every step so far appended all 64 lanes, lane l's slot advances by exactly 4 chunks per step
  int ad = chunk_addr(ebuf + lane);
  int adH = ad * 4, adP = ad * 2;  // byte offsets into H (dwords) and P (halfwords)
  uint32_t pz2 = ((uint32_t)ibase & 0x7FFFu) << 1;  // (position mod 2^15) << 1; the 16-bit store truncates the carry
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t F0 = pf[4 * j], F1 = pf[4 * j + 1];
    const uint32_t R0 = pr[4 * j], R1 = pr[4 * j + 1];
    const uint32_t fw = __builtin_amdgcn_alignbit(F0, F1, shF);
    const uint32_t ra = __builtin_amdgcn_alignbit(R1, R0, shR);
    const uint32_t rv = last16 ? R1 : ra;
    bool valid = fw != rv;  // strand-ambiguous k-mers are not entries
    if (EDGE) {
      const int i = ibase + 64 * j;
      valid = valid && i >= K - 1 && i < len;
    }
    const uint64_t vm = __ballot(valid);
    const uint32_t hh = mix32(min(fw, rv));
    const uint32_t pz = pz2 + (fw > rv ? 1u : 0u);
    pz2 += 128;
    if (vm == ~0ull) {  // nothing dropped: lane-linear address
      *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s.H) + adH) = hh;
      *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(s.P) + adP) = (uint16_t)pz;
      run += 64;
      adH += 4 * CST * 4, adP += 4 * CST * 2;
    } else {
      if (valid) {
        const int idx = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0u));
        const int ax = chunk_addr(ebuf + run + idx);
        s.H[ax] = hh;
        s.P[ax] = (uint16_t)pz;
      }
      run += __builtin_popcountll(vm);
      ad = chunk_addr(ebuf + run + lane);
      adH = ad * 4, adP = ad * 2;
    }
  }
  return run;
}

// ---------------------------------------------------------------------------------------------------------
// phase B1: window minima of absolute chunk q (buffer chunk bq).  SPECIAL handles q < 5 and the read's end.
// ---------------------------------------------------------------------------------------------------------
template <bool SPECIAL, int A, typename Lds>
__device__ __forceinline__ void phase_b1(Lds &s, int lane, int q, int bq, bool active, int E) {
  constexpr int W = 16 * A;
  uint32_t v[16], wm[16];
  uint32_t c = INF;
  if (active) {
    lds_read16(&s.H[bq * CST], v);
    c = v[0];
#pragma unroll
    for (int o = 1; o < 16; ++o) c = min(c, v[o]);
  }
  // minima of the four preceding chunks: neighbours' registers, or LDS for chunks older than this round
  uint32_t m4 = INF;
#pragma unroll
  for (int d = 1; d <= A - 1; ++d) {
    uint32_t cd = (uint32_t)__shfl_up((int)c, d, 64);
    if (lane < d) cd = (bq - d >= 0) ? s.C[bq - d] : INF;
    if (SPECIAL && q - d < 0) cd = INF;
    m4 = min(m4, cd);
  }
  if (active) {
    uint32_t sfx[17];
    sfx[16] = INF;
    if (!SPECIAL || q >= A) {
      uint32_t u[16];
      lds_read16(&s.H[(bq - A) * CST], u);
      sfx[15] = u[15];
#pragma unroll
      for (int o = 14; o >= 1; --o) sfx[o] = min(sfx[o + 1], u[o]);
    } else {
#pragma unroll
      for (int o = 1; o < 16; ++o) sfx[o] = INF;
    }
    uint32_t p = INF, mx = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      p = min(p, v[o]);
      uint32_t x = min(min(sfx[o + 1], m4), p);
      if (SPECIAL) {
        const int jj = q * CH + o;
        if (jj < W - 1 || jj >= E) x = 0;  // not a full window / beyond the last entry
      }
      wm[o] = x;
      mx = max(mx, x);
    }
    lds_write16(&s.Wm[bq * CST], wm);
    s.C[bq] = c;
    s.M[bq] = mx;
  }
}

}  // namespace

// FUSED = false: the read's L0 minimizers go to its slab.  FUSED = true: they are reduced `levels` times on the fly
// (reduce_flush) and only the final level reaches the slab -- L0 never leaves the CU.
// PACKED (round 6): the read comes from its 2-bit pack (pgx_pack.hip: forward strand from dword poff[rid] on, base i in bits 2 (i % 16) .. + 1 of
// dword i / 16) instead of the seqdb's bytes: a lane's block is ONE dword, F its 16 fields in reverse order, R its complement -- 6 lane-ops
// instead of ~40 for the nibble decode, a quarter of the bytes, and the read starts at its first block (lead = 0).  A read with an ambiguous base
// has no pack (nflag): it is flagged like one whose bytes fail the one-hot test.
template <bool FUSED, int A, bool PACKED = false>
__global__ __launch_bounds__(64) void k_sketch_wave(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                    const uint32_t *__restrict__ list, uint32_t n_list,
                                                    pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                                                    uint32_t *__restrict__ counts, uint32_t *__restrict__ flags, int rs,
                                                    int levels, uint32_t *__restrict__ need, int off_by_list,
                                                    const uint64_t *__restrict__ poff = nullptr, const uint32_t *__restrict__ nflag = nullptr) {
  // need (optional): the number of elements the read produces, written even when its slab was too small (a second launch
  // with exact slabs then redoes exactly those reads); off_by_list: slab_off is indexed by the position in `list`, not by slot
  constexpr int W = 16 * A;  // window size in entries = A chunks of 16
  using Lds = LdsT<nb_for(A)>;
  __shared__ __attribute__((aligned(16))) Lds s;
  __shared__ RedLds red;
  RedState rst[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  const int lane = threadIdx.x;
  if (blockIdx.x >= n_list) return;
  const uint32_t slot = list ? list[blockIdx.x] : blockIdx.x;
  const ReadDesc rd = reads[slot];
  const int len = (int)rd.len;
  const int lead = PACKED ? 0 : (int)(rd.off & 15);
  const uint8_t *base = PACKED ? seq + poff[rd.rid] * 4 : seq + (rd.off - (uint64_t)lead);
  const int span = lead + len;
  const int ntiles = (span + TILE - 1) / TILE;
  const uint32_t oi = off_by_list ? blockIdx.x : slot;
  pgx_mm128 *out = slab + slab_off[oi];
  const uint32_t cap = (uint32_t)(slab_off[oi + 1] - slab_off[oi]);

  int E = 0;       // entries produced so far
  int qbase = 0;   // absolute chunk number of buffer chunk 0
  int wdone = 0;   // chunks whose window minima are in Wm
  int ddone = 0;   // chunks already decided
  uint32_t nout = 0;
  uint32_t bad = 0;
  uint32_t Fkeep = 0, Rkeep = 0;  // this lane's packs of the previous tile (lane 63's become block -1)
  uint32_t fw_mv = 0, fw_e79 = 0;  // first window (mm_sketch.c:116-128): minimum of entries 0 .. W-2, entry W-1,
  int fw_m = -1;                   // and the minimum's rightmost occurrence -- taken when chunk 0 is decided

  // fused mode: push the staged L0 minimizers through level 0 (and level 1); final-level elements go to the slab
  auto fused_flush = [&]() {
    const uint64_t yhi = (uint64_t)rd.rid << 32;
    auto to_slab = [&](bool emit, int idx, int tot, uint32_t hh, uint32_t yy) {
      if (nout + (uint32_t)tot <= cap) {
        if (emit) out[nout + (uint32_t)idx] = pgx_mm128{((uint64_t)hh << 8) | (uint64_t)K, yhi | yy};
      } else {
        bad |= 1;
      }
      nout += (uint32_t)tot;
    };
    if (levels == 1) {
      reduce_flush(red, rst[0], 0, rs, lane, to_slab);
    } else {
      auto to_l1 = [&](bool emit, int idx, int tot, uint32_t hh, uint32_t yy) {
        const int w = rst[1].ncarry + rst[1].nnew + idx;  // level-0 flushes stage at most RSTAGE elements: fits
        if (emit) red.h[1][w] = hh, red.y[1][w] = yy;
        rst[1].nnew += tot;
      };
      reduce_flush(red, rst[0], 0, rs, lane, to_l1);
      reduce_flush(red, rst[1], 1, rs, lane, to_slab);
    }
  };
  uint4 raw_next = make_uint4(0, 0, 0, 0);
  if (PACKED) {
    if (nflag[rd.rid] & 1u) bad |= 1;   // (an ambiguous base somewhere in the read: the run-by-run path takes it)
    if (lane * 16 < span) raw_next.x = reinterpret_cast<const uint32_t *>(base)[lane];
  } else if (lane * 16 < span) {
    raw_next = *reinterpret_cast<const uint4 *>(base + lane * 16);
  }
  for (int t = 0; t < ntiles; ++t) {
    // ---- compaction: move the live chunks [ddone, ceil(E/16)) to the front of the buffer -----------------------
    if (ddone > qbase) {
      const int shift = (ddone - qbase) * CST;
      const int live = ((E + CH - 1) / CH - ddone) * CST;  // dwords to keep (<= A + 2 <= 10 chunks = 200 dwords)
      uint32_t th[4], tw[4];
      uint16_t tp[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = lane + 64 * r;
        th[r] = tw[r] = 0, tp[r] = 0;
        if (d < live) th[r] = s.H[shift + d], tw[r] = s.Wm[shift + d], tp[r] = s.P[shift + d];
      }
      uint32_t tc = 0, tm = 0;
      if (lane * CST < live) tc = s.C[ddone - qbase + lane], tm = s.M[ddone - qbase + lane];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = lane + 64 * r;
        if (d < live) s.H[d] = th[r], s.Wm[d] = tw[r], s.P[d] = tp[r];
      }
      if (lane * CST < live) s.C[lane] = tc, s.M[lane] = tm;
      qbase = ddone;
    }
    // ---- phase A: load, decode, pack ------------------------------------------------------------------------
    const int b0 = t * TILE + lane * 16;  // byte offset from `base` of this lane's 16-base block
    const int i0 = b0 - lead;             // read position of the block's first base
    const uint4 raw = raw_next;           // loaded one tile ahead: the HBM latency hides behind the previous tile
    raw_next = make_uint4(0, 0, 0, 0);
    if (PACKED) {
      if (b0 + TILE < span) raw_next.x = reinterpret_cast<const uint32_t *>(base)[(b0 + TILE) >> 4];
    } else if (b0 + TILE < span) {
      raw_next = *reinterpret_cast<const uint4 *>(base + b0 + TILE);
    }
    const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t F = 0;
    const bool inside = i0 >= 0 && i0 + 16 <= len;
    if (PACKED) {   // the block's 16 fields in reverse order: earlier bases at higher bits
      const uint32_t br = __builtin_bitreverse32(raw.x);
      F = ((br & 0x55555555u) << 1) | ((br >> 1) & 0x55555555u);
    }
#pragma unroll
    for (int d = 0; d < 4 && !PACKED; ++d) {
      const uint32_t n = dw[d] & 0x0F0F0F0Fu;  // forward-strand one-hot nibbles (src/shmr_utils.c:18-30)
      const uint32_t c = ((n >> 1) & 0x07070707u) - ((n >> 3) & 0x01010101u);  // {1,2,4,8} -> {0,1,2,3} bytewise
      const uint32_t tt = n - 0x01010101u;
      uint32_t bd = (tt & ~n & 0x80808080u) | (n & tt);  // a nibble that is zero or has two bits set
      if (!inside) {  // partial block at a read end: test byte by byte
        bd = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = i0 + 4 * d + b;
          const uint32_t nb = (n >> (8 * b)) & 0xF;
          if (i >= 0 && i < len && __builtin_popcount(nb) != 1) bd = 1;
        }
      }
      bad |= bd;
      const uint32_t f8 = ((c << 6) | (c >> 4) | (c >> 14) | (c >> 24)) & 0xFFu;  // earlier bases at higher bits
      F = (F << 8) | f8;
    }
    const uint32_t rr = __builtin_bitreverse32(~F);
    const uint32_t R = ((rr & 0x55555555u) << 1) | ((rr >> 1) & 0x55555555u);  // complement, later bases higher
    if (lane == 63) s.F[0] = Fkeep, s.R[0] = Rkeep;
    s.F[1 + lane] = F;
    s.R[1 + lane] = R;
    Fkeep = F, Rkeep = R;
    __syncthreads();
    const bool last = (t == ntiles - 1);
    const bool edge = (t * TILE - lead < K - 1) || ((t + 1) * TILE - lead > len);
    const int ebuf = E - qbase * CH;
    E += edge ? phase_a_steps<true, Lds>(s, lane, t, lead, len, ebuf) : phase_a_steps<false, Lds>(s, lane, t, lead, len, ebuf);
    const int hch = last ? (E + CH - 1) / CH : E / CH;  // chunks whose hashes are final
    if (last) {                                          // pad the tail of the last chunk
      const int e = E + lane;
      if (lane < CH && e < hch * CH) s.H[(e / CH - qbase) * CST + (e & 15)] = INF;
    }
    __syncthreads();

    // ---- phase B1 -------------------------------------------------------------------------------------------
    for (int q0 = wdone; q0 < hch; q0 += 64) {
      const int q = q0 + lane;
      if (q0 < A || last) phase_b1<true, A, Lds>(s, lane, q, q - qbase, q < hch, E);
      else phase_b1<false, A, Lds>(s, lane, q, q - qbase, q < hch, E);
      __syncthreads();
    }
    wdone = hch;
    if (last) {  // windows past the end do not exist: A+1 all-zero chunks
      const int bz = hch - qbase;
#pragma unroll
      for (int z = 0; z < A + 1; z += 4)
        if (z + (lane >> 4) < A + 1) s.Wm[(bz + z + (lane >> 4)) * CST + (lane & 15)] = 0;
      if (lane < A + 1) s.M[bz + lane] = 0;
      __syncthreads();
    }

    // ---- phase B2: decide and emit ----------------------------------------------------------------------------
    const int dlimit = last ? hch : (wdone - A > 0 ? wdone - A : 0);
    const bool short_read = last && E < W;  // fewer than w entries: emit only the rightmost smallest
    for (int q0 = ddone; q0 < dlimit; q0 += 64) {
      const int q = q0 + lane, bq = q - qbase;
      uint32_t emask = 0;
      uint32_t v[16];
      if (q < dlimit) {
        uint32_t wq[16], wn[16];
        lds_read16(&s.H[bq * CST], v);
        lds_read16(&s.Wm[bq * CST], wq);
        lds_read16(&s.Wm[(bq + A) * CST], wn);
        uint32_t m4 = 0;
#pragma unroll
        for (int d = 1; d <= A - 1; ++d) m4 = max(m4, s.M[bq + d]);
        uint32_t sm[16];
        sm[15] = wq[15];
#pragma unroll
        for (int o = 14; o >= 0; --o) sm[o] = max(sm[o + 1], wq[o]);
        uint32_t pm = 0, rmask = 0;  // rmask collects the decisions MSB-first (bit 15-o ... reversed below)
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const uint32_t f = max(max(sm[o], m4), pm);  // pm = prefix max of chunk q+5 up to o-1
          pm = max(pm, wn[o]);
          // rmask = (rmask << 1) | (f >= v[o])   as v_cmp + v_addc
          asm("v_cmp_ge_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rmask) : "v"(f), "v"(v[o]) : "vcc");
        }
        emask = __builtin_bitreverse32(rmask) >> 16;
        if (last) {  // entries past the end are padding
          const int nvalid = E - q * CH;
          if (nvalid < 16) emask &= (1u << (nvalid < 0 ? 0 : nvalid)) - 1u;
        }
        if (short_read) emask = 0;
      }
      if (q0 == 0) {  // first-window correction / short-read rule; entries 0..79 are buffer chunks 0..4 (qbase == 0: nothing was decided yet)
        const int lim = short_read ? E : W - 1;  // candidates are entries [0, lim)
        const uint32_t a = lane < lim ? s.H[(lane >> 4) * CST + (lane & 15)] : INF;
        const uint32_t b = lane + 64 < lim ? s.H[(4 + (lane >> 4)) * CST + (lane & 15)] : INF;  // W <= 128 = 2 x 64 lanes
        const uint32_t mv = wave_min_u32(min(a, b));
        const uint64_t mb = __ballot(lane + 64 < lim && b == mv), ma = __ballot(lane < lim && a == mv);
        const int m = mb ? 64 + (63 - __builtin_clzll(mb)) : (ma ? 63 - __builtin_clzll(ma) : -1);
        fw_mv = mv, fw_m = m;
        if (short_read) {
          if (m >= 0 && q == m / CH) emask = 1u << (m % CH);
        } else {
          fw_e79 = s.H[((W - 1) >> 4) * CST + ((W - 1) & 15)];
        }
      }
      // The correction applies to entries 0 .. W-2, i.e. chunks 0 .. A-1 -- in WHICHEVER pass decides them.  A read whose entries
      // are sparse (long runs of strand-ambiguous k-mers: a (GC)n array at the read start, c4s read 279,270) gets fewer than 2 A
      // chunks out of its first tile, chunk 0 is decided alone and chunks 1 .. A-1 in a later pass, after the compaction has dropped
      // chunk 0: hence the minimum, its rightmost occurrence and entry W-1 are kept from the pass that decides chunk 0.  (Rounds 1-3
      // applied the correction only in that pass and emitted the rightmost tie although entry W-1 equalled it: mm_sketch.c:126-128.)
      if (!short_read && q < A && q < dlimit) {
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const int pidx = q * CH + o;
          if (pidx <= W - 2 && v[o] == fw_mv) {
            if (pidx != fw_m) emask |= 1u << o;
            else if (fw_e79 > fw_mv) emask |= 1u << o;
            else emask &= ~(1u << o);
          }
        }
      }
      const int ec = __builtin_popcount(emask);
      if (__ballot(ec != 0)) {
        const int einc = wave_incl_scan(ec, lane);
        const int etot = __shfl(einc, 63, 64);
        // positions were stored modulo 2^15; the newest base seen so far bounds them from above
        const int imax = (t + 1) * TILE - lead - 1;
        if (!FUSED) {
          uint32_t w = nout + (uint32_t)(einc - ec);
          if (nout + (uint32_t)etot <= cap) {
            uint32_t em = emask;
            while (em) {
              const int o = __builtin_ctz(em);
              em &= em - 1;
              const uint32_t pz = s.P[bq * CST + o];
              const int i = imax - ((imax - (int)(pz >> 1)) & 0x7FFF);
              pgx_mm128 e;
              e.x = ((uint64_t)s.H[bq * CST + o] << 8) | (uint64_t)K;
              e.y = ((uint64_t)rd.rid << 32) | ((uint64_t)(uint32_t)i << 1) | (uint64_t)(pz & 1u);
              out[w++] = e;
            }
          } else {
            bad |= 1;  // slab overflow: redone with a larger slab (exact-slab pass of the index stage / the run-by-run path)
          }
          nout += (uint32_t)etot;
        } else {
          // the step's entries go to the level-0 staging area in position order; a burst of ties (a homopolymer or a short-period
          // tandem array makes EVERY position a tied window minimum: up to 1,024 entries in a step) is staged and flushed in
          // pieces of at most RSTAGE
          for (int done = 0; done < etot;) {  // (wave-uniform)
            if (rst[0].nnew + min(etot - done, RSTAGE) > RSTAGE) fused_flush();
            const int take = min(RSTAGE - rst[0].nnew, etot - done);
            const int w0 = rst[0].ncarry + rst[0].nnew - done;
            int g = einc - ec;  // ordinal of this lane's first entry among the step's
            uint32_t em = emask;
            while (em) {
              const int o = __builtin_ctz(em);
              em &= em - 1;
              if (g >= done && g < done + take) {
                const uint32_t pz = s.P[bq * CST + o];
                const int i = imax - ((imax - (int)(pz >> 1)) & 0x7FFF);
                red.h[0][w0 + g] = s.H[bq * CST + o];
                red.y[0][w0 + g] = ((uint32_t)i << 1) | (pz & 1u);
              }
              ++g;
            }
            rst[0].nnew += take;
            done += take;
            __syncthreads();
            if (rst[0].nnew >= 96) fused_flush();  // a flush costs ~300 wave instructions: amortise it over ~4 tiles
          }
        }
      }
    }
    ddone = dlimit;
    __syncthreads();
  }
  if (FUSED) fused_flush();
  const uint64_t anybad = __ballot(bad != 0);
  if (lane == 0) {
    counts[slot] = anybad ? 0u : nout;
    if (anybad) flags[slot] = 1;
    if (need) need[slot] = nout;
  }
}

// =========================================================================================================
// k_sketch_blk -- the closed form again, in POSITION space, one lane per 16-base block (round 2).
//
// k_sketch_wave above gives a lane one BASE per step in phase A (k-mers through LDS, entries compacted into an
// entry-indexed LDS buffer, a second layout for phase B).  Here lane l of tile t owns the 16 bases of one aligned
// 16-byte load -- global block G = 64 t + l -- for BOTH phases: its 16 k-mers come from two funnel shifts each of its own
// and its left neighbour's 2-bit packs (compile-time shift amounts), its 16 hashes stay in registers, and the window
// minima / maxima of the closed form are the chunked van Herk scheme over those registers:
//   WM(i)  = min h over the w entries ending at position i  = min3(suffix of block G-A from offset o+1, minima of blocks
//            G-A+1..G-1, prefix of block G up to o)                                        (w = 16 A, A = 5)
//   G(p)  <=> max over the windows containing p of WM >= h(p); the windows containing p = (G-A, o) END in block G-A
//            from o on, in blocks G-A+1..G-1, and in block G before o: lane G decides block G-A, so every cross-lane
//            dependence points BACKWARDS (to lower lanes or the previous tile) and a tile never waits for the next one.
// Cross-lane data moves through one 5 KiB LDS exchange buffer (16 hashes, later 16 suffix maxima per block, rows padded to
// 20 dwords for conflict-free 128-bit accesses) plus per-block minima / maxima; the last A blocks of a tile are carried.
// Strand-ambiguous k-mers ("drops", fw == rv: 4^-8 per position, one read in five has one) are not entries: a window that
// spans a drop at position x reaches one position further back, i.e. its suffix / prefix index moves by one -- handled in
// MODE 2 of the tile body; MODE 1 is the read's first and last tiles (positions outside [k-1, len), windows that do not
// exist, the first-window correction); MODE 0 the interior.  Anything else -- two drops within ~1.2 k bases, a drop in the
// first 200 positions, an ambiguous base, reads shorter than a window + 200, a burst of ties overflowing the staging -- sets
// the read's flag and k_sketch_wave (the validated general form) redoes that read.
// The emitted L0 minimizers of a tile are compacted (positions only), their hash and strand recomputed from the pack ring by
// one lane each, and pushed through the streaming two-level mm_reduce (reduce_flush above): L0 never leaves the CU.
// =========================================================================================================
namespace {
constexpr int BST = 20;      // dwords per block row in the exchange buffer
constexpr int QCAP = 192;    // emitted positions per tile the queue holds (a tile emits ~25)
constexpr int XNONE = -(1 << 29);   // "no drop"

template <int A>
struct BlkLds {
  uint32_t X[64 * BST];       // phase 1: the 16 hashes of every block of the tile; phase 2: suffix maxima of its window minima
  uint32_t Vc[2][A * BST];    // the last A blocks' hashes of the previous tile (by tile parity)
  uint32_t Sc[2][A * BST];    // ... and their suffix maxima
  uint32_t C[A + 64];         // block minima: [0, A) carried from the previous tile, A + lane = this tile
  uint32_t M[A + 64];         // block maxima of the window minima, same layout
  uint32_t Fr[128], Rr[128];  // 2-bit packs of the last two tiles' blocks (index = global block & 127): hash / strand of emitted entries
  uint32_t Q[QCAP];           // positions of the tile's emitted entries, in order
  uint32_t fw_mv, fw_e;       // first-window correction: minimum of entries 0..W-2, entry W-1
  int fw_m;                   // position of the rightmost smallest of entries 0..W-2
};

// per-read state the tile body shares with the kernel
struct BlkState {
  int len, lead;
  int x_drop;                   // position of the most recent strand-ambiguous k-mer (XNONE: none)
  uint32_t bad;                 // why the read goes to the general kernel (bits: see the end of the kernel)
};

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, int sh, uint32_t b) { return (a << sh) | b; }

// 16 one-hot bytes -> 2-bit pack of the COMPLEMENT codes (earlier bases at higher bits); the caller inverts it.  v_perm_b32 is
// the nibble -> code table: A(1)->3, C(2)->2, G(4)->1, T(8)-> selector 8 = sign of table byte 1 (0x03: clear) = 0x00; a zero nibble
// (and 3, 5, 6, 7, 13..15) reads a byte with the flag bit 0x04.  With no flagged nibble the block is clean iff its nibbles hold
// exactly 16 set bits (the remaining two-bit nibbles 9, 10, 12 raise the count).
__device__ __forceinline__ uint32_t decode16c(const uint4 raw, uint32_t &flagacc, uint32_t &popc) {
  const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
  uint32_t Fc = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const uint32_t n = dw[d] & 0x0F0F0F0Fu;
    const uint32_t pm = __builtin_amdgcn_perm(0x04040401u, 0x04020304u, n);
    flagacc |= pm;
    popc += (uint32_t)__builtin_popcount(n);
    const uint32_t c = pm & 0x03030303u;
    const uint32_t t1 = lshl_or(c, 10, c);
    const uint32_t t2 = lshl_or(t1, 20, t1);      // top byte = c0<<6 | c1<<4 | c2<<2 | c3
    Fc = __builtin_amdgcn_alignbit(Fc, t2, 24);   // (Fc << 8) | (t2 >> 24)
  }
  return Fc;
}
__device__ __forceinline__ uint32_t revcomp_of_comp(uint32_t Fcomp) {   // Fcomp = ~F
  const uint32_t rr = __builtin_bitreverse32(Fcomp);
  return ((rr & 0x55555555u) << 1) | ((rr >> 1) & 0x55555555u);  // complement, later bases at higher bits
}

// One tile: B1 (window minima of the lane's block), B2 (decisions for block G - A); returns the 16-bit emission mask of block
// G - A.  MODE 0: interior tile, every position an entry.  MODE 1: the tile touches a read end (positions outside [K-1, len),
// windows that do not exist, the first-window correction) but no drop is near.  MODE 2: a drop in the tile or within a window
// before it (and anything MODE 1 handles).
template <int A, int MODE>
__device__ __forceinline__ uint32_t blk_tile(BlkLds<A> &s, BlkState &st, const int lane, const int t, const uint32_t F,
                                             const uint32_t Fp, const uint32_t R, const uint32_t Rp, uint32_t (&v)[16]) {
  constexpr int W = 16 * A;
  constexpr bool SPECIAL = MODE != 0, DROP = MODE == 2;
  const int par = t & 1;
  const int ibase = t * TILE + lane * 16 - st.lead;
  const int row_own = lane * BST;
  const bool from_carry = lane < A;
  const int row_src = from_carry ? lane * BST : (lane - A) * BST;
  uint32_t vmask = 0xFFFFu;  // which of the block's positions are entries
  if (MODE == 1) {
    vmask = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const int i = ibase + o;
      if ((uint32_t)(i - (K - 1)) < (uint32_t)(st.len - (K - 1))) vmask |= 1u << o;
      else v[o] = INF;
    }
  }
  if (DROP) {
    uint32_t dbits = 0;
    vmask = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const uint32_t fw = o == 15 ? F : __builtin_amdgcn_alignbit(Fp, F, 2 * (15 - o));
      const uint32_t rv = o == 15 ? R : __builtin_amdgcn_alignbit(R, Rp, 2 * (o + 1));
      const int i = ibase + o;
      const bool inpos = i >= K - 1 && i < st.len;
      if (inpos && fw == rv) dbits |= 1u << o;
      if (inpos && fw != rv) vmask |= 1u << o;
      else v[o] = INF;
    }
    const uint64_t dm = __ballot(dbits != 0);
    if (dm) {
      // one drop at a time: the tile body knows a single x.  (A second one close by, or one inside the first window, is the
      // general kernel's business.)
      const int dl = __builtin_ctzll(dm);
      const uint32_t db = (uint32_t)__builtin_amdgcn_readlane((int)dbits, dl);
      const int xnew = (t * 64 + dl) * 16 + __builtin_ctz(db) - st.lead;
      if (__builtin_popcountll(dm) > 1 || __builtin_popcount(db) > 1) st.bad |= 4;
      if (xnew < 200) st.bad |= 8;
      if (st.x_drop != XNONE && xnew - st.x_drop < TILE + 2 * W) st.bad |= 16;
      st.x_drop = xnew;
    }
  }
  const int x = st.x_drop;

  // ---- B1: window minima of the block's 16 window ends -----------------------------------------------------------------
  uint32_t pre[16];   // prefix minima of the block's own hashes (B1 needs them anyway): the last one is the block minimum
  pre[0] = v[0];
#pragma unroll
  for (int o = 1; o < 16; ++o) pre[o] = min(pre[o - 1], v[o]);
  const uint32_t c = pre[15];
  lds_write16(&s.X[row_own], v);
  if (lane >= 64 - A) lds_write16(&s.Vc[par][(lane - (64 - A)) * BST], v);
  s.C[A + lane] = c;
  __syncthreads();
  uint32_t u[16];
  lds_read16(from_carry ? &s.Vc[par ^ 1][row_src] : &s.X[row_src], u);
  uint32_t m4 = INF;
#pragma unroll
  for (int d = 1; d <= A - 1; ++d) m4 = min(m4, s.C[A + lane - d]);
  uint32_t wm[16];
  {
    uint32_t S[17];
    S[16] = INF;
#pragma unroll
    for (int o = 15; o >= 0; --o) S[o] = min(S[o + 1], u[o]);
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const uint32_t p = pre[o];
      if (!SPECIAL) {
        wm[o] = min3u(S[o + 1], m4, p);
      } else {
        const int i = ibase + o;
        const bool dsh = DROP && (uint32_t)(i - x - 1) < (uint32_t)(W - 1);   // the window spans the drop: one position further back
        const uint32_t w = min3u(dsh ? S[o] : S[o + 1], m4, p);
        const bool exists = ((vmask >> o) & 1u) && i >= W + K - 2;    // a full window of entries ends here
        wm[o] = exists ? w : 0u;
      }
    }
  }
  if (SPECIAL && t == 0) {  // first-window correction (mm_sketch.c:116-125): entries 0..W-2 are positions K-1..W+K-3 (no drop there)
    const int lim = W - 1;
    const int ba = st.lead + K - 1 + lane, bb = ba + 64;
    const uint32_t a = lane < lim ? s.X[(ba >> 4) * BST + (ba & 15)] : INF;
    const uint32_t b = lane + 64 < lim ? s.X[(bb >> 4) * BST + (bb & 15)] : INF;
    const uint32_t mv = wave_min_u32(min(a, b));
    const uint64_t mb = __ballot(lane + 64 < lim && b == mv), ma = __ballot(lane < lim && a == mv);
    const int be = st.lead + K - 1 + W - 1;
    if (lane == 0) {
      s.fw_mv = mv;
      s.fw_m = K - 1 + (mb ? 64 + (63 - __builtin_clzll(mb)) : (63 - __builtin_clzll(ma)));
      s.fw_e = s.X[(be >> 4) * BST + (be & 15)];
    }
  }

  // ---- B2: decide block G - A --------------------------------------------------------------------------------------------
  {
    uint32_t sm[16];
    sm[15] = wm[15];
#pragma unroll
    for (int o = 14; o >= 0; --o) sm[o] = max(sm[o + 1], wm[o]);
    __syncthreads();  // (every lane has read its source row of hashes)
    lds_write16(&s.X[row_own], sm);
    if (lane >= 64 - A) lds_write16(&s.Sc[par][(lane - (64 - A)) * BST], sm);
  }
#pragma unroll
  for (int o = 1; o < 16; ++o) wm[o] = max(wm[o - 1], wm[o]);   // prefix maxima of the block's own window minima
  const uint32_t Mx = wm[15];                                   // = the block's maximum
  s.M[A + lane] = Mx;
  __syncthreads();
  uint32_t emask = 0;
  {
    uint32_t sq[16];
    lds_read16(from_carry ? &s.Sc[par ^ 1][row_src] : &s.X[row_src], sq);
    uint32_t m4x = 0;
#pragma unroll
    for (int d = 1; d <= A - 1; ++d) m4x = max(m4x, s.M[A + lane - d]);
    uint32_t rmask = 0;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      uint32_t pmx = o ? wm[o - 1] : 0u;
      if (DROP) {
        const int ip = ibase - W + o;
        if ((uint32_t)(x - ip - 1) < (uint32_t)(W - 1)) pmx = wm[o];   // a drop inside: the last window ends one position later
      }
      const uint32_t f = max3u(sq[o], m4x, pmx);
      asm("v_cmp_ge_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(rmask) : "v"(f), "v"(u[o]) : "vcc");
    }
    emask = __builtin_bitreverse32(rmask) >> 16;
    if (SPECIAL && t == 0) {  // ties of the first window's minimum (positions K-1 .. W+K-3)
      const uint32_t fw_mv = s.fw_mv, fw_e = s.fw_e;
      const int fw_m = s.fw_m;
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        const int ip = ibase - W + o;
        if (ip >= K - 1 && ip <= W + K - 3 && u[o] == fw_mv) {
          if (ip != fw_m || fw_e > fw_mv) emask |= 1u << o;
          else emask &= ~(1u << o);
        }
      }
    }
  }
  // carries for the next tile; the exchange buffer is free again after the barrier
  __syncthreads();
  if (lane >= 64 - A) s.C[lane - (64 - A)] = c, s.M[lane - (64 - A)] = Mx;
  return emask;
}
}  // namespace

template <int A, bool PACKED = false>   // PACKED: from the read's 2-bit pack, as in k_sketch_wave above
__global__ __launch_bounds__(64, 4) void k_sketch_blk(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                      uint32_t n_reads, pgx_mm128 *__restrict__ slab,
                                                      const uint64_t *__restrict__ slab_off, uint32_t *__restrict__ counts,
                                                      uint32_t *__restrict__ flags, int rs, int levels, int dbg,
                                                      const uint64_t *__restrict__ poff = nullptr, const uint32_t *__restrict__ nflag = nullptr) {
  constexpr int W = 16 * A;          // window size in entries
  __shared__ __attribute__((aligned(16))) BlkLds<A> s;
  __shared__ RedLds red;
  RedState rst[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  const int lane = threadIdx.x;
  const uint32_t slot = blockIdx.x;
  if (slot >= n_reads) return;
  const ReadDesc rd = reads[slot];
  const int len = (int)rd.len;
  const int lead = PACKED ? 0 : (int)(rd.off & 15);
  const uint8_t *base = PACKED ? seq + poff[rd.rid] * 4 : seq + (rd.off - (uint64_t)lead);
  const int span = lead + len;
  if (len < W + K - 1 + 200) {  // fewer than a window of entries, or too short for the drop rules: the general kernel
    if (lane == 0) counts[slot] = 0, flags[slot] = 1;   // (flag bits: see the end of the kernel)
    return;
  }
  if (PACKED && (nflag[rd.rid] & 1u)) {   // an ambiguous base somewhere in the read (no 2-bit code): the run-by-run path
    if (lane == 0) counts[slot] = 0, flags[slot] = 2;
    return;
  }
  const int ntiles = (((span - 1) >> 4) + A) / 64 + 1;  // the block of the last base is decided by global lane + A
  pgx_mm128 *out = slab + slab_off[slot];
  const uint32_t cap = (uint32_t)(slab_off[slot + 1] - slab_off[slot]);
  uint32_t nout = 0;
  BlkState st{len, lead, XNONE, 0u};

  auto fused_flush = [&]() {
    const uint64_t yhi = (uint64_t)rd.rid << 32;
    auto to_slab = [&](bool emit, int idx, int tot, uint32_t hh, uint32_t yy) {
      if (nout + (uint32_t)tot <= cap) {
        if (emit) out[nout + (uint32_t)idx] = pgx_mm128{((uint64_t)hh << 8) | (uint64_t)K, yhi | yy};
      } else {
        st.bad |= 128;
      }
      nout += (uint32_t)tot;
    };
    if (levels == 1) {
      reduce_flush(red, rst[0], 0, rs, lane, to_slab);
    } else {
      auto to_l1 = [&](bool emit, int idx, int tot, uint32_t hh, uint32_t yy) {
        const int w = rst[1].ncarry + rst[1].nnew + idx;
        if (emit) red.h[1][w] = hh, red.y[1][w] = yy;
        rst[1].nnew += tot;
      };
      reduce_flush(red, rst[0], 0, rs, lane, to_l1);
      reduce_flush(red, rst[1], 1, rs, lane, to_slab);
    }
  };

  // carries of "tile -1": no entries
  for (int i = lane; i < A * BST; i += 64) s.Vc[1][i] = INF, s.Sc[1][i] = 0;
  if (lane < A) s.C[lane] = INF, s.M[lane] = 0;
  uint32_t Fc = 0, Rc = 0;        // packs of the block left of the tile (wave uniform)

  uint4 raw_next = make_uint4(0, 0, 0, 0);
  if (PACKED) {
    if (lane * 16 < span) raw_next.x = reinterpret_cast<const uint32_t *>(base)[lane];
  } else if (lane * 16 < span) {
    raw_next = *reinterpret_cast<const uint4 *>(base + lane * 16);
  }
  for (int t = 0; t < ntiles; ++t) {
    const int G = t * 64 + lane;
    const int b0 = t * TILE + lane * 16;   // byte offset of the block from `base`
    const int ibase = b0 - lead;           // read position of its first base
    const uint4 raw = raw_next;
    raw_next = make_uint4(0, 0, 0, 0);
    if (PACKED) {
      if (b0 + TILE < span) raw_next.x = reinterpret_cast<const uint32_t *>(base)[(b0 + TILE) >> 4];
    } else if (b0 + TILE < span) {
      raw_next = *reinterpret_cast<const uint4 *>(base + b0 + TILE);
    }
    const bool edge = (t == 0) || ((t + 1) * TILE - lead > len);
    // ---- decode ----------------------------------------------------------------------------------------------------
    uint32_t flagacc = 0, popc = 0;
    uint32_t F, R;
    if (PACKED) {   // the block's dword: F = its 16 fields in reverse order (earlier bases at higher bits), R = its complement as it stands
      const uint32_t br = __builtin_bitreverse32(raw.x);
      F = ((br & 0x55555555u) << 1) | ((br >> 1) & 0x55555555u);
      R = ~raw.x;
    } else {
      const uint32_t Fcomp = decode16c(raw, flagacc, popc);
      F = ~Fcomp;
      R = revcomp_of_comp(Fcomp);
    }
    if (PACKED) {
    } else if (!edge) {
      if ((flagacc & 0x04040404u) | (popc != 16u ? 1u : 0u)) st.bad |= 2;
    } else {  // blocks that reach over a read end: only the read's own bytes count
      const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = ibase + 4 * d + b;
          const uint32_t nb = (dw[d] >> (8 * b)) & 0xF;
          if (i >= 0 && i < len && __builtin_popcount(nb) != 1) st.bad |= 2;
        }
    }
    uint32_t Fp = (uint32_t)__shfl_up((int)F, 1, 64), Rp = (uint32_t)__shfl_up((int)R, 1, 64);
    if (lane == 0) Fp = Fc, Rp = Rc;
    Fc = (uint32_t)__builtin_amdgcn_readlane((int)F, 63), Rc = (uint32_t)__builtin_amdgcn_readlane((int)R, 63);
    s.Fr[G & 127] = F, s.Rr[G & 127] = R;

    // ---- the 16 hashes of the block ----------------------------------------------------------------------------------
    uint32_t v[16];
    bool anyinv = false;
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const uint32_t fw = o == 15 ? F : __builtin_amdgcn_alignbit(Fp, F, 2 * (15 - o));
      const uint32_t rv = o == 15 ? R : __builtin_amdgcn_alignbit(R, Rp, 2 * (o + 1));
      anyinv |= fw == rv;
      v[o] = mix32(min(fw, rv));
    }
    uint32_t emask;
    if (!(dbg & 4) && (st.x_drop >= t * TILE - lead - (W + 32) || __ballot(anyinv)))
      emask = blk_tile<A, 2>(s, st, lane, t, F, Fp, R, Rp, v);
    else if (!(dbg & 4) && edge)
      emask = blk_tile<A, 1>(s, st, lane, t, F, Fp, R, Rp, v);
    else
      emask = blk_tile<A, 0>(s, st, lane, t, F, Fp, R, Rp, v);

    // ---- emission: positions in order -> queue -> one lane per emitted entry -------------------------------------------------
    if (dbg & 1) emask = 0;   // (timing experiments only: PGX_BLK_DBG)
    const int ec = __builtin_popcount(emask);
    if (__ballot(emask != 0)) {
      // exclusive prefix of ec over the lanes: six v_add_u32_dpp (round 2 took it from three ballots of ec's bit planes and six
      // mbcnt: twice the instructions)
      const int einc = wave_incl_scan_dpp(ec);
      const int ex = einc - ec;
      const int etot = __builtin_amdgcn_readlane(einc, 63);
      // A burst of ties (a homopolymer or a short-period tandem array makes EVERY position a tied window minimum: up to 1,024 entries of a
      // tile) is queued and staged in pieces of at most EB entries (round 6; through round 5 such a tile sent its read to k_sketch_wave: 10.7 %
      // of the reads of the repeat-seeded set, 4.4 ms per chunk of redo).
      constexpr int EB = QCAP < RSTAGE ? QCAP : RSTAGE;
      const int pbase = ibase - W;
      for (int b0 = 0; b0 < etot; b0 += EB) {   // (wave-uniform; one piece for an ordinary tile)
        const int bn = min(EB, etot - b0);
        int w = ex - b0;
        uint32_t em = emask;
        while (em) {
          const int o = __builtin_ctz(em);
          em &= em - 1;
          if ((uint32_t)w < (uint32_t)bn) s.Q[w] = (uint32_t)(pbase + o);
          ++w;
        }
        __syncthreads();
        if (rst[0].nnew + bn > RSTAGE) {
          if (dbg & 2) rst[0].nnew = 0;
          else fused_flush();
        }
        for (int e0 = 0; e0 < bn; e0 += 64) {
          const int e = e0 + lane;
          if (e < bn) {
            const int ip = (int)s.Q[e];
            const int bb = ip + lead, B = bb >> 4, o = bb & 15;
            const uint64_t fq = ((uint64_t)s.Fr[(B - 1) & 127] << 32) | s.Fr[B & 127];
            const uint64_t rq = ((uint64_t)s.Rr[B & 127] << 32) | s.Rr[(B - 1) & 127];
            const uint32_t fw = (uint32_t)(fq >> (2 * (15 - o))), rv = (uint32_t)(rq >> (2 * (o + 1)));
            const int wq = rst[0].ncarry + rst[0].nnew + e;
            red.h[0][wq] = mix32(min(fw, rv));
            red.y[0][wq] = ((uint32_t)ip << 1) | (fw > rv ? 1u : 0u);
          }
        }
        rst[0].nnew += bn;
        __syncthreads();
      }
    }
  }
  fused_flush();
  // why a read goes to the general kernel (flag bits): 1 shorter than a window + 200, 2 ambiguous base, 4 two drops in one tile,
  // 8 a drop in the first 200 positions, 16 two drops within a tile + two windows, 64 a burst of ties, 128 slab overflow
  uint32_t why = st.bad;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) why |= (uint32_t)__shfl_xor((int)why, o, 64);
  if (lane == 0) {
    counts[slot] = why ? 0u : nout;
    if (why) flags[slot] = why;
  }
}

// host side ------------------------------------------------------------------------------------------------
// the closed-form kernels read the 2-bit packs once a stage has built them (the first overlap stage on the database: pgx_pack.hip) -- always when
// the bytes were released; PGX_SKETCH_PACKED=0 keeps them on the bytes (the parity tests run both forms)
static bool sketch_from_packs(const pgx_seqdb *db) {
  if (!seq_packs_valid(db)) return false;
  if (!db->d_seq.p) return true;
  const char *e = getenv("PGX_SKETCH_PACKED");
  return !(e && atoi(e) == 0);
}
bool sketch_wave_eligible(const ReadDesc &rd, int w, int k) {
  return k == K && (w == 64 || w == 80 || w == 96 || w == 128) && rd.len < (1u << 30);
}

template <bool FUSED, int A>
static void launch_w(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n, pgx_mm128 *d_slab,
                     const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags, int rs, int levels, uint32_t *d_need = nullptr,
                     int off_by_list = 0) {
  if (FUSED && sketch_from_packs(db)) {   // (round 6: from the 2-bit packs once they exist)
    hipLaunchKernelGGL((k_sketch_wave<FUSED, A, true>), dim3(n), dim3(64), 0, ctx().stream, reinterpret_cast<const uint8_t *>(db->d_pack.p), d_reads, d_list, n,
                       d_slab, d_slab_off, d_counts, d_flags, rs, levels, d_need, off_by_list, db->d_poff.p, db->d_nflag.p);
    return;
  }
  PGX_REQUIRE(db->d_seq.p, PGX_ESTATE, "the seqdb's bytes were released (pgx_seqdb_release_bytes): this sketch form needs them");
  hipLaunchKernelGGL((k_sketch_wave<FUSED, A>), dim3(n), dim3(64), 0, ctx().stream, db->d_seq.p, d_reads, d_list, n, d_slab,
                     d_slab_off, d_counts, d_flags, rs, levels, d_need, off_by_list, (const uint64_t *)nullptr, (const uint32_t *)nullptr);
}

void launch_sketch_wave(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int w,
                        int k, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags) {
  (void)k;
  if (!n_list) return;
  switch (w) {
    case 64: launch_w<false, 4>(db, d_reads, d_list, n_list, d_slab, d_slab_off, d_counts, d_flags, 0, 0); break;
    case 80: launch_w<false, 5>(db, d_reads, d_list, n_list, d_slab, d_slab_off, d_counts, d_flags, 0, 0); break;
    case 96: launch_w<false, 6>(db, d_reads, d_list, n_list, d_slab, d_slab_off, d_counts, d_flags, 0, 0); break;
    case 128: launch_w<false, 8>(db, d_reads, d_list, n_list, d_slab, d_slab_off, d_counts, d_flags, 0, 0); break;
    default: PGX_REQUIRE(false, PGX_EARG, "window %d has no closed-form kernel", w);
  }
  PGX_HIP(hipGetLastError());
}

// fused sketch + reduce x levels (w = 80 only); requires rs <= RCARRY + 1 and levels in {1, 2}
bool sketch_fused_supported(int w, int rs, int levels) {
  return w == 80 && rs >= 1 && rs <= RCARRY + 1 && (levels == 1 || levels == 2);
}
void launch_sketch_fused(const pgx_seqdb *db, const ReadDesc *d_reads, uint32_t n, int rs, int levels, pgx_mm128 *d_slab,
                         const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags) {
  if (!n) return;
  launch_w<true, 5>(db, d_reads, nullptr, n, d_slab, d_slab_off, d_counts, d_flags, rs, levels);
  PGX_HIP(hipGetLastError());
}

// round 2: the block-per-lane closed form (fused with the streaming reduce).  Reads it flags are redone by
// launch_sketch_fused_list (k_sketch_wave on the listed slots).
bool sketch_blk_supported(int w, int k, int rs, int levels) { return k == K && sketch_fused_supported(w, rs, levels); }
void launch_sketch_blk(const pgx_seqdb *db, const ReadDesc *d_reads, uint32_t n, int rs, int levels, pgx_mm128 *d_slab,
                       const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags) {
  if (!n) return;
  const int dbg = 0;   // (round 2's timing experiments: bit 1 no emission, 2 no reduce, 4 no edge variants)
  if (sketch_from_packs(db)) {
    hipLaunchKernelGGL((k_sketch_blk<5, true>), dim3(n), dim3(64), 0, ctx().stream, reinterpret_cast<const uint8_t *>(db->d_pack.p), d_reads, n, d_slab,
                       d_slab_off, d_counts, d_flags, rs, levels, dbg, db->d_poff.p, db->d_nflag.p);
  } else {
    PGX_REQUIRE(db->d_seq.p, PGX_ESTATE, "the seqdb's bytes were released (pgx_seqdb_release_bytes) and its packs are gone");
    hipLaunchKernelGGL((k_sketch_blk<5>), dim3(n), dim3(64), 0, ctx().stream, db->d_seq.p, d_reads, n, d_slab, d_slab_off, d_counts,
                       d_flags, rs, levels, dbg, (const uint64_t *)nullptr, (const uint32_t *)nullptr);
  }
  PGX_HIP(hipGetLastError());
}
void launch_sketch_fused_list(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int rs,
                              int levels, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags,
                              uint32_t *d_need, int off_by_list) {
  if (!n_list) return;
  launch_w<true, 5>(db, d_reads, d_list, n_list, d_slab, d_slab_off, d_counts, d_flags, rs, levels, d_need, off_by_list);
  PGX_HIP(hipGetLastError());
}

}  // namespace pgx
