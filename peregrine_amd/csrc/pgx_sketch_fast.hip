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
__device__ __forceinline__ uint32_t mix32(uint32_t key) {  // src/mm_sketch.c:23-32 with mask = 2^32-1 (k = 16)
  key = lshl_add<21>(key, ~key);                  // ~key + (key << 21)
  key ^= key >> 24;
  key = lshl_add<8>(key, lshl_add<3>(key, key));  // key + (key<<3) + (key<<8)   (no v_mul_lo)
  key ^= key >> 14;
  key = lshl_add<4>(key, lshl_add<2>(key, key));  // key + (key<<2) + (key<<4)
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
    if (valid) {
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
template <bool FUSED, int A>
__global__ __launch_bounds__(64) void k_sketch_wave(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                    const uint32_t *__restrict__ list, uint32_t n_list,
                                                    pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                                                    uint32_t *__restrict__ counts, uint32_t *__restrict__ flags, int rs,
                                                    int levels) {
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
  const int lead = (int)(rd.off & 15);
  const uint8_t *base = seq + (rd.off - (uint64_t)lead);
  const int span = lead + len;
  const int ntiles = (span + TILE - 1) / TILE;
  pgx_mm128 *out = slab + slab_off[slot];
  const uint32_t cap = (uint32_t)(slab_off[slot + 1] - slab_off[slot]);

  int E = 0;       // entries produced so far
  int qbase = 0;   // absolute chunk number of buffer chunk 0
  int wdone = 0;   // chunks whose window minima are in Wm
  int ddone = 0;   // chunks already decided
  uint32_t nout = 0;
  uint32_t bad = 0;
  uint32_t Fkeep = 0, Rkeep = 0;  // this lane's packs of the previous tile (lane 63's become block -1)

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
  if (lane * 16 < span) raw_next = *reinterpret_cast<const uint4 *>(base + lane * 16);
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
    if (b0 + TILE < span) raw_next = *reinterpret_cast<const uint4 *>(base + b0 + TILE);
    const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t F = 0;
    const bool inside = i0 >= 0 && i0 + 16 <= len;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
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
      if (q0 == 0) {  // first-window correction / short-read rule; entries 0..79 are buffer chunks 0..4 (qbase == 0)
        const int lim = short_read ? E : W - 1;  // candidates are entries [0, lim)
        const uint32_t a = lane < lim ? s.H[(lane >> 4) * CST + (lane & 15)] : INF;
        const uint32_t b = lane + 64 < lim ? s.H[(4 + (lane >> 4)) * CST + (lane & 15)] : INF;  // W <= 128 = 2 x 64 lanes
        const uint32_t mv = wave_min_u32(min(a, b));
        const uint64_t mb = __ballot(lane + 64 < lim && b == mv), ma = __ballot(lane < lim && a == mv);
        const int m = mb ? 64 + (63 - __builtin_clzll(mb)) : (ma ? 63 - __builtin_clzll(ma) : -1);
        if (short_read) {
          if (m >= 0 && q == m / CH) emask = 1u << (m % CH);
        } else if (q < A && q < dlimit) {
          const uint32_t e79 = s.H[((W - 1) >> 4) * CST + ((W - 1) & 15)];
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            const int pidx = q * CH + o;
            if (pidx <= W - 2 && v[o] == mv) {
              if (pidx != m) emask |= 1u << o;
              else if (e79 > mv) emask |= 1u << o;
              else emask &= ~(1u << o);
            }
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
            bad |= 1;  // slab overflow: the literal kernel redoes this read
          }
          nout += (uint32_t)etot;
        } else {
          if (etot > RSTAGE) {
            bad |= 1;  // a burst of ties (low-complexity read): the general path redoes this read
          } else {
            if (rst[0].nnew + etot > RSTAGE) fused_flush();
            int w = rst[0].ncarry + rst[0].nnew + (einc - ec);
            uint32_t em = emask;
            while (em) {
              const int o = __builtin_ctz(em);
              em &= em - 1;
              const uint32_t pz = s.P[bq * CST + o];
              const int i = imax - ((imax - (int)(pz >> 1)) & 0x7FFF);
              red.h[0][w] = s.H[bq * CST + o];
              red.y[0][w] = ((uint32_t)i << 1) | (pz & 1u);
              ++w;
            }
            rst[0].nnew += etot;
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
  }
}

// host side ------------------------------------------------------------------------------------------------
bool sketch_wave_eligible(const ReadDesc &rd, int w, int k) {
  return k == K && (w == 64 || w == 80 || w == 96 || w == 128) && rd.len < (1u << 30);
}

template <bool FUSED, int A>
static void launch_w(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n, pgx_mm128 *d_slab,
                     const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags, int rs, int levels) {
  hipLaunchKernelGGL((k_sketch_wave<FUSED, A>), dim3(n), dim3(64), 0, ctx().stream, db->d_seq.p, d_reads, d_list, n, d_slab,
                     d_slab_off, d_counts, d_flags, rs, levels);
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

}  // namespace pgx
