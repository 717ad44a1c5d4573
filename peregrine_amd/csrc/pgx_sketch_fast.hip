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
// Mapping to the machine.  Phase A (lane = 16 consecutive bases, one aligned 16-byte load per lane per 1 KiB tile):
// nibble -> 2-bit codes, forward and reverse-complement 32-bit packs, k-mers by v_alignbit across the neighbour lane's
// pack, canonical k-mer, 32-bit invertible hash, entries compacted into an LDS ring.  Phase B (lane = chunk of 16
// consecutive ENTRIES): sliding-window minimum WM over the last 80 entries by the chunked prefix/suffix-min method,
// then G(p) == (max of WM over the 80 windows containing p) >= hash(p), again chunked (suffix/prefix max).  Emitted
// entries are appended to the read's slab in global memory.  All cross-lane traffic goes through the LDS ring, which
// is indexed by absolute entry number, so strand-ambiguous k-mers and tile boundaries need no special cases.
#include "pgx_internal.h"

namespace pgx {

namespace {
constexpr int W = 80, K = 16;
constexpr int CH = 16;               // entries per chunk
constexpr int RING_CH = 80;          // chunks in the LDS ring (>= 64 new + 1 partial + 6 pad + 5 lag)
constexpr int RING = RING_CH * CH;   // entries
constexpr int TILE = 1024;           // bases per tile (64 lanes x 16)
constexpr uint32_t INF = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t mix32(uint32_t key) {  // src/mm_sketch.c:23-32 with mask = 2^32-1 (k = 16)
  key = ~key + (key << 21);
  key ^= key >> 24;
  key = key + (key << 3) + (key << 8);
  key ^= key >> 14;
  key = key + (key << 2) + (key << 4);
  key ^= key >> 28;
  key = key + (key << 31);
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
__device__ __forceinline__ int ring_ch(int q) {  // chunk number (may be slightly negative) -> ring chunk slot
  int r = q % RING_CH;
  return r < 0 ? r + RING_CH : r;
}
}  // namespace

__global__ __launch_bounds__(64) void k_sketch_wave(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                    const uint32_t *__restrict__ list, uint32_t n_list,
                                                    pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                                                    uint32_t *__restrict__ counts, uint32_t *__restrict__ flags) {
  __shared__ __attribute__((aligned(16))) uint32_t sH[RING];  // hash of entry e at e % RING
  __shared__ __attribute__((aligned(16))) uint32_t sP[RING];  // lastPos<<1 | strand
  __shared__ __attribute__((aligned(16))) uint32_t sW[RING];  // window minimum for the window ENDING at entry e
  __shared__ uint32_t sC[RING_CH];                            // per chunk: min hash
  __shared__ uint32_t sM[RING_CH];                            // per chunk: max of sW
  const int lane = threadIdx.x;
  if (blockIdx.x >= n_list) return;
  const uint32_t slot = list[blockIdx.x];
  const ReadDesc rd = reads[slot];
  const int len = (int)rd.len;
  const int lead = (int)(rd.off & 15);
  const uint8_t *base = seq + (rd.off - (uint64_t)lead);
  const int span = lead + len;
  const int ntiles = (span + TILE - 1) / TILE;
  pgx_mm128 *out = slab + slab_off[slot];
  const uint32_t cap = (uint32_t)(slab_off[slot + 1] - slab_off[slot]);

  int E = 0;        // entries produced so far
  int Er = 0;       // E % RING
  int wdone = 0;    // chunks whose window minima are in sW
  int ddone = 0;    // chunks already decided
  uint32_t nout = 0;
  uint32_t bad = 0;
  uint32_t Fcarry = 0, Rcarry = 0;

  for (int t = 0; t < ntiles; ++t) {
    // ------------------------------------------------------------------------------------------------------
    // phase A: 16 bases per lane -> up to 16 entries per lane, compacted into the ring
    // ------------------------------------------------------------------------------------------------------
    const int b0 = t * TILE + lane * 16;  // byte offset from `base`
    const int i0 = b0 - lead;             // read position of this lane's first base
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (b0 < span) raw = *reinterpret_cast<const uint4 *>(base + b0);
    const uint32_t dw[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t F = 0;
    const bool inside = i0 >= 0 && i0 + 16 <= len;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t n = dw[d] & 0x0F0F0F0Fu;  // forward-strand one-hot nibbles (src/shmr_utils.c:18-30)
      // one-hot {1,2,4,8} -> {0,1,2,3}: (n>>1) - (n>>3), bytewise
      const uint32_t c = ((n >> 1) & 0x07070707u) - ((n >> 3) & 0x01010101u);
      // ambiguity check: a nibble that is zero or has two bits set
      const uint32_t tt = n - 0x01010101u;
      uint32_t bd = (tt & ~n & 0x80808080u) | (n & tt);
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
      // pack: earlier bases at higher bits (kmer = kmer<<2 | c)
      const uint32_t f8 = ((c << 6) | (c >> 4) | (c >> 14) | (c >> 24)) & 0xFFu;
      F = (F << 8) | f8;
    }
    const uint32_t rr = __builtin_bitreverse32(~F);
    const uint32_t R = ((rr & 0x55555555u) << 1) | ((rr >> 1) & 0x55555555u);  // complement, later bases higher
    uint32_t Fp = (uint32_t)__shfl_up((int)F, 1, 64), Rp = (uint32_t)__shfl_up((int)R, 1, 64);
    if (lane == 0) Fp = Fcarry, Rp = Rcarry;
    Fcarry = (uint32_t)__builtin_amdgcn_readlane((int)F, 63);
    Rcarry = (uint32_t)__builtin_amdgcn_readlane((int)R, 63);

    uint32_t h[16];
    uint32_t vmask = 0, zmask = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t fw = __builtin_amdgcn_alignbit(Fp, F, 2 * (15 - j));
      const uint32_t rv = (j == 15) ? R : __builtin_amdgcn_alignbit(R, Rp, 2 * (j + 1));
      h[j] = mix32(min(fw, rv));
      const int i = i0 + j;
      if (fw != rv && i >= K - 1 && i < len) vmask |= 1u << j;  // strand-ambiguous k-mers are not entries
      if (fw > rv) zmask |= 1u << j;
    }
    const int cnt = __builtin_popcount(vmask);
    const int incl = wave_incl_scan(cnt, lane);
    const int total = __shfl(incl, 63, 64);
    {
      int r = Er + (incl - cnt);
      if (r >= RING) r -= RING;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (vmask & (1u << j)) {
          sH[r] = h[j];
          sP[r] = ((uint32_t)(i0 + j) << 1) | ((zmask >> j) & 1u);
          if (++r == RING) r = 0;
        }
      }
    }
    E += total;
    Er += total;
    if (Er >= RING) Er -= RING;
    const bool last = (t == ntiles - 1);
    const int hch = last ? (E + CH - 1) / CH : E / CH;  // chunks whose hashes are final
    if (last) {                                          // pad the tail of the last chunk
      const int e = E + lane;
      if (lane < CH && e < hch * CH) sH[e % RING] = INF;
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------
    // phase B1: window minima.  WM[16q+o] = min( suffix-min of chunk q-5 from o+1, min(c[q-4..q-1]), prefix-min of
    // chunk q up to o ).  Windows that are not full (end < w-1) or end beyond the last entry get 0.
    // ------------------------------------------------------------------------------------------------------
    for (int q0 = wdone; q0 < hch; q0 += 64) {  // pass 1: chunk minima
      const int q = q0 + lane;
      if (q < hch) {
        uint32_t v[16];
        lds_read16(&sH[ring_ch(q) * CH], v);
        uint32_t c = v[0];
#pragma unroll
        for (int o = 1; o < 16; ++o) c = min(c, v[o]);
        sC[ring_ch(q)] = c;
      }
    }
    __syncthreads();
    for (int q0 = wdone; q0 < hch; q0 += 64) {  // pass 2
      const int q = q0 + lane;
      if (q < hch) {
        uint32_t v[16], s[17], wm[16];
        lds_read16(&sH[ring_ch(q) * CH], v);
#pragma unroll
        for (int o = 0; o < 17; ++o) s[o] = INF;
        uint32_t m4 = INF;
        if (q >= 5) {
          uint32_t u[16];
          lds_read16(&sH[ring_ch(q - 5) * CH], u);
          s[15] = u[15];
#pragma unroll
          for (int o = 14; o >= 1; --o) s[o] = min(s[o + 1], u[o]);
        }
#pragma unroll
        for (int d = 1; d <= 4; ++d)
          if (q - d >= 0) m4 = min(m4, sC[ring_ch(q - d)]);
        uint32_t p = INF, mx = 0;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          p = min(p, v[o]);
          uint32_t x = min(min(s[o + 1], m4), p);
          const int jj = q * CH + o;
          if (jj < W - 1 || jj >= E) x = 0;  // E is the final entry count whenever a chunk can extend past it
          wm[o] = x;
          mx = max(mx, x);
        }
        lds_write16(&sW[ring_ch(q) * CH], wm);
        sM[ring_ch(q)] = mx;
      }
    }
    wdone = hch;
    if (last) {  // windows past the end do not exist: six all-zero chunks
      if (lane < 6 * CH) sW[(ring_ch(hch) * CH + lane) % RING] = 0;
      if (lane < 32) sW[(ring_ch(hch) * CH + 64 + lane) % RING] = 0;
      if (lane < 6) sM[ring_ch(hch + lane)] = 0;
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------
    // phase B2: decide and emit.  G(p) == max(WM over windows ending at p..p+79) >= hash(p).
    // ------------------------------------------------------------------------------------------------------
    const int dlimit = last ? hch : (wdone - 5 > 0 ? wdone - 5 : 0);
    const bool short_read = last && E < W;  // fewer than w entries: emit only the rightmost smallest
    for (int q0 = ddone; q0 < dlimit; q0 += 64) {
      const int q = q0 + lane;
      uint32_t emask = 0;
      uint32_t v[16];
      if (q < dlimit) {
        uint32_t wq[16], wn[16];
        lds_read16(&sH[ring_ch(q) * CH], v);
        lds_read16(&sW[ring_ch(q) * CH], wq);
        lds_read16(&sW[ring_ch(q + 5) * CH], wn);
        uint32_t m4 = 0;
#pragma unroll
        for (int d = 1; d <= 4; ++d) m4 = max(m4, sM[ring_ch(q + d)]);
        uint32_t sm[16];
        sm[15] = wq[15];
#pragma unroll
        for (int o = 14; o >= 0; --o) sm[o] = max(sm[o + 1], wq[o]);
        uint32_t pm = 0;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const uint32_t f = max(max(sm[o], m4), pm);  // pm = prefix max of chunk q+5 up to o-1
          pm = max(pm, wn[o]);
          if (f >= v[o] && q * CH + o < E) emask |= 1u << o;
        }
        if (short_read) emask = 0;
      }
      if (q0 == 0) {  // first-window correction / short-read rule; entries 0..79 sit at ring slots 0..79
        const int lim = short_read ? E : W - 1;  // candidates are entries [0, lim)
        const uint32_t a = lane < lim ? sH[lane] : INF;
        const uint32_t b = lane + 64 < lim ? sH[lane + 64] : INF;
        const uint32_t mv = wave_min_u32(min(a, b));
        const uint64_t mb = __ballot(lane + 64 < lim && b == mv), ma = __ballot(lane < lim && a == mv);
        const int m = mb ? 64 + (63 - __builtin_clzll(mb)) : (ma ? 63 - __builtin_clzll(ma) : -1);
        if (short_read) {
          if (m >= 0 && q == m / CH) emask = 1u << (m % CH);
        } else if (q < 5 && q < dlimit) {
          const uint32_t e79 = sH[W - 1];
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
      const int einc = wave_incl_scan(ec, lane);
      const int etot = __shfl(einc, 63, 64);
      if (etot) {
        uint32_t w = nout + (uint32_t)(einc - ec);
        if (nout + (uint32_t)etot <= cap) {
          const uint32_t *pp = &sP[ring_ch(q) * CH];
#pragma unroll
          for (int o = 0; o < 16; ++o)
            if (emask & (1u << o)) {
              pgx_mm128 e;
              e.x = ((uint64_t)v[o] << 8) | (uint64_t)K;
              e.y = ((uint64_t)rd.rid << 32) | (uint64_t)pp[o];
              out[w++] = e;
            }
        } else {
          bad |= 1;  // slab overflow: the literal kernel redoes this read
        }
        nout += (uint32_t)etot;
      }
    }
    ddone = dlimit;
    __syncthreads();
  }
  const uint64_t anybad = __ballot(bad != 0);
  if (lane == 0) {
    counts[slot] = anybad ? 0u : nout;
    if (anybad) flags[slot] = 1;
  }
}

// host side ------------------------------------------------------------------------------------------------
bool sketch_wave_eligible(const ReadDesc &rd, int w, int k) { return w == W && k == K && rd.len < (1u << 30); }

void launch_sketch_wave(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int w,
                        int k, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags) {
  (void)w, (void)k;
  if (!n_list) return;
  hipLaunchKernelGGL(k_sketch_wave, dim3(n_list), dim3(64), 0, ctx().stream, db->d_seq.p, d_reads, d_list, n_list, d_slab,
                     d_slab_off, d_counts, d_flags);
  PGX_HIP(hipGetLastError());
}

}  // namespace pgx
