// pgx_sketch_n.hip -- minimizers of reads the closed-form kernels cannot take WHOLE: reads with ambiguous bases (and, as the
// universal exact fallback, anything else a closed-form kernel flagged), wave-parallel.  Replaces round 1-3's one-lane-per-read
// state machine.
//
// What mm_sketch does at an ambiguous base (/root/reference/src/mm_sketch.c:112-113): it resets the run length `l` and nothing else --
// the window ring, its tracked minimum and the two rolling k-mers stay as they are (an ambiguous base shifts nothing into the k-mers).
// Restated as a stream of ENTRIES (one per position that is not a strand-ambiguous k-mer; an entry is a k-mer hash once `l >= k`,
// "infinite" otherwise -- ambiguous bases and the k-1 positions after one):
//   * the tracked minimum is always the RIGHTMOST smallest of the last w entries, whatever happened before (mm_sketch.c:126-147 keep
//     that invariant in all three branches), so the machine has no memory beyond the last w entries and `l`;
//   * everything it emits is gated by l >= w+k-1 (:116,127,130,138), i.e. by a window of w finite entries, which lies inside ONE
//     run of unambiguous bases (a "segment"); the minimum that is pending when an ambiguous base arrives is never emitted;
//   * the end of the sequence emits the tracked minimum unconditionally (:150): the rightmost smallest of the last w entries of the
//     WHOLE stream, which may reach back across ambiguous bases into an earlier segment.
// Hence, exactly (tools/nsketch_model.py: 6,000 adversarial strings against the oracle, incl. k = 4 where one k-mer in sixteen is
// strand-ambiguous):
//   sketch(read) = for every segment, in order: sketch(segment as a read of its own) WITHOUT its last element (that is the
//                  segment's pending minimum: mm_sketch.c:150 of the segment-as-read), positions shifted by the segment's start;
//                  then ONE element: the rightmost smallest finite entry among the last w entries of the read, if any.
// One subtlety makes "segment as a read of its own" exact: the first k-1 positions after an ambiguous base test strand ambiguity on
// k-mers that still hold bases from BEFORE it (nothing was flushed), and a position found ambiguous does not advance `l`.  A fresh
// read never skips there (its k-mers are zero-filled: fwd == rev is impossible before k bases).  So the segment handed to the
// closed-form kernel starts at f-(k-1), f = the position where `l` really reaches k (k_nseg_virtual walks those <= ~2k positions
// with the true k-mer state); from f on both agree position by position.
//
// Kernels: k_nseg_scan (a wavefront per read: ballot the unambiguous positions 64 at a time, segment starts / ends from the bit
// pattern), k_nseg_virtual (a lane per segment), the closed-form sketch kernel of the (w, k) over the segment descriptors
// (k_sketch_wave / k_sketch_general: unchanged), k_nread_end (a lane per read: the last w entries, found from a bounded look-back
// that is widened until it provably covers them), k_nread_assemble (a wavefront per read), and for the fused index path
// k_reduce_long (mm_reduce x levels per read, any length, /root/reference/src/shmr_reduce.c:53-90).
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "pgx_internal.h"

namespace pgx {

void launch_sketch_wave(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int w, int k,
                        pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags);   // pgx_sketch_fast.hip
bool sketch_wave_eligible(const ReadDesc &rd, int w, int k);
void launch_sketch_general(const pgx_seqdb *db, const ReadDesc *d_reads, const std::vector<uint32_t> &lens, const uint32_t *d_list,
                           int w, int k, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags);   // pgx_kernels.hip

namespace {

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ uint64_t kmer_hash(uint64_t key, uint64_t mask) {   // src/mm_sketch.c:23-32
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}
__device__ __forceinline__ int base_code(uint32_t b) {   // one-hot low nibble A=1 C=2 G=4 T=8 (src/shmr_utils.c:18-30); anything else: ambiguous
  b &= 0xF;
  return (b == 1) ? 0 : (b == 2) ? 1 : (b == 4) ? 2 : (b == 8) ? 3 : 4;
}

// the two rolling k-mers as mm_sketch keeps them (:102-103), fed with the up to k unambiguous bases that precede position `at`
// (ambiguous bases shift nothing in), from the all-zero start of a read
struct Kmers {
  uint64_t fwd, rev;
};
__device__ __forceinline__ Kmers kmers_before(const uint8_t *s, uint32_t at, int k, uint64_t mask, uint64_t top) {
  uint64_t ctxc = 0;   // code of the j-th newest base at bits [2j, 2j+1]
  int nctx = 0;
  for (int64_t p = (int64_t)at - 1; p >= 0 && nctx < k; --p) {
    const int c = base_code(s[p]);
    if (c < 4) ctxc |= (uint64_t)c << (2 * nctx), ++nctx;
  }
  Kmers km{0, 0};
  for (int j = nctx - 1; j >= 0; --j) {
    const uint64_t c = (ctxc >> (2 * j)) & 3;
    km.fwd = (km.fwd << 2 | c) & mask;
    km.rev = (km.rev >> 2) | (3ULL ^ c) << top;
  }
  return km;
}

// ---- segments: maximal runs of unambiguous bases, a wavefront per read ------------------------------------------------------------
// A tile of 1 KiB per step: lane l loads the 16 bytes [16 l, 16 l + 16) of the tile with one aligned 16-byte load (the read's first tile
// starts at the 16-byte boundary below the read; bytes outside the read count as ambiguous) and turns them into a 16-bit mask of
// unambiguous positions.  A segment starts where a set bit follows a clear one and ends (exclusively) where a clear bit follows a set
// one -- across lanes through the neighbour's top bit, across tiles through a carried bit.  Counting pass, then (FILL) the same walk
// writes starts / ends at the offsets a wave prefix sum of the per-lane counts gives.  (The first version walked 64 BYTES per step:
// 235 dependent steps for a 15 kb read, 1 ms per pass for 15 k reads.)
__device__ __forceinline__ uint32_t valid16(uint4 v) {   // bit i: byte i holds a one-hot low nibble (exactly one of its four bits set)
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t n = (w[d] >> (8 * b)) & 0xFu;
      m |= ((n != 0 && (n & (n - 1)) == 0) ? 1u : 0u) << (4 * d + b);
    }
  return m;
}
template <bool FILL>
__global__ __launch_bounds__(64) void k_nseg_scan(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                  const uint32_t *__restrict__ list, uint32_t nn, uint64_t *__restrict__ seg_cnt,
                                                  const uint64_t *__restrict__ seg_off, uint32_t *__restrict__ seg_s,
                                                  uint32_t *__restrict__ seg_e, uint32_t *__restrict__ seg_read) {
  const uint32_t it = blockIdx.x;
  if (it >= nn) return;
  const ReadDesc rd = reads[list[it]];
  const int lead = (int)(rd.off & 15);
  const uint8_t *base = seq + (rd.off - (uint64_t)lead);   // (the seqdb buffer is 16-byte aligned and padded: pgx_seqdb_adopt_dev / _upload)
  const int len = (int)rd.len, span = lead + len;
  const int lane = threadIdx.x;
  const uint64_t out0 = FILL ? seg_off[it] : 0;
  uint64_t nstart = 0, nend = 0;
  uint32_t carry = 0;   // the position before the tile is an unambiguous base (wave-uniform)
  for (int b0 = 0; b0 < span; b0 += 1024) {
    const int mine = b0 + lane * 16;              // byte offset (from base) of this lane's 16 bytes
    uint32_t m = 0;
    if (mine < span) {
      m = valid16(*reinterpret_cast<const uint4 *>(base + mine));
      const int lo = lead - mine, hi = span - mine;   // bytes [lo, hi) of the 16 belong to the read
      if (lo > 0) m &= ~((1u << (lo > 16 ? 16 : lo)) - 1u);
      if (hi < 16) m &= (1u << (hi < 0 ? 0 : hi)) - 1u;
    }
    uint32_t prevtop = (uint32_t)__shfl_up((int)(m >> 15), 1, 64);
    if (lane == 0) prevtop = carry;
    const uint32_t pm = ((m << 1) | (prevtop & 1u)) & 0xFFFFu;   // bit i: the position before byte i is unambiguous
    const uint32_t st = m & ~pm, en = ~m & pm & 0xFFFFu;          // (an end bit sits on the first position after a segment)
    const int cs = __builtin_popcount(st), ce = __builtin_popcount(en);
    int is = cs, ie = ce;                                        // inclusive scans over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int ts = __shfl_up(is, o, 64), te = __shfl_up(ie, o, 64);
      if (lane >= o) is += ts, ie += te;
    }
    if (FILL) {
      uint64_t ws = out0 + nstart + (uint64_t)(is - cs), we = out0 + nend + (uint64_t)(ie - ce);
      for (uint32_t x = st; x; x &= x - 1) seg_s[ws] = (uint32_t)(mine + __builtin_ctz(x) - lead), seg_read[ws] = it, ++ws;
      for (uint32_t x = en; x; x &= x - 1) seg_e[we++] = (uint32_t)(mine + __builtin_ctz(x) - lead);
    }
    nstart += (uint64_t)__shfl(is, 63, 64), nend += (uint64_t)__shfl(ie, 63, 64);
    carry = (uint32_t)__shfl((int)(m >> 15), 63, 64) & 1u;
  }
  if (carry) {   // the read ends inside a segment exactly on a tile boundary: its end was not seen
    if (FILL && lane == 0) seg_e[out0 + nend] = (uint32_t)len;
    ++nend;
  }
  if (!FILL && lane == 0) seg_cnt[it] = nstart;
}

// ---- the segment as the closed-form kernel must see it: from f-(k-1), f = where the run length really reaches k ---------------------
__global__ void k_nseg_virtual(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads, const uint32_t *__restrict__ list,
                               const uint32_t *__restrict__ seg_s, const uint32_t *__restrict__ seg_e, const uint32_t *__restrict__ seg_read,
                               uint64_t ns, int k, uint32_t slab_div, uint32_t slab_min, ReadDesc *__restrict__ vdesc,
                               uint32_t *__restrict__ vpos0, uint64_t *__restrict__ vcap) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const ReadDesc rd = reads[list[seg_read[i]]];
  const uint8_t *s = seq + rd.off;
  const uint32_t a = seg_s[i], e = seg_e[i];
  const uint64_t mask = (1ULL << (2 * k)) - 1, top = 2ULL * (uint64_t)(k - 1);
  Kmers km = kmers_before(s, a, k, mask, top);
  int run = 0;
  int64_t f = -1;
  for (uint32_t p = a; p < e; ++p) {
    const uint64_t c = (uint64_t)(base_code(s[p]) & 3);
    km.fwd = (km.fwd << 2 | c) & mask;
    km.rev = (km.rev >> 2) | (3ULL ^ c) << top;
    if (km.fwd != km.rev && ++run == k) {   // (a strand-ambiguous k-mer does not advance the run length: mm_sketch.c:104-105)
      f = (int64_t)p;
      break;
    }
  }
  if (f < 0) {   // shorter than a k-mer: nothing to sketch (an empty, aligned descriptor: the kernels run zero tiles over it)
    vdesc[i] = ReadDesc{rd.off & ~15ULL, 0u, rd.rid};
    vpos0[i] = 0;
    vcap[i] = 0;
    return;
  }
  const uint32_t vs = (uint32_t)(f - (k - 1));
  vdesc[i] = ReadDesc{rd.off + vs, e - vs, rd.rid};
  vpos0[i] = vs;
  vcap[i] = slab_div ? (uint64_t)(e - vs) / slab_div + slab_min : (uint64_t)(e - vs) + 1;   // slab_div 0: one element per position always fits
}

// ---- the element the end of the sequence emits (mm_sketch.c:150): rightmost smallest finite entry of the last w entries -----------
// A lane per read walks a SUFFIX of the read twice (count the entries, then take the minimum over the last w).  Starting in the
// middle of a read the k-mers are exact (kmers_before) but the run length is only known from below: an entry is CERTAIN once an
// ambiguous base was passed (the run length restarted in view) or the lower bound reached k.  The suffix is widened (x 4) until
// the last w entries are all certain, or it is the whole read.
struct EndWalk {
  uint64_t total;      // entries in [s0, len)
  uint64_t certain;    // index of the first certain entry (~0: none)
  uint64_t bx, by;     // pass 2: the rightmost smallest finite entry with index >= lo
  bool has;
};
template <bool PICK>
__device__ __forceinline__ EndWalk end_walk(const uint8_t *s, uint32_t s0, uint32_t len, int k, uint64_t mask, uint64_t top, uint64_t lo,
                                            uint32_t rid) {
  Kmers km = kmers_before(s, s0, k, mask, top);
  bool exact = s0 == 0;
  int64_t runlb = 0;
  EndWalk r{0, ~0ULL, ~0ULL, 0, false};
  for (uint32_t p = s0; p < len; ++p) {
    const int cc = base_code(s[p]);
    bool finite = false;
    if (cc < 4) {
      const uint64_t c = (uint64_t)cc;
      km.fwd = (km.fwd << 2 | c) & mask;
      km.rev = (km.rev >> 2) | (3ULL ^ c) << top;
      if (km.fwd == km.rev) continue;
      ++runlb;
      finite = runlb >= k;
    } else {
      runlb = 0, exact = true;
    }
    if (!PICK) {
      if ((exact || finite) && r.certain == ~0ULL) r.certain = r.total;
    } else if (finite && r.total >= lo) {
      const bool z = km.fwd > km.rev;
      const uint64_t x = kmer_hash(z ? km.rev : km.fwd, mask);
      if (!r.has || x <= r.bx) r.bx = x, r.by = (uint64_t)rid << 32 | (uint64_t)p << 1 | (z ? 1ULL : 0ULL), r.has = true;
    }
    ++r.total;
  }
  return r;
}
__global__ void k_nread_end(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads, const uint32_t *__restrict__ list,
                            uint32_t nn, int w, int k, pgx_mm128 *__restrict__ end_el, uint32_t *__restrict__ end_has) {
  const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= nn) return;
  const ReadDesc rd = reads[list[it]];
  const uint8_t *s = seq + rd.off;
  const uint64_t mask = (1ULL << (2 * k)) - 1, top = 2ULL * (uint64_t)(k - 1);
  uint32_t back = (uint32_t)(w + 3 * k + 16);
  uint32_t s0;
  EndWalk c;
  for (;;) {
    s0 = rd.len > back ? rd.len - back : 0;
    c = end_walk<false>(s, s0, rd.len, k, mask, top, 0, rd.rid);
    if (s0 == 0 || (c.total >= (uint64_t)w && c.certain <= c.total - (uint64_t)w)) break;
    back = back > (1u << 29) ? ~0u : back * 4;
  }
  const EndWalk b = end_walk<true>(s, s0, rd.len, k, mask, top, c.total > (uint64_t)w ? c.total - (uint64_t)w : 0, rd.rid);
  end_has[it] = b.has ? 1u : 0u;
  if (b.has) end_el[it] = pgx_mm128{b.bx << 8 | (uint64_t)k, b.by};
}

// ---- assembly: every segment's list without its last element, positions shifted; then the end element ------------------------------
__global__ void k_nread_total(const uint64_t *__restrict__ seg_off, const uint32_t *__restrict__ vcnt, const uint32_t *__restrict__ end_has,
                              uint32_t nn, uint64_t *__restrict__ tot) {
  const uint32_t it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= nn) return;
  uint64_t t = end_has[it];
  for (uint64_t j = seg_off[it]; j < seg_off[it + 1]; ++j) t += vcnt[j] ? vcnt[j] - 1 : 0;
  tot[it] = t;
}
__global__ __launch_bounds__(64) void k_nread_assemble(const uint64_t *__restrict__ seg_off, const uint32_t *__restrict__ vcnt,
                                                       const uint32_t *__restrict__ vpos0, const pgx_mm128 *__restrict__ vslab,
                                                       const uint64_t *__restrict__ vslab_off, const pgx_mm128 *__restrict__ end_el,
                                                       const uint32_t *__restrict__ end_has, uint32_t nn,
                                                       const uint64_t *__restrict__ out_off, pgx_mm128 *__restrict__ out) {
  const uint32_t it = blockIdx.x;
  if (it >= nn) return;
  pgx_mm128 *dst = out + out_off[it];
  uint64_t w = 0;
  for (uint64_t j = seg_off[it]; j < seg_off[it + 1]; ++j) {
    const uint32_t c = vcnt[j];
    if (c < 2) continue;
    const pgx_mm128 *src = vslab + vslab_off[j];
    const uint64_t shift = (uint64_t)vpos0[j] << 1;
    for (uint32_t i = threadIdx.x; i < c - 1; i += 64) dst[w + i] = pgx_mm128{src[i].x, src[i].y + shift};
    w += c - 1;
  }
  if (threadIdx.x == 0 && end_has[it]) dst[w] = end_el[it];
}

// ---- mm_reduce x levels over ONE read's list of any length, a wavefront per read (src/shmr_reduce.c:53-90) -----------------------
// Element t closes the window [t-rs+1, t] once t >= rs-1; winner = smallest x>>8, ties to the lowest ring slot (index % rs, :42-48);
// emitted iff its y differs from the previous window's winner (:83-88; the first window always emits).  Level l reads buf[l & 1]
// (level 0: the assembled list) and writes buf[(l + 1) & 1], both laid out like the input (same offsets).
__global__ __launch_bounds__(64) void k_reduce_long(const pgx_mm128 *__restrict__ in0, pgx_mm128 *__restrict__ bufA, pgx_mm128 *__restrict__ bufB,
                                                    const uint64_t *__restrict__ off, uint32_t nn, int rs, int levels,
                                                    uint32_t *__restrict__ cnt_out) {
  const uint32_t it = blockIdx.x;
  if (it >= nn) return;
  const int lane = threadIdx.x;
  const uint64_t o = off[it];
  uint64_t n = off[it + 1] - o;
  const pgx_mm128 *src = in0 + o;
  for (int lv = 0; lv < levels; ++lv) {
    pgx_mm128 *dst = ((lv & 1) ? bufB : bufA) + o;
    uint64_t nout = 0;
    uint64_t carry_y = 0;
    for (uint64_t t0 = 0; t0 < n; t0 += 64) {
      const uint64_t t = t0 + lane;
      const bool valid = t < n && t + 1 >= (uint64_t)rs;
      pgx_mm128 best{0, 0};
      if (valid) {
        uint64_t u = t + 1 - rs;
        int sl = (int)(u % (uint64_t)rs);
        best = src[u];
        uint64_t bh = best.x >> 8;
        int bsl = sl;
        for (int j = 1; j < rs; ++j) {
          ++u;
          if (++sl == rs) sl = 0;
          const pgx_mm128 e = src[u];
          const uint64_t h = e.x >> 8;
          if (h < bh || (h == bh && sl < bsl)) best = e, bh = h, bsl = sl;
        }
      }
      uint64_t prevy = (uint64_t)__shfl_up((int)(best.y >> 32), 1, 64) << 32 | (uint32_t)__shfl_up((int)best.y, 1, 64);
      if (lane == 0) prevy = carry_y;
      carry_y = (uint64_t)__shfl((int)(best.y >> 32), 63, 64) << 32 | (uint32_t)__shfl((int)best.y, 63, 64);
      const bool em = valid && (t + 1 == (uint64_t)rs || best.y != prevy);
      const uint64_t m = __ballot(em);
      if (em) dst[nout + (uint64_t)__builtin_popcountll(m & ((1ULL << lane) - 1))] = best;
      nout += (uint64_t)__builtin_popcountll(m);
    }
    __syncthreads();   // (one wavefront per block: this level's global writes before the next level's reads)
    n = nout;
    src = dst;
  }
  if (lane == 0) cnt_out[it] = (uint32_t)n;
}

__global__ void k_set_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ list, uint32_t nn, uint32_t v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nn) dst[list[i]] = v;
}
__global__ void k_scatter_cnt(uint32_t *__restrict__ dst, const uint32_t *__restrict__ list, uint32_t nn, const uint64_t *__restrict__ off) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nn) dst[list[i]] = (uint32_t)(off[i + 1] - off[i]);
}
__global__ void k_scatter_cnt32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ list, uint32_t nn, const uint32_t *__restrict__ cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nn) dst[list[i]] = cnt[i];
}

uint64_t exclusive_offsets(uint64_t *d_vals_then_offs, uint64_t n) {
  // d[0..n) = values on entry; on exit d[0..n] = exclusive prefix sums (n + 1 entries); returns the total
  hipStream_t st = ctx().stream;
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_vals_then_offs, d_vals_then_offs, (int)(n + 1), st));
  void *tmp = ws_raw("nsk.scan_tmp", bytes);
  PGX_HIP(hipMemsetAsync(d_vals_then_offs + n, 0, sizeof(uint64_t), st));
  PGX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, d_vals_then_offs, d_vals_then_offs, (int)(n + 1), st));
  uint64_t total = 0;
  PGX_HIP(hipMemcpyAsync(&total, d_vals_then_offs + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  sync();
  return total;
}

}  // namespace

// The level-0 minimizers of the listed reads (slots of d_reads, ascending), packed in list order: nl0[nl0_off[i] .. nl0_off[i+1]).
void dev_sketch_nreads(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t nn, int w, int k,
                       DevBuf<pgx_mm128> &nl0, DevBuf<uint64_t> &nl0_off, uint64_t *n_total) {
  *n_total = 0;
  nl0_off.alloc((size_t)nn + 1);
  if (nn == 0) return;
  hipStream_t st = ctx().stream;
  PGX_REQUIRE(db->d_seq.p, PGX_ESTATE, "the seqdb's bytes were released (pgx_seqdb_release_bytes): reads with ambiguous bases are sketched from them");
  const uint8_t *seq = db->d_seq.p;
  // segments
  DevBuf<uint64_t> seg_off((size_t)nn + 1);
  hipLaunchKernelGGL(k_nseg_scan<false>, dim3(nn), dim3(64), 0, st, seq, d_reads, d_list, nn, seg_off.p, (const uint64_t *)nullptr,
                     (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
  const uint64_t ns = exclusive_offsets(seg_off.p, nn);
  PGX_REQUIRE(ns < (1ULL << 31), PGX_EARG, "too many runs of unambiguous bases in one chunk (%llu)", (unsigned long long)ns);
  DevBuf<pgx_mm128> end_el(nn);
  DevBuf<uint32_t> end_has(nn);
  hipLaunchKernelGGL(k_nread_end, dim3(cdiv(nn, 64)), dim3(64), 0, st, seq, d_reads, d_list, nn, w, k, end_el.p, end_has.p);
  DevBuf<uint32_t> seg_s(ns), seg_e(ns), seg_read(ns), vpos0(ns), vcnt(ns), vflag(ns);
  DevBuf<ReadDesc> vdesc(ns);
  DevBuf<uint64_t> vslab_off(ns + 1);
  DevBuf<pgx_mm128> vslab;
  if (ns) {
    hipLaunchKernelGGL(k_nseg_scan<true>, dim3(nn), dim3(64), 0, st, seq, d_reads, d_list, nn, (uint64_t *)nullptr, seg_off.p, seg_s.p,
                       seg_e.p, seg_read.p);
    const bool wave = k == 16 && (w == 64 || w == 80 || w == 96 || w == 128);
    std::vector<ReadDesc> hdesc;
    std::vector<uint32_t> lens;
    // slabs of len / 8 + 64 elements (5x the expected density); a low-complexity segment that outgrows its slab sends the whole
    // batch through once more with one element per position, which always fits
    for (int attempt = 0; attempt < 2; ++attempt) {
      hipLaunchKernelGGL(k_nseg_virtual, dim3(cdiv(ns, 256)), dim3(256), 0, st, seq, d_reads, d_list, seg_s.p, seg_e.p, seg_read.p, ns, k,
                         attempt ? 0u : 8u, 64u, vdesc.p, vpos0.p, vslab_off.p);
      const uint64_t cap = exclusive_offsets(vslab_off.p, ns);
      vslab.alloc(cap ? cap : 1);
      PGX_HIP(hipMemsetAsync(vcnt.p, 0, ns * sizeof(uint32_t), st));
      PGX_HIP(hipMemsetAsync(vflag.p, 0, ns * sizeof(uint32_t), st));
      if (wave) {
        launch_sketch_wave(db, vdesc.p, nullptr, (uint32_t)ns, w, k, vslab.p, vslab_off.p, vcnt.p, vflag.p);
      } else {
        if (hdesc.empty()) {
          hdesc.resize(ns);
          vdesc.download(hdesc.data(), ns);
          sync();
          lens.resize(ns);
          for (uint64_t i = 0; i < ns; ++i) lens[i] = hdesc[i].len;
        }
        launch_sketch_general(db, vdesc.p, lens, nullptr, w, k, vslab.p, vslab_off.p, vcnt.p, vflag.p);
      }
      uint32_t *d_nbad = ws<uint32_t>("nsk.nbad", 1);
      PGX_HIP(hipMemsetAsync(d_nbad, 0, sizeof(uint32_t), st));
      size_t rb = 0;
      PGX_HIP(hipcub::DeviceReduce::Max(nullptr, rb, vflag.p, d_nbad, (int)ns, st));
      void *rt = ws_raw("nsk.red_tmp", rb);
      PGX_HIP(hipcub::DeviceReduce::Max(rt, rb, vflag.p, d_nbad, (int)ns, st));
      uint32_t anybad = 0;
      PGX_HIP(hipMemcpyAsync(&anybad, d_nbad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      sync();
      if (!anybad) break;
      PGX_REQUIRE(attempt == 0, PGX_EARG, "a run of unambiguous bases was flagged by the closed-form kernel although its slab holds an element per position");
    }
  }
  // totals, offsets, assembly
  hipLaunchKernelGGL(k_nread_total, dim3(cdiv(nn, 256)), dim3(256), 0, st, seg_off.p, vcnt.p, end_has.p, nn, nl0_off.p);
  const uint64_t total = exclusive_offsets(nl0_off.p, nn);
  nl0.alloc(total ? total : 1);
  hipLaunchKernelGGL(k_nread_assemble, dim3(nn), dim3(64), 0, st, seg_off.p, vcnt.p, vpos0.p, vslab.p, vslab_off.p, end_el.p, end_has.p, nn,
                     nl0_off.p, nl0.p);
  PGX_HIP(hipGetLastError());
  sync();   // (the temporaries above go back to the block cache; stream-ordered reuse makes that safe, the sync keeps error reports local)
  *n_total = total;
}

// the listed reads' minimizers reduced `levels` times per read (any list length): top[off[i] .. off[i] + cnt[i]) -- off = nl0_off
void dev_reduce_nreads(const DevBuf<pgx_mm128> &nl0, const DevBuf<uint64_t> &nl0_off, uint32_t nn, uint64_t total, int rs, int levels,
                       DevBuf<pgx_mm128> &top, DevBuf<uint32_t> &cnt) {
  cnt.alloc(nn ? nn : 1);
  if (nn == 0) return;
  hipStream_t st = ctx().stream;
  DevBuf<pgx_mm128> a(total ? total : 1), b(levels > 1 ? (total ? total : 1) : 1);
  hipLaunchKernelGGL(k_reduce_long, dim3(nn), dim3(64), 0, st, nl0.p, a.p, b.p, nl0_off.p, nn, rs, levels, cnt.p);
  PGX_HIP(hipGetLastError());
  top = std::move((levels & 1) ? a : b);   // level l writes buf[l & 1]: the last level is levels - 1
}

void dev_scatter_counts(uint32_t *d_counts_by_slot, const uint32_t *d_list, uint32_t nn, const uint64_t *d_off, const uint32_t *d_cnt) {
  if (!nn) return;
  if (d_cnt) hipLaunchKernelGGL(k_scatter_cnt32, dim3(cdiv(nn, 256)), dim3(256), 0, ctx().stream, d_counts_by_slot, d_list, nn, d_cnt);
  else hipLaunchKernelGGL(k_scatter_cnt, dim3(cdiv(nn, 256)), dim3(256), 0, ctx().stream, d_counts_by_slot, d_list, nn, d_off);
}
void dev_mark_slots(uint32_t *d_by_slot, const uint32_t *d_list, uint32_t nn, uint32_t v) {
  if (nn) hipLaunchKernelGGL(k_set_u32, dim3(cdiv(nn, 256)), dim3(256), 0, ctx().stream, d_by_slot, d_list, nn, v);
}

}  // namespace pgx
