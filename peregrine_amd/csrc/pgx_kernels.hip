// pgx_kernels.hip -- gfx950 device code of libpgx.so (index stage + banded O(ND) confirmation).
//
// Kernels (each cites the reference routine whose results it must reproduce bit-for-bit):
//   (reads with ambiguous bases -- and whatever else a closed-form kernel flags -- are cut into runs of unambiguous bases that
//    the same closed-form kernels sketch; pgx_sketch_n.hip)
//   k_sketch_general : mm_sketch      closed form for any (w, k), one wavefront per read, entries in global scratch
//   k_sketch_wave    : mm_sketch      closed form, one wavefront per read            (pgx_sketch_fast.hip)
//   k_reduce_*       : mm_reduce      src/shmr_reduce.c:53-90
//   count            : mm_count       src/shmr_utils.c:131-160  radix sort + run-length
//   k_align4         : ovlp_match     src/DWmatch.c:66-204      four candidates per wavefront          (pgx_align.hip)
#include <chrono>

#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "pgx_internal.h"

namespace pgx {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// =========================================================================================================
// minimizer hash (src/mm_sketch.c:23-32), 64-bit form for any k <= 28
// =========================================================================================================
__device__ __forceinline__ uint64_t mix64(uint64_t key, uint64_t mask) {
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}

__device__ __forceinline__ int code_of_nibble(uint32_t b) {
  // seqdb low nibble is one-hot A=1 C=2 G=4 T=8 (src/shmr_utils.c:18-30); anything else decodes to 'N'
  b &= 0xF;
  return (b == 1) ? 0 : (b == 2) ? 1 : (b == 4) ? 2 : (b == 8) ? 3 : 4;
}

// =========================================================================================================
// mm_reduce (src/shmr_reduce.c:53-90), data-parallel restatement:
//   element t of a read segment [s, e) closes the window [t-rs+1, t] once t-s >= rs-1; the winner is the
//   smallest x>>8 with ties to the lowest ring slot ((t'-s) % rs); it is emitted iff its y differs from the
//   winner of the previous window (== the last emitted element; the first window of a read always emits).
// =========================================================================================================
__global__ void k_mark_starts(const pgx_mm128 *__restrict__ in, size_t n, uint8_t *__restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  flag[t] = (t == 0) || ((in[t].y >> 32) != (in[t - 1].y >> 32));
}

__device__ __forceinline__ size_t seg_start_of(const uint64_t *starts, uint32_t nseg, size_t t) {
  uint32_t lo = 0, hi = nseg;  // last start <= t
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (starts[mid] <= t) lo = mid;
    else hi = mid;
  }
  return starts[lo];
}

__device__ __forceinline__ pgx_mm128 reduce_winner(const pgx_mm128 *in, size_t s, size_t t, int rs) {
  // window elements t-rs+1 .. t ; slot of element u is (u - s) % rs
  pgx_mm128 best = in[t - rs + 1];
  uint64_t bh = best.x >> 8;
  int bslot = (int)((t - rs + 1 - s) % (size_t)rs);
  for (int j = 1; j < rs; ++j) {
    const size_t u = t - rs + 1 + j;
    const pgx_mm128 e = in[u];
    const uint64_t h = e.x >> 8;
    const int slot = (int)((u - s) % (size_t)rs);
    if (h < bh || (h == bh && slot < bslot)) best = e, bh = h, bslot = slot;
  }
  return best;
}

__global__ void k_reduce_flag(const pgx_mm128 *__restrict__ in, size_t n, const uint64_t *__restrict__ starts,
                              uint32_t nseg, int rs, pgx_mm128 *__restrict__ win, uint8_t *__restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t s = seg_start_of(starts, nseg, t);
  const size_t off = t - s;
  uint8_t f = 0;
  pgx_mm128 w{0, 0};
  if (off >= (size_t)rs - 1) {
    w = reduce_winner(in, s, t, rs);
    if (off == (size_t)rs - 1) f = 1;
    else f = reduce_winner(in, s, t - 1, rs).y != w.y;
  }
  win[t] = w;
  flag[t] = f;
}

// generic temp-storage helper for hipcub
struct CubTemp {
  DevBuf<uint8_t> buf;
  void *get(size_t bytes) {
    if (bytes > buf.n) buf.alloc(bytes + (bytes >> 2) + 256);
    return buf.p;
  }
};

using CountIt = hipcub::CountingInputIterator<uint64_t, ptrdiff_t>;

void dev_reduce(const pgx_mm128 *d_in, size_t n, int rs, DevBuf<pgx_mm128> &out, size_t &n_out) {
  n_out = 0;
  if (n == 0) { out.alloc(0); return; }
  hipStream_t st = ctx().stream;
  KernelTimer tm("reduce", n);
  DevBuf<uint8_t> flag(n);
  DevBuf<uint64_t> starts(n);  // worst case every element its own read
  DevBuf<uint64_t> d_num(1);
  CubTemp tmp;
  hipLaunchKernelGGL(k_mark_starts, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, n, flag.p);
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, CountIt(0), flag.p, starts.p, d_num.p, (int)n, st));
  PGX_HIP(hipcub::DeviceSelect::Flagged(tmp.get(bytes), bytes, CountIt(0), flag.p, starts.p, d_num.p, (int)n, st));
  uint64_t nseg = 0;
  d_num.download(&nseg, 1);
  sync();
  DevBuf<pgx_mm128> win(n);
  hipLaunchKernelGGL(k_reduce_flag, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, n, starts.p, (uint32_t)nseg, rs, win.p,
                     flag.p);
  out.alloc(n);
  bytes = 0;
  PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, win.p, flag.p, out.p, d_num.p, (int)n, st));
  PGX_HIP(hipcub::DeviceSelect::Flagged(tmp.get(bytes), bytes, win.p, flag.p, out.p, d_num.p, (int)n, st));
  uint64_t m = 0;
  d_num.download(&m, 1);
  sync();
  n_out = (size_t)m;
}

// =========================================================================================================
// mm_count: multiplicity of x>>8.  Radix sort + run-length encode; output sorted by mer.
// =========================================================================================================
__global__ void k_extract_hash(const pgx_mm128 *__restrict__ in, size_t n, uint64_t *__restrict__ keys) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) keys[t] = in[t].x >> 8;
}
__global__ void k_pack_counts(const uint64_t *__restrict__ mer, const uint32_t *__restrict__ cnt, size_t n,
                              pgx_mm_count *__restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = pgx_mm_count{mer[t], cnt[t], 0u};
}

void dev_count(const pgx_mm128 *d_in, size_t n, int kmer_bits, DevBuf<pgx_mm_count> &out, size_t &n_out) {
  n_out = 0;
  if (n == 0) { out.alloc(0); return; }
  hipStream_t st = ctx().stream;
  KernelTimer tm("count", n);
  DevBuf<uint64_t> keys(n), sorted(n), uniq(n);
  DevBuf<uint32_t> cnt(n);
  DevBuf<uint64_t> d_num(1);
  CubTemp tmp;
  hipLaunchKernelGGL(k_extract_hash, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, n, keys.p);
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, keys.p, sorted.p, (int)n, 0, kmer_bits, st));
  PGX_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.get(bytes), bytes, keys.p, sorted.p, (int)n, 0, kmer_bits, st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, bytes, sorted.p, uniq.p, cnt.p, d_num.p, (int)n, st));
  PGX_HIP(hipcub::DeviceRunLengthEncode::Encode(tmp.get(bytes), bytes, sorted.p, uniq.p, cnt.p, d_num.p, (int)n, st));
  uint64_t m = 0;
  d_num.download(&m, 1);
  sync();
  n_out = (size_t)m;
  out.alloc(n_out);
  if (n_out) hipLaunchKernelGGL(k_pack_counts, dim3(cdiv(n_out, 256)), dim3(256), 0, st, uniq.p, cnt.p, n_out, out.p);
}

// =========================================================================================================
// sketch driver
// =========================================================================================================
bool sketch_wave_eligible(const ReadDesc &rd, int w, int k);  // pgx_sketch_fast.hip
void launch_sketch_wave(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int w,
                        int k, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags);
bool sketch_fused_supported(int w, int rs, int levels);
bool sketch_blk_supported(int w, int k, int rs, int levels);
void launch_sketch_blk(const pgx_seqdb *db, const ReadDesc *d_reads, uint32_t n, int rs, int levels, pgx_mm128 *d_slab,
                       const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags);
void launch_sketch_fused_list(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t n_list, int rs,
                              int levels, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags,
                              uint32_t *d_need = nullptr, int off_by_list = 0);
void launch_sketch_fused(const pgx_seqdb *db, const ReadDesc *d_reads, uint32_t n, int rs, int levels, pgx_mm128 *d_slab,
                         const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags);

__global__ void k_need_of_list(const uint32_t *__restrict__ need, const uint32_t *__restrict__ list, uint32_t n, uint64_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = need[list[i]];
}
// skip (optional): reads whose elements live elsewhere (the exact-size slabs of the second redo pass); off_by_list: slab_off is
// indexed by the position in `list`
__global__ void k_gather_slabs(const pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                               const uint32_t *__restrict__ list, uint32_t n_list, const uint32_t *__restrict__ counts,
                               const uint64_t *__restrict__ out_off, pgx_mm128 *__restrict__ out,
                               const uint32_t *__restrict__ skip = nullptr, int off_by_list = 0) {
  if (blockIdx.x >= n_list) return;
  const uint32_t slot = list ? list[blockIdx.x] : blockIdx.x;
  if (skip && skip[slot]) return;
  const pgx_mm128 *src = slab + slab_off[off_by_list ? blockIdx.x : slot];
  pgx_mm128 *dst = out + out_off[slot];
  const uint32_t n = counts[slot];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// Running extremum within blocks of w consecutive elements (block b = [b w, (b+1) w)), the whole wavefront walking the array 64
// elements at a time with coalesced accesses: a segmented Hillis-Steele scan (segment heads at multiples of w) plus a carry
// between the 64-element slices.  FWD: dst[i] = op(src[block start .. i]);  !FWD: dst[i] = op(src[i .. block end]).
template <bool MIN, bool FWD>
__device__ __forceinline__ void block_running_extremum(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, int count, int w,
                                                       int lane) {
  const uint64_t neutral = MIN ? ~0ULL : 0ULL;
  auto op = [](uint64_t a, uint64_t b) { return MIN ? (a < b ? a : b) : (a > b ? a : b); };
  const int nslice = (count + 63) / 64;
  uint64_t carry = neutral;
  for (int sl = 0; sl < nslice; ++sl) {
    const int c0 = (FWD ? sl : nslice - 1 - sl) * 64;
    const int i = c0 + (FWD ? lane : 63 - lane);  // scan order: ascending i when FWD, descending otherwise
    uint64_t v = i < count ? src[i] : neutral;
    // a segment starts at this lane (in scan order) when i is the first (FWD) / last (!FWD) element of its block
    int f = i < count && (FWD ? (i % w == 0) : (i % w == w - 1 || i == count - 1)) ? 1 : 0;
    const int head = f;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t vu = (uint64_t)__shfl_up((int)(v >> 32), d, 64) << 32 | (uint32_t)__shfl_up((int)v, d, 64);
      const int fu = __shfl_up(f, d, 64);
      if (lane >= d) {
        if (!f) v = op(v, vu);
        f |= fu;
      }
    }
    if (!f) v = op(v, carry);  // no segment head at or before this lane inside the slice: the open segment continues
    if (i < count) dst[i] = v;
    (void)head;
    carry = (uint64_t)__shfl((int)(v >> 32), 63, 64) << 32 | (uint32_t)__shfl((int)v, 63, 64);
  }
}

// =========================================================================================================
// k_sketch_general: the closed form of pgx_sketch_fast.hip (see its header) for ANY window and k-mer size
// (0 < w < 256, 0 < k <= 28; pg_run.py exposes both as --shimmer-w / --shimmer-k), one wavefront per read, three
// passes over the read's ENTRIES (non strand-ambiguous k-mers, numbered in position order) kept in global scratch:
//   1. entries: every lane builds the k-mer ending at its base, canonical strand, 64-bit hash (mm_sketch.c:23-32);
//      ballot-compacted to H[] (hash) and PY[] (position << 1 | strand);
//   2. WM[s] = minimum hash of the full window of w entries starting at s, and GX[p] = maximum of WM over the windows that
//      contain p, both in O(1) per entry from running extrema over blocks of w (van Herk / Gil-Werman);
//   3. entry p is emitted iff GX[p] == H[p] (some full window containing it has its hash as minimum), corrected for the first window
//      (m = rightmost smallest of entries 0..w-2: its ties are always emitted, m itself iff H[w-1] > H[m]); a read with
//      fewer than w entries emits only its rightmost smallest entry.
// O(k) work per base for the k-mers, O(1) per entry for the windows.
// Reads with an ambiguous base (the state machine restarts there) or more minimizers than their slab holds are
// flagged; the caller cuts them into runs of unambiguous bases / gives them larger slabs (pgx_sketch_n.hip).
// =========================================================================================================
__global__ __launch_bounds__(64) void k_sketch_general(const uint8_t *__restrict__ seq, const ReadDesc *__restrict__ reads,
                                                       const uint32_t *__restrict__ list, uint32_t n_list, int w, int k,
                                                       const uint64_t *__restrict__ scr_off, uint64_t *__restrict__ Hs,
                                                       uint32_t *__restrict__ PYs, uint64_t *__restrict__ WMs,
                                                       uint64_t *__restrict__ T1s, uint64_t *__restrict__ T2s,
                                                       pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                                                       uint32_t *__restrict__ counts, uint32_t *__restrict__ flags) {
  const int lane = threadIdx.x;
  const uint64_t mask = (1ULL << (2 * k)) - 1, top = 2ULL * (uint64_t)(k - 1);
  for (uint32_t it = blockIdx.x; it < n_list; it += gridDim.x) {
    const uint32_t slot = list ? list[it] : it;
    const ReadDesc rd = reads[slot];
    const uint8_t *s = seq + rd.off;
    const int len = (int)rd.len;
    uint64_t *H = Hs + scr_off[it], *WM = WMs + scr_off[it], *T1 = T1s + scr_off[it], *T2 = T2s + scr_off[it];
    uint32_t *PY = PYs + scr_off[it];
    // ---- pass 1: every lane rolls the two k-mers over 16 consecutive bases (k-1 bases of run-in per lane) -----------
    int n = 0;
    bool bad = false;
    for (int t0 = 0; t0 < len; t0 += 64 * 16) {
      const int start = t0 + lane * 16;
      uint64_t hh[16];
      uint32_t vmask = 0, smask = 0;
      if (start < len) {
        uint64_t fwd = 0, rev = 0;
        for (int j = start - (k - 1) > 0 ? start - (k - 1) : 0; j < start; ++j) {
          const uint64_t c = (uint64_t)(code_of_nibble(s[j]) & 3);
          fwd = (fwd << 2 | c) & mask;
          rev = (rev >> 2) | (3ULL ^ c) << top;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int i = start + u;
          hh[u] = 0;
          if (i < len) {
            const int cc = code_of_nibble(s[i]);
            if (cc > 3) bad = true;
            const uint64_t c = (uint64_t)(cc & 3);
            fwd = (fwd << 2 | c) & mask;
            rev = (rev >> 2) | (3ULL ^ c) << top;
            if (i >= k - 1 && fwd != rev) {
              const uint32_t strand = fwd < rev ? 0u : 1u;
              hh[u] = mix64(strand ? rev : fwd, mask);
              vmask |= 1u << u, smask |= strand << u;
            }
          }
        }
      }
      const int cnt = __builtin_popcount(vmask);
      int incl = cnt;  // wave inclusive scan of the per-lane entry counts
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      int r = n + incl - cnt;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (vmask >> u & 1u) {
          H[r] = hh[u];
          PY[r] = (uint32_t)(start + u) << 1 | (smask >> u & 1u);
          ++r;
        }
      n += __shfl(incl, 63, 64);
    }
    if (__ballot(bad)) {
      if (lane == 0) counts[slot] = 0, flags[slot] = 1;
      continue;
    }
    __syncthreads();  // (one wavefront per block: orders this wave's global writes before its reads below)
    // ---- pass 2: window minima in O(1) per entry (van Herk / Gil-Werman over blocks of w entries) ---------------------
    // T1 = running minimum from the start of the entry's block, T2 = running minimum from the end of its block;
    // the window [s, s+w) is the tail of block(s) plus the head of the next block: WM[s] = min(T2[s], T1[s+w-1]).
    const int nwin = n - w + 1;  // number of full windows (<= 0: short read)
    if (nwin > 0) {
      block_running_extremum<true, true>(H, T1, n, w, lane);
      block_running_extremum<true, false>(H, T2, n, w, lane);
      __syncthreads();
      for (int v0 = 0; v0 < nwin; v0 += 64) {
        const int v = v0 + lane;
        if (v < nwin) WM[v] = (v % w == 0) ? T1[v + w - 1] : min(T2[v], T1[v + w - 1]);
      }
      __syncthreads();
      // the same over WM with maxima: GX(p) = max of WM over the (up to w) windows that contain entry p
      block_running_extremum<false, true>(WM, T1, nwin, w, lane);
      block_running_extremum<false, false>(WM, T2, nwin, w, lane);
    }
    // rightmost smallest of the first min(n, w - 1) entries (all n entries for a short read)
    const int lim = nwin > 0 ? w - 1 : n;
    uint64_t bh = ~0ULL;
    int bi = -1;
    for (int v = lane; v < lim; v += 64) {
      const uint64_t x = H[v];
      if (x <= bh) bh = x, bi = v;  // ascending v per lane: <= keeps the rightmost
    }
    for (int d = 32; d; d >>= 1) {
      const uint64_t oh = (uint64_t)__shfl_xor((int)(bh >> 32), d, 64) << 32 | (uint32_t)__shfl_xor((int)bh, d, 64);
      const int oi = __shfl_xor(bi, d, 64);
      if (oi >= 0 && (bi < 0 || oh < bh || (oh == bh && oi > bi))) bh = oh, bi = oi;
    }
    const int m_idx = bi;
    const uint64_t m_h = bh;
    const uint64_t h_last = nwin > 0 ? H[w - 1] : 0;
    __syncthreads();
    // ---- pass 3 ----------------------------------------------------------------------------------------------
    const uint64_t cap = slab_off[slot + 1] - slab_off[slot];
    pgx_mm128 *dst = slab + slab_off[slot];
    uint32_t nout = 0;
    bool over = false;
    for (int p0 = 0; p0 < n; p0 += 64) {
      const int p = p0 + lane;
      bool emit = false;
      uint64_t hp = 0;
      if (p < n) {
        hp = H[p];
        if (nwin > 0) {
          const int s0 = p - w + 1 > 0 ? p - w + 1 : 0, s1 = p < nwin - 1 ? p : nwin - 1;
          uint64_t gx = 0;
          if (s1 - s0 + 1 == w) {  // all w windows exist: tail of block(s0) + head of the next block
            gx = (s0 % w == 0) ? T1[s1] : max(T2[s0], T1[s1]);
          } else {  // the first / last w-1 entries of the read: fewer windows, scanned directly
            for (int sidx = s0; sidx <= s1; ++sidx) gx = max(gx, WM[sidx]);
          }
          emit = gx == hp;  // (a window that contains p has minimum <= hash(p))
          if (p <= w - 2 && hp == m_h) emit = p != m_idx ? true : h_last > m_h;
        } else {
          emit = p == m_idx;
        }
      }
      const uint64_t em = __ballot(emit);
      if (emit) {
        const uint32_t r = nout + (uint32_t)__builtin_popcountll(em & ((1ULL << lane) - 1));
        if (r < cap) dst[r] = pgx_mm128{hp << 8 | (uint64_t)k, (uint64_t)rd.rid << 32 | PY[p]};
        else over = true;
      }
      nout += (uint32_t)__builtin_popcountll(em);
    }
    const bool anyover = __ballot(over) != 0;
    if (lane == 0) {
      counts[slot] = anyover ? 0u : nout;
      if (anyover) flags[slot] = 1;
    }
    __syncthreads();
  }
}

// k_sketch_general over the listed reads (d_list == nullptr: slots 0 .. lens.size()-1); lens[i] = length of the i-th listed read.
// Entry scratch (hash, position|strand, window minimum, two running-extremum arrays: 36 B per base) in batches of at most
// ~256 Mbases.
void launch_sketch_general(const pgx_seqdb *db, const ReadDesc *d_reads, const std::vector<uint32_t> &lens, const uint32_t *d_list,
                           int w, int k, pgx_mm128 *d_slab, const uint64_t *d_slab_off, uint32_t *d_counts, uint32_t *d_flags) {
  hipStream_t st = ctx().stream;
  const uint64_t batch_bases = 256ull << 20;
  DevBuf<uint32_t> iota;
  if (!d_list) {   // the kernel walks a list: the identity
    std::vector<uint32_t> id(lens.size());
    for (size_t i = 0; i < id.size(); ++i) id[i] = (uint32_t)i;
    iota.alloc(id.size());
    iota.upload(id.data(), id.size());
    sync();
    d_list = iota.p;
  }
  for (size_t b0 = 0; b0 < lens.size();) {
    std::vector<uint64_t> so;
    uint64_t acc = 0;
    size_t b1 = b0;
    while (b1 < lens.size() && (b1 == b0 || acc + lens[b1] <= batch_bases)) so.push_back(acc), acc += lens[b1], ++b1;
    uint64_t *d_so = ws<uint64_t>("sk.gen_off", so.size());
    uint64_t *H = ws<uint64_t>("sk.gen_h", acc), *WMv = ws<uint64_t>("sk.gen_wm", acc);
    uint64_t *T1 = ws<uint64_t>("sk.gen_t1", acc), *T2 = ws<uint64_t>("sk.gen_t2", acc);
    uint32_t *PY = ws<uint32_t>("sk.gen_py", acc);
    PGX_HIP(hipMemcpyAsync(d_so, so.data(), so.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)std::min<size_t>(b1 - b0, (size_t)ctx().num_cu * 16);
    PGX_REQUIRE(db->d_seq.p, PGX_ESTATE, "the seqdb's bytes were released (pgx_seqdb_release_bytes): the general sketch kernel (other w / k, L0 output) needs them");
    hipLaunchKernelGGL(k_sketch_general, dim3(grid), dim3(64), 0, st, db->d_seq.p, d_reads, d_list + b0, (uint32_t)(b1 - b0), w, k,
                       d_so, H, PY, WMv, T1, T2, d_slab, d_slab_off, d_counts, d_flags);
    PGX_HIP(hipGetLastError());
    sync();  // (so[] is reused by the next batch)
    b0 = b1;
  }
}

void dev_sketch_nreads(const pgx_seqdb *db, const ReadDesc *d_reads, const uint32_t *d_list, uint32_t nn, int w, int k,
                       DevBuf<pgx_mm128> &nl0, DevBuf<uint64_t> &nl0_off, uint64_t *n_total);   // pgx_sketch_n.hip
void dev_reduce_nreads(const DevBuf<pgx_mm128> &nl0, const DevBuf<uint64_t> &nl0_off, uint32_t nn, uint64_t total, int rs, int levels,
                       DevBuf<pgx_mm128> &top, DevBuf<uint32_t> &cnt);
void dev_scatter_counts(uint32_t *d_counts_by_slot, const uint32_t *d_list, uint32_t nn, const uint64_t *d_off, const uint32_t *d_cnt);
void dev_mark_slots(uint32_t *d_by_slot, const uint32_t *d_list, uint32_t nn, uint32_t v);

// Level-0 minimizers of `reads`, one contiguous list in `reads` order.  Every read goes through the closed-form kernel of the
// (w, k) -- k_sketch_wave for k = 16 and w in {64, 80, 96, 128}, k_sketch_general otherwise -- single pass into per-read slabs, then
// an ordered gather; the reads it flags (an ambiguous base, a slab outgrown by low-complexity sequence) are cut into runs of
// unambiguous bases that the same kernels sketch (pgx_sketch_n.hip: dev_sketch_nreads).  n_literal (name kept from the C-ABI's
// reads_literal): how many reads took that second path.
void dev_sketch(const pgx_seqdb *db, const std::vector<ReadDesc> &reads, int w, int k, DevBuf<pgx_mm128> &out,
                size_t &n_out, uint32_t *n_literal) {
  n_out = 0;
  if (n_literal) *n_literal = 0;
  const uint32_t n = (uint32_t)reads.size();
  if (n == 0) { out.alloc(0); return; }
  hipStream_t st = ctx().stream;
  DevBuf<ReadDesc> d_reads(n);
  d_reads.upload(reads.data(), n);
  DevBuf<uint32_t> counts(n), d_flag(n);
  DevBuf<uint64_t> offs(n + 1);
  PGX_HIP(hipMemsetAsync(counts.p, 0, n * sizeof(uint32_t), st));
  PGX_HIP(hipMemsetAsync(d_flag.p, 0, n * sizeof(uint32_t), st));

  std::vector<uint64_t> slab_off(n + 1, 0);
  std::vector<uint32_t> lens(n);
  uint64_t bases = 0;
  // (w, k) outside the specialised kernel's set: the general closed-form kernel takes the wave kernel's place
  const bool general = !(k == 16 && (w == 64 || w == 80 || w == 96 || w == 128));
  for (uint32_t i = 0; i < n; ++i) {
    PGX_REQUIRE(reads[i].len < (1u << 30), PGX_EARG, "read %u is longer than 2^30 bases", reads[i].rid);
    // slab capacity: 5x the expected density 2/(w+1); a read that outgrows it (low complexity) takes the second path
    slab_off[i + 1] = slab_off[i] + (uint64_t)reads[i].len / 8 + 64;
    lens[i] = reads[i].len, bases += reads[i].len;
  }
  DevBuf<uint64_t> d_slab_off(n + 1);
  d_slab_off.upload(slab_off.data(), n + 1);
  DevBuf<pgx_mm128> slab(slab_off[n]);
  if (!general) {
    KernelTimer tm("sketch", bases);
    launch_sketch_wave(db, d_reads.p, nullptr, n, w, k, slab.p, d_slab_off.p, counts.p, d_flag.p);
  } else {
    KernelTimer tm("sketch_general", bases);
    launch_sketch_general(db, d_reads.p, lens, nullptr, w, k, slab.p, d_slab_off.p, counts.p, d_flag.p);
  }
  std::vector<uint32_t> flag(n);
  d_flag.download(flag.data(), n);
  sync();
  std::vector<uint32_t> second;
  for (uint32_t i = 0; i < n; ++i)
    if (flag[i]) second.push_back(i);
  // the flagged reads: segments of unambiguous bases through the same kernels, exact slabs on demand
  DevBuf<uint32_t> d_second(second.size()), d_skip;
  DevBuf<pgx_mm128> nl0;
  DevBuf<uint64_t> nl0_off;
  if (!second.empty()) {
    KernelTimer tm("sketch_nreads", 0);
    d_second.upload(second.data(), second.size());
    uint64_t tot2 = 0;
    dev_sketch_nreads(db, d_reads.p, d_second.p, (uint32_t)second.size(), w, k, nl0, nl0_off, &tot2);
    dev_scatter_counts(counts.p, d_second.p, (uint32_t)second.size(), nl0_off.p, nullptr);
    d_skip.alloc(n);
    PGX_HIP(hipMemsetAsync(d_skip.p, 0, n * sizeof(uint32_t), st));
    dev_mark_slots(d_skip.p, d_second.p, (uint32_t)second.size(), 1u);
  }
  if (n_literal) *n_literal = (uint32_t)second.size();
  {
    CubTemp tmp;
    size_t bytes = 0;
    PGX_HIP(hipMemsetAsync(offs.p, 0, sizeof(uint64_t), st));
    PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, counts.p, offs.p + 1, (int)n, st));
    PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp.get(bytes), bytes, counts.p, offs.p + 1, (int)n, st));
    uint64_t total = 0;
    PGX_HIP(hipMemcpyAsync(&total, offs.p + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    sync();
    n_out = (size_t)total;
  }
  out.alloc(n_out);
  if (n_out == 0) return;
  {
    KernelTimer tm("sketch_gather", bases);
    hipLaunchKernelGGL(k_gather_slabs, dim3(n), dim3(64), 0, st, slab.p, d_slab_off.p, (const uint32_t *)nullptr, n, counts.p, offs.p, out.p,
                       (const uint32_t *)d_skip.p, 0);
    if (!second.empty())
      hipLaunchKernelGGL(k_gather_slabs, dim3((unsigned)second.size()), dim3(64), 0, st, nl0.p, nl0_off.p, (const uint32_t *)d_second.p,
                         (uint32_t)second.size(), counts.p, offs.p, out.p, (const uint32_t *)nullptr, 1);
  }
  sync();
}

// =========================================================================================================
// k_reduce_read: mm_reduce (src/shmr_reduce.c:53-90) applied `levels` times to ONE read's minimizers, in LDS, in place
// in the read's slab.  Same restatement as k_reduce_flag (winner = smallest x>>8, ties to the lowest ring slot
// offset % rs; emitted iff its y differs from the previous window's winner; the first window always emits).
// =========================================================================================================
constexpr int RMAX = 1024;  // minimizers per read handled in LDS (a 15 kb read has ~375 at w = 80)

__global__ __launch_bounds__(64) void k_reduce_read(pgx_mm128 *__restrict__ slab, const uint64_t *__restrict__ slab_off,
                                                    const ReadDesc *__restrict__ reads,
                                                    const uint32_t *__restrict__ counts0,
                                                    const uint32_t *__restrict__ flags, uint32_t n, int rs, int levels,
                                                    uint32_t *__restrict__ counts_top, uint32_t *__restrict__ nbad) {
  __shared__ uint64_t sx[RMAX];
  __shared__ uint32_t sy[RMAX];
  const int lane = threadIdx.x;
  const uint32_t slot = blockIdx.x;
  if (slot >= n) return;
  const uint32_t c0 = counts0[slot];
  if (flags[slot] || c0 > (uint32_t)RMAX) {
    if (lane == 0) atomicAdd(nbad, 1u), counts_top[slot] = 0;
    return;
  }
  pgx_mm128 *p = slab + slab_off[slot];
  for (uint32_t i = lane; i < c0; i += 64) {
    const pgx_mm128 e = p[i];
    sx[i] = e.x, sy[i] = (uint32_t)e.y;
  }
  int ncur = (int)c0;
  for (int lv = 0; lv < levels; ++lv) {
    __syncthreads();
    uint64_t wx[RMAX / 64];
    uint32_t wy[RMAX / 64];
    uint32_t emit = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int r = 0; r < RMAX / 64; ++r) {
      const int t = lane + 64 * r;
      const bool valid = t < ncur && t >= rs - 1;
      uint64_t bx = 0;
      uint32_t by = 0;
      if (valid) {
        int u = t - rs + 1, sl = (t + 1) % rs;  // slot of element u is u % rs (offset within the read)
        bx = sx[u], by = sy[u];
        uint64_t bh = bx >> 8;
        int bsl = sl;
        for (int j = 1; j < rs; ++j) {
          ++u;
          if (++sl == rs) sl = 0;
          const uint64_t x = sx[u], hsh = x >> 8;
          if (hsh < bh || (hsh == bh && sl < bsl)) bx = x, by = sy[u], bh = hsh, bsl = sl;
        }
      }
      uint32_t prevy = (uint32_t)__shfl_up((int)by, 1, 64);
      if (lane == 0) prevy = carry;
      carry = (uint32_t)__builtin_amdgcn_readlane((int)by, 63);
      if (valid && (t == rs - 1 || by != prevy)) emit |= 1u << r;
      wx[r] = bx, wy[r] = by;
    }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int r = 0; r < RMAX / 64; ++r) {
      const bool em = (emit >> r) & 1u;
      const uint64_t m = __ballot(em);
      if (em) {
        const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        sx[idx] = wx[r], sy[idx] = wy[r];
      }
      base += __builtin_popcountll(m);
    }
    ncur = base;
  }
  __syncthreads();
  const uint64_t yhi = (uint64_t)reads[slot].rid << 32;
  for (int i = lane; i < ncur; i += 64) p[i] = pgx_mm128{sx[i], yhi | sy[i]};
  if (lane == 0) counts_top[slot] = (uint32_t)ncur;
}

__global__ void k_set_aside_ambiguous(uint32_t *__restrict__ flags, uint32_t n, uint32_t *__restrict__ mark) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = flags[i];
  mark[i] = (f & 2u) ? 1u : 0u;   // k_sketch_blk's "ambiguous base" bit
  if (f & 2u) flags[i] = 0;
}
__global__ void k_restore_marks(uint32_t *__restrict__ flags, uint32_t n, const uint32_t *__restrict__ mark) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && mark[i]) flags[i] |= 2u;
}
__global__ void k_count_flags(const uint32_t *__restrict__ flags, uint32_t n, uint32_t *__restrict__ nbad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) atomicAdd(nbad, 1u);
}

static double trace_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the fused index path's plan cache: which read selection's descriptors and slab offsets the "ix.reads" / "ix.slab_off"
// workspaces hold.  pgx_shutdown() frees the workspaces, so it must forget the plan too (a pgx_seqdb that outlives a
// shutdown + init would otherwise run the sketch kernels on uninitialised descriptors).
namespace {
struct FusedPlan {
  uint64_t dev_serial = 0, slab_total = 0, plan_bases = 0, last_use = 0;
  int plan_w = 0, plan_k = 0;
  uint64_t plan_div = 0;
  bool plan_ok = false;
  DevBuf<ReadDesc> d_reads;       // the selection's descriptors and slab offsets stay on the device with their plan (one pair per selection:
  DevBuf<uint64_t> d_slab_off;    // a pipeline that cycles through a job's chunks uploads each once)
};
std::vector<FusedPlan> g_plans;   // at most 32, least recently used replaced
uint64_t g_plan_clock = 0;
ShutdownHook g_plan_reset([] { g_plans.clear(); });
}  // namespace

bool dev_index_fused(const pgx_seqdb *db, const std::vector<ReadDesc> &reads, int w, int k, int rs, int levels,
                     const pgx_mm128 **d_top, size_t *n_top, uint64_t plan_serial, uint32_t *n_second) {
  if (n_second) *n_second = 0;
  const uint32_t n = (uint32_t)reads.size();
  if (n == 0 || levels < 1 || levels > 2 || rs < 1) return false;
  const bool trace = getenv("PGX_TRACE") != nullptr;
  const double tr0 = trace ? trace_ms() : 0;
  // slab offsets and read descriptors: computed and uploaded once per plan (file-scope state above: reset by pgx_shutdown)
  FusedPlan *fp = nullptr;
  if (plan_serial)
    for (auto &pl : g_plans)
      if (pl.dev_serial == plan_serial) fp = &pl;
  if (!fp)   // (a selection without a serial, or a new one: an entry that holds no plan first)
    for (auto &pl : g_plans)
      if (pl.dev_serial == 0) fp = &pl;
  if (!fp) {
    if (g_plans.size() < 32) {
      g_plans.emplace_back();
      fp = &g_plans.back();
    } else {
      fp = &g_plans[0];
      for (auto &pl : g_plans)
        if (pl.last_use < fp->last_use) fp = &pl;
      sync();   // (its buffers may still be read by what the last stage enqueued)
      *fp = FusedPlan();
    }
  }
  FusedPlan &g_plan = *fp;
  g_plan.last_use = ++g_plan_clock;
  uint64_t &dev_serial = g_plan.dev_serial, &slab_total = g_plan.slab_total, &plan_bases = g_plan.plan_bases;
  int &plan_w = g_plan.plan_w, &plan_k = g_plan.plan_k;
  bool &plan_ok = g_plan.plan_ok;
  static const char *mode_env = getenv("PGX_SKETCH");
  static const bool want_fuse = (getenv("PGX_FUSE") && atoi(getenv("PGX_FUSE")) != 0) || (mode_env && !strcmp(mode_env, "fuse"));
  static const bool want_wave = mode_env && !strcmp(mode_env, "wave");
  // slab of a read: len / slab_div + 64 elements.  The fused kernels only ever write the TOP-level list there (1 element per 408 bases at
  // l = 2, per 142 at l = 1: L0 never leaves the CU), so their slabs are len / 48 resp. len / 24 -- six and three times the expected list; a
  // read that outgrows its slab (low-complexity sequence) is flagged and redone into an exact one below, as ever.  Round 1-4 reserved
  // len / 8 for every path: 27 GB of workspace for one index chunk of full-size configs[3], resident through the overlap stages too (round 5:
  // the HBM ledger).  The unfused path (k_sketch_wave + k_reduce_read: L0 goes through the slab) keeps len / 8.
  // (PGX_SLAB_DIV / PGX_SLAB_MIN: test knobs that make reads outgrow their slabs)
  const bool fused_out = !want_wave && (want_fuse ? sketch_fused_supported(w, rs, levels) : sketch_blk_supported(w, k, rs, levels));
  const uint64_t slab_div = getenv("PGX_SLAB_DIV") ? std::max(1ll, atoll(getenv("PGX_SLAB_DIV"))) : !fused_out ? 8 : levels >= 2 ? 48 : 24;
  const bool cached = plan_serial != 0 && plan_serial == dev_serial && plan_w == w && plan_k == k && g_plan.plan_div == slab_div;
  hipStream_t st = ctx().stream;
  if (!cached) {
    MemTag plan_tag("index.plans");
    g_plan.d_reads.alloc(n), g_plan.d_slab_off.alloc((size_t)n + 1);
  }
  ReadDesc *d_reads = g_plan.d_reads.p;
  uint64_t *d_slab_off = g_plan.d_slab_off.p;
  if (!cached) {
    dev_serial = 0;
    std::vector<uint64_t> slab_off(n + 1, 0);
    uint64_t bases = 0;
    plan_ok = true;
    const uint64_t slab_min = getenv("PGX_SLAB_MIN") ? std::max(1ll, atoll(getenv("PGX_SLAB_MIN"))) : 64;
    for (uint32_t i = 0; i < n; ++i) {
      if (!sketch_wave_eligible(reads[i], w, k)) plan_ok = false;
      slab_off[i + 1] = slab_off[i] + (uint64_t)reads[i].len / slab_div + slab_min;
      bases += reads[i].len;
    }
    if (plan_ok) {
      PGX_HIP(hipMemcpyAsync(d_reads, reads.data(), n * sizeof(ReadDesc), hipMemcpyHostToDevice, st));
      PGX_HIP(hipMemcpyAsync(d_slab_off, slab_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
      sync();  // (slab_off is a local)
    }
    slab_total = slab_off[n], plan_bases = bases, plan_w = w, plan_k = k, g_plan.plan_div = slab_div;
    dev_serial = plan_serial;
  }
  if (!plan_ok) return false;
  const uint64_t bases = plan_bases;
  uint32_t *d_cnt = ws<uint32_t>("ix.cnt", 3 * (size_t)n + 4);  // [counts0 | flags | counts_top | nbad]
  uint32_t *d_flags = d_cnt + n, *d_ctop = d_cnt + 2 * (size_t)n, *d_nbad = d_cnt + 3 * (size_t)n;
  uint64_t *d_offs = ws<uint64_t>("ix.offs", n + 1);
  pgx_mm128 *slab = ws<pgx_mm128>("ix.slab", slab_total);
  PGX_HIP(hipMemsetAsync(d_cnt, 0, (3 * (size_t)n + 4) * sizeof(uint32_t), st));
  if (trace) {
    sync();
    fprintf(stderr, "[pgx] index: plan (slab offsets, descriptors, workspaces, uploads) %.2f ms\n", trace_ms() - tr0);
  }
  // Default (round 2): k_sketch_blk -- block-per-lane closed form fused with the streaming reduce, L0 never leaves the CU, HBM
  // traffic == the algorithmic 1.04 B/base -- and k_sketch_wave (fused form) for the reads it flags (two drops close together,
  // bursts of ties, very short reads).  PGX_SKETCH=wave: k_sketch_wave + k_reduce_read (round 1's default); PGX_SKETCH=fuse (or
  // PGX_FUSE=1): k_sketch_wave in its fused form for every read.
  uint32_t n_redo2 = 0;               // reads redone into exact-size slabs (slab2, offsets by list position)
  pgx_mm128 *slab2 = nullptr;
  uint64_t *d_off2_keep = nullptr;
  uint32_t *d_list2_keep = nullptr, *d_in2 = nullptr;
  if (!want_fuse && !want_wave && sketch_blk_supported(w, k, rs, levels)) {
    {
      KernelTimer tm("sketch", bases);
      launch_sketch_blk(db, d_reads, n, rs, levels, slab, d_slab_off, d_ctop, d_flags);
    }
    // reads with an ambiguous base (flag bit 2) skip the two redo passes below -- the fused wave kernel would walk them whole only to flag
    // them again -- and go straight to the run-by-run path at the end (their mark is put back once the passes are through)
    uint32_t *d_nmark = ws<uint32_t>("ix.nmark", n);
    hipLaunchKernelGGL(k_set_aside_ambiguous, dim3(cdiv(n, 256)), dim3(256), 0, st, d_flags, n, d_nmark);
    // the flagged reads, once more on the general closed-form kernel
    uint32_t *d_list = ws<uint32_t>("ix.redo", (size_t)n + 1);
    size_t sbytes = 0;
    hipcub::CountingInputIterator<uint32_t, ptrdiff_t> iota(0);
    PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, sbytes, iota, d_flags, d_list, d_list + n, (int)n, st));
    void *stmp = ws_raw("ix.sel_tmp", sbytes);
    PGX_HIP(hipcub::DeviceSelect::Flagged(stmp, sbytes, iota, d_flags, d_list, d_list + n, (int)n, st));
    uint32_t nredo = 0;
    PGX_HIP(hipMemcpyAsync(&nredo, d_list + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sync();
    if (getenv("PGX_TRACE")) {
      std::vector<uint32_t> hf(n);
      PGX_HIP(hipMemcpy(hf.data(), d_flags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
      unsigned why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (uint32_t f : hf)
        for (int b = 0; b < 8; ++b) why[b] += (f >> b) & 1u;
      fprintf(stderr, "[pgx] index: %u of %u reads redone by the general closed-form kernel (short %u, ambiguous base %u, two drops in a tile %u, "
              "drop in the first window %u, close drops %u, tie burst %u / %u, slab %u)\n", nredo, n, why[0], why[1], why[2], why[3], why[4],
              why[5], why[6], why[7]);
    }
    if (nredo) {
      KernelTimer tm("sketch_redo", 0);
      uint32_t *d_need = ws<uint32_t>("ix.need", n);
      PGX_HIP(hipMemsetAsync(d_flags, 0, (size_t)n * sizeof(uint32_t), st));
      launch_sketch_fused_list(db, d_reads, d_list, nredo, rs, levels, slab, d_slab_off, d_ctop, d_flags, d_need, 0);
      // Low-complexity reads (a homopolymer or a short-period tandem array makes every position a tied minimum, on every level)
      // can outgrow their slab; the kernel reports how many elements each such read has, and a second launch redoes exactly
      // those reads into slabs of exactly that size.  (Round 1 redid the WHOLE chunk on the slow general path when a single
      // read was left over: 0.4 s instead of 15 ms at 9 Gbases with 1 % low-complexity sequence.)
      uint32_t *d_list2 = ws<uint32_t>("ix.redo2", (size_t)n + 1);
      PGX_HIP(hipcub::DeviceSelect::Flagged(stmp, sbytes, iota, d_flags, d_list2, d_list2 + n, (int)n, st));
      PGX_HIP(hipMemcpyAsync(&n_redo2, d_list2 + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      sync();
      if (n_redo2) {
        uint64_t *d_off2 = ws<uint64_t>("ix.off2", (size_t)n_redo2 + 1);
        hipLaunchKernelGGL(k_need_of_list, dim3(cdiv(n_redo2, 256)), dim3(256), 0, st, d_need, d_list2, n_redo2, d_off2 + 1);
        PGX_HIP(hipMemsetAsync(d_off2, 0, sizeof(uint64_t), st));
        size_t b2 = 0;
        PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, b2, d_off2 + 1, d_off2 + 1, (int)n_redo2, st));
        void *t2 = ws_raw("ix.scan2_tmp", b2);
        PGX_HIP(hipcub::DeviceScan::InclusiveSum(t2, b2, d_off2 + 1, d_off2 + 1, (int)n_redo2, st));
        uint64_t total2 = 0;
        PGX_HIP(hipMemcpyAsync(&total2, d_off2 + n_redo2, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        sync();
        slab2 = ws<pgx_mm128>("ix.slab2", std::max<uint64_t>(total2, 1));
        d_off2_keep = d_off2, d_list2_keep = d_list2;
        d_in2 = ws<uint32_t>("ix.in2", n);
        PGX_HIP(hipMemcpyAsync(d_in2, d_flags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));   // (1 for the reads of list 2)
        PGX_HIP(hipMemsetAsync(d_flags, 0, (size_t)n * sizeof(uint32_t), st));
        launch_sketch_fused_list(db, d_reads, d_list2, n_redo2, rs, levels, slab2, d_off2, d_ctop, d_flags, nullptr, 1);
        if (trace) fprintf(stderr, "[pgx] index: %u reads outgrew their slabs and were redone into exact ones (%llu elements)\n", n_redo2, (unsigned long long)total2);
      }
    }
    hipLaunchKernelGGL(k_restore_marks, dim3(cdiv(n, 256)), dim3(256), 0, st, d_flags, n, d_nmark);
    hipLaunchKernelGGL(k_count_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, d_flags, n, d_nbad);
  } else if (want_fuse && sketch_fused_supported(w, rs, levels)) {
    KernelTimer tm("sketch", bases);
    launch_sketch_fused(db, d_reads, n, rs, levels, slab, d_slab_off, d_ctop, d_flags);
    hipLaunchKernelGGL(k_count_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, d_flags, n, d_nbad);
  } else {
    {
      KernelTimer tm("sketch", bases);
      launch_sketch_wave(db, d_reads, nullptr, n, w, k, slab, d_slab_off, d_cnt, d_flags);
    }
    KernelTimer tm("reduce", bases);
    hipLaunchKernelGGL(k_reduce_read, dim3(n), dim3(64), 0, st, slab, d_slab_off, d_reads, d_cnt, d_flags, n, rs, levels,
                       d_ctop, d_nbad);
  }
  // Reads still flagged here hold an ambiguous base (mm_sketch.c:112-113): they are cut into runs of unambiguous bases, every run
  // sketched by the unfused closed-form kernel, the read's list assembled and reduced per read (pgx_sketch_n.hip)
  DevBuf<pgx_mm128> n_l0, n_toplist;
  DevBuf<uint64_t> n_off;
  DevBuf<uint32_t> n_cnt;
  uint32_t n3 = 0;
  uint32_t *d_list3 = nullptr;
  uint32_t *d_skip = d_in2, *d_skip3 = nullptr;
  {
    uint32_t nb = 0;
    PGX_HIP(hipMemcpyAsync(&nb, d_nbad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sync();
    if (nb && !want_wave) {
      KernelTimer tm("sketch_nreads", 0);
      d_list3 = ws<uint32_t>("ix.redo3", (size_t)n + 1);
      size_t sb = 0;
      hipcub::CountingInputIterator<uint32_t, ptrdiff_t> iota(0);
      PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, sb, iota, d_flags, d_list3, d_list3 + n, (int)n, st));
      void *stmp = ws_raw("ix.sel_tmp", sb);
      PGX_HIP(hipcub::DeviceSelect::Flagged(stmp, sb, iota, d_flags, d_list3, d_list3 + n, (int)n, st));
      PGX_HIP(hipMemcpyAsync(&n3, d_list3 + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      sync();
      uint64_t tot3 = 0;
      dev_sketch_nreads(db, d_reads, d_list3, n3, w, k, n_l0, n_off, &tot3);
      dev_reduce_nreads(n_l0, n_off, n3, tot3, rs, levels, n_toplist, n_cnt);
      dev_scatter_counts(d_ctop, d_list3, n3, nullptr, n_cnt.p);
      if (!d_skip) {
        d_skip = ws<uint32_t>("ix.in2", n);
        PGX_HIP(hipMemsetAsync(d_skip, 0, (size_t)n * sizeof(uint32_t), st));
      }
      dev_mark_slots(d_skip, d_list3, n3, 1u);
      d_skip3 = ws<uint32_t>("ix.in3", n);   // (the exact-slab gather below must step over them too: a read can be in both lists)
      PGX_HIP(hipMemsetAsync(d_skip3, 0, (size_t)n * sizeof(uint32_t), st));
      dev_mark_slots(d_skip3, d_list3, n3, 1u);
      dev_mark_slots(d_flags, d_list3, n3, 0u);
      PGX_HIP(hipMemsetAsync(d_nbad, 0, sizeof(uint32_t), st));
      if (n_second) *n_second = n3;
      if (trace) fprintf(stderr, "[pgx] index: %u reads with ambiguous bases sketched run by run (%llu level-0 minimizers)\n", n3, (unsigned long long)tot3);
    }
  }
  size_t bytes = 0;
  PGX_HIP(hipMemsetAsync(d_offs, 0, sizeof(uint64_t), st));
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, d_ctop, d_offs + 1, (int)n, st));
  void *tmp = ws_raw("ix.scan_tmp", bytes);
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp, bytes, d_ctop, d_offs + 1, (int)n, st));
  uint64_t total = 0;
  uint32_t nbad = 0;
  PGX_HIP(hipMemcpyAsync(&total, d_offs + n, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  PGX_HIP(hipMemcpyAsync(&nbad, d_nbad, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  sync();
  if (trace) fprintf(stderr, "[pgx] index: sketch + reduce done at +%.2f ms\n", trace_ms() - tr0);
  if (trace && nbad) {
    std::vector<uint32_t> hf(n);
    PGX_HIP(hipMemcpy(hf.data(), d_flags, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    unsigned why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t f : hf)
      for (int b = 0; b < 8; ++b) why[b] += (f >> b) & 1u;
    fprintf(stderr, "[pgx] index: %u reads still flagged after the general closed-form kernel (bits 0..7: %u %u %u %u %u %u %u %u)\n", nbad, why[0], why[1], why[2],
            why[3], why[4], why[5], why[6], why[7]);
  }
  if (nbad) return false;  // some read needs the general path; the caller redoes the chunk there
  pgx_mm128 *top = ws<pgx_mm128>("ix.top", total);
  if (total) {
    KernelTimer tm("sketch_gather", bases);
    hipLaunchKernelGGL(k_gather_slabs, dim3(n), dim3(64), 0, st, slab, d_slab_off, (const uint32_t *)nullptr, n, d_ctop, d_offs,
                       top, (const uint32_t *)d_skip, 0);
    if (n_redo2)
      hipLaunchKernelGGL(k_gather_slabs, dim3(n_redo2), dim3(64), 0, st, slab2, d_off2_keep, (const uint32_t *)d_list2_keep, n_redo2, d_ctop,
                         d_offs, top, (const uint32_t *)d_skip3, 1);
    if (n3)
      hipLaunchKernelGGL(k_gather_slabs, dim3(n3), dim3(64), 0, st, n_toplist.p, n_off.p, (const uint32_t *)d_list3, n3, d_ctop, d_offs, top,
                         (const uint32_t *)nullptr, 1);
    if (n3) sync();   // (n_toplist goes back to the block cache when this function returns)
  }
  *d_top = top;
  *n_top = (size_t)total;
  return true;
}

}  // namespace pgx
