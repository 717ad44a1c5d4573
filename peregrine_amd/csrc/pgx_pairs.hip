// pgx_pairs.hip -- shimmer-pair candidate join on the GPU: what build_map (/root/reference/src/shmr_utils.c:295-404)
// computes, as data-parallel kernels + device radix sorts + segmented reductions.
//
//   counts   : all MC entries  -> sort by mer + reduce-by-key            (aggregate_mm_count, shmr_utils.c:162-176)
//   keep     : lower <= count(x>>8) <= upper per shimmer (binary search in the aggregated table); the scan starts at
//              the first shimmer with lower <= count < upper (STRICT, :311-320)
//   chain    : previous kept shimmer by an inclusive max-scan; a kept shimmer pairs with it iff same read and position
//              gap >= 100 (:329-336); the anchor always advances to the kept shimmer (:402)
//   records  : forward record [a.x][b.x] if (a.x>>8)%T == c%T, reverse record [b.x][a.x] with mirrored coordinates if
//              (b.x>>8)%T == c%T (:337-400); record index == insertion order ("seq")
//   buckets  : stable LSD radix sorts (position descending, key1, key0) give every (key0,key1) bucket contiguous and
//              already in the order the reference's qsort produces (descending position, ties in insertion order,
//              shmr_overlap.c:46-50,217); segmented min/max of seq give the first/last insertion of every bucket and
//              key0 group, which is all the host needs to replay klib-khash's slot order on DISTINCT keys only.
#include <chrono>

#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

namespace {
struct Tmp {
  DevBuf<uint8_t> buf;
  void *get(size_t bytes) {
    if (bytes > buf.n) buf.alloc(bytes + (bytes >> 2) + 256);
    return buf.p;
  }
};
using CountIt = hipcub::CountingInputIterator<uint32_t, ptrdiff_t>;

__global__ void k_split_counts(const pgx_mm_count *__restrict__ in, size_t n, uint64_t *__restrict__ mer,
                               uint32_t *__restrict__ cnt) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) mer[t] = in[t].mer, cnt[t] = in[t].count;
}

__device__ __forceinline__ bool lookup_count(const uint64_t *mer, const uint32_t *cnt, uint32_t nu, uint64_t key,
                                             uint32_t *out) {
  uint32_t lo = 0, hi = nu;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (mer[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  if (lo < nu && mer[lo] == key) {
    *out = cnt[lo];
    return true;
  }
  return false;
}

// keep[i] = lower <= count <= upper ; first_strict = min i with lower <= count < upper ; missing = hash not in the table
__global__ void k_keep(const pgx_mm128 *__restrict__ mm, uint32_t n, const uint64_t *__restrict__ mer,
                       const uint32_t *__restrict__ cnt, uint32_t nu, uint32_t lower, uint32_t upper,
                       uint8_t *__restrict__ keep, uint32_t *__restrict__ first_strict, uint32_t *__restrict__ missing) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t c = 0;
  if (!lookup_count(mer, cnt, nu, mm[i].x >> 8, &c)) {
    atomicAdd(missing, 1u);
    keep[i] = 0;
    return;
  }
  keep[i] = (c >= lower && c <= upper);
  if (c >= lower && c < upper) atomicMin(first_strict, i);
}

__global__ void k_chain_in(const uint8_t *__restrict__ keep, uint32_t n, const uint32_t *__restrict__ first_strict,
                           int32_t *__restrict__ v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  v[i] = (keep[i] && i >= *first_strict) ? (int32_t)i : -1;
}

struct MaxOp {
  __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; }
};

__device__ __forceinline__ uint32_t pos28(uint64_t y) { return (uint32_t)((y >> 1) & 0xFFFFFFF); }
__device__ __forceinline__ uint32_t pos_of(uint64_t y) { return (uint32_t)((y & 0xFFFFFFFFu) >> 1); }
__device__ __forceinline__ uint64_t flip_y(uint64_t y, uint64_t x, const uint32_t *rlen) {
  const uint32_t span = (uint32_t)(x & 0xFF), rid = (uint32_t)(y >> 32);
  const uint32_t rpos = rlen[rid] - (pos_of(y) + 1) + span - 1;  // shmr_utils.c:378-385
  return ((y & 0xFFFFFFFF00000001ULL) | (uint64_t)(rpos << 1)) ^ 1ULL;
}

// MODE 0: number of records shimmer i produces (0..2) ; MODE 1: write them at off[i]
template <int MODE>
__global__ void k_records(const pgx_mm128 *__restrict__ mm, uint32_t n, const int32_t *__restrict__ chain,
                          uint32_t T, uint32_t c, const uint32_t *__restrict__ rlen, uint32_t *__restrict__ nrec,
                          const uint32_t *__restrict__ off, uint64_t *__restrict__ key0, uint64_t *__restrict__ key1,
                          uint64_t *__restrict__ y0, uint8_t *__restrict__ dir, uint32_t *__restrict__ npos,
                          uint64_t *__restrict__ y1) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t cnt = 0;
  bool fwd = false, rev = false;
  pgx_mm128 a{0, 0}, b{0, 0};
  if (i > 0 && chain[i] == (int32_t)i) {  // i is kept
    const int32_t p = chain[i - 1];       // previous kept shimmer
    if (p >= 0) {
      a = mm[p], b = mm[i];
      if ((a.y >> 32) == (b.y >> 32) && pos28(b.y) - pos28(a.y) >= 100u) {
        fwd = (a.x >> 8) % T == c;
        rev = (b.x >> 8) % T == c;
        cnt = (fwd ? 1u : 0u) + (rev ? 1u : 0u);
      }
    }
  }
  if (MODE == 0) {
    nrec[i] = cnt;
  } else if (cnt) {
    uint32_t o = off[i];
    if (fwd) {
      key0[o] = a.x, key1[o] = b.x, y0[o] = a.y, dir[o] = 0, npos[o] = ~pos_of(a.y);
      if (y1) y1[o] = b.y;
      ++o;
    }
    if (rev) {
      const uint64_t fy = flip_y(b.y, b.x, rlen);
      key0[o] = b.x, key1[o] = a.x, y0[o] = fy, dir[o] = 1, npos[o] = ~pos_of(fy);
      if (y1) y1[o] = flip_y(a.y, a.x, rlen);  // shmr_utils.c:388-396
    }
  }
}

__global__ void k_iota(uint32_t *__restrict__ v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}
__global__ void k_gather_u64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n,
                             uint64_t *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}
__global__ void k_gather_out(const uint64_t *__restrict__ y0, const uint8_t *__restrict__ dir,
                             const uint32_t *__restrict__ perm, uint32_t n, uint64_t *__restrict__ sy0,
                             uint8_t *__restrict__ sdir) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sy0[i] = y0[perm[i]], sdir[i] = dir[perm[i]];
}
__global__ void k_flags(const uint64_t *__restrict__ k0, const uint64_t *__restrict__ k1, uint32_t n,
                        uint8_t *__restrict__ fb, uint8_t *__restrict__ fg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool g = (i == 0) || k0[i] != k0[i - 1];
  fg[i] = g;
  fb[i] = g || k1[i] != k1[i - 1];
}
// which bucket does each group start with: position of the group's first record among the bucket starts
__global__ void k_group_first_bucket(const uint32_t *__restrict__ gstart, uint32_t ng,
                                     const uint32_t *__restrict__ bstart, uint32_t nbk, uint32_t *__restrict__ gb) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const uint32_t key = gstart[g];
  uint32_t lo = 0, hi = nbk;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (bstart[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  gb[g] = lo;
}

// sort key of a bucket for the host's table replay: (its key0 group, its first insertion)
__global__ void k_bucket_order_key(const uint32_t *__restrict__ gbucket, uint32_t ng, const uint32_t *__restrict__ bfirst,
                                   uint32_t nbk, uint64_t *__restrict__ key) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbk) return;
  uint32_t lo = 0, hi = ng;  // last group whose first bucket is <= b
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (gbucket[mid] <= b) lo = mid;
    else hi = mid;
  }
  key[b] = (uint64_t)lo << 32 | bfirst[b];
}

template <typename T>
HostArray<T> to_host(const DevBuf<T> &d, size_t n, size_t extra = 0) {  // (extra: room for a sentinel the caller appends)
  HostArray<T> h(n + extra);
  d.download(h.data(), n);
  return h;
}
}  // namespace

void dev_build_pairs(const uint32_t *d_rlen, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
                     const PairParams &pp, PairTables &out, unsigned flags, const pgx_mm128 *d_mmers, const pgx_mm_count *d_counts,
                     DevicePairs *keep_dev) {
  out = PairTables();
  if (keep_dev) *keep_dev = DevicePairs();
  if (n_mm == 0) return;
  PGX_REQUIRE(n_mm < (1ULL << 31) && n_counts < (1ULL << 31), PGX_EARG, "shimmer list too long for one chunk");
  hipStream_t st = ctx().stream;
  KernelTimer tm("pairs", n_mm);
  Tmp tmp;
  size_t bytes = 0;
  const uint32_t n = (uint32_t)n_mm;

  // ---- aggregated counts ----------------------------------------------------------------------------------
  DevBuf<pgx_mm_count> cin_own(d_counts ? 0 : n_counts);  // (lists that are already on the device are used in place)
  if (!d_counts) cin_own.upload(counts, n_counts);
  const pgx_mm_count *cin = d_counts ? d_counts : cin_own.p;
  DevBuf<uint64_t> mer(n_counts), mer_s(n_counts), umer(n_counts);
  DevBuf<uint32_t> cnt(n_counts), cnt_s(n_counts), ucnt(n_counts), d_nu(1);
  uint32_t nu = 0;
  if (n_counts) {
    hipLaunchKernelGGL(k_split_counts, dim3(cdiv(n_counts, 256)), dim3(256), 0, st, cin, n_counts, mer.p, cnt.p);
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, mer.p, mer_s.p, cnt.p, cnt_s.p, (int)n_counts, 0, 56, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, mer.p, mer_s.p, cnt.p, cnt_s.p, (int)n_counts, 0, 56, st));
    bytes = 0;
    PGX_HIP(hipcub::DeviceReduce::ReduceByKey(nullptr, bytes, mer_s.p, umer.p, cnt_s.p, ucnt.p, d_nu.p, hipcub::Sum(),
                                             (int)n_counts, st));
    PGX_HIP(hipcub::DeviceReduce::ReduceByKey(tmp.get(bytes), bytes, mer_s.p, umer.p, cnt_s.p, ucnt.p, d_nu.p, hipcub::Sum(),
                                             (int)n_counts, st));
    d_nu.download(&nu, 1);
    sync();
  }

  // ---- keep flags, chain, records -------------------------------------------------------------------------
  DevBuf<pgx_mm128> mm_own(d_mmers ? 0 : n);
  if (!d_mmers) mm_own.upload(mmers, n);
  const pgx_mm128 *mm_dev = d_mmers ? d_mmers : mm_own.p;
  DevBuf<uint8_t> keep(n);
  DevBuf<uint32_t> d_misc(2);  // [0] first strict index, [1] missing hashes
  const uint32_t init[2] = {0xFFFFFFFFu, 0u};
  d_misc.upload(init, 2);
  hipLaunchKernelGGL(k_keep, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, umer.p, ucnt.p, nu, pp.lower, pp.upper, keep.p,
                     d_misc.p, d_misc.p + 1);
  DevBuf<int32_t> chain_in(n), chain(n);
  hipLaunchKernelGGL(k_chain_in, dim3(cdiv(n, 256)), dim3(256), 0, st, keep.p, n, d_misc.p, chain_in.p);
  bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(tmp.get(bytes), bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  // chain[i] == i  <=> i is kept ; chain[i-1] = previous kept shimmer (or -1)
  const uint32_t T = pp.total, c = pp.mychunk % T;
  DevBuf<uint32_t> nrec(n), off(n + 1);
  hipLaunchKernelGGL(k_records<0>, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, chain.p, T, c, d_rlen, nrec.p,
                     (const uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr,
                     (uint8_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr);
  PGX_HIP(hipMemsetAsync(off.p, 0, sizeof(uint32_t), st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, nrec.p, off.p + 1, (int)n, st));
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp.get(bytes), bytes, nrec.p, off.p + 1, (int)n, st));
  uint32_t misc[2], nr = 0;
  d_misc.download(misc, 2);
  PGX_HIP(hipMemcpyAsync(&nr, off.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  sync();
  PGX_REQUIRE(misc[1] == 0, PGX_EARG, "%u shimmer hashes are missing from the MC files", misc[1]);
  out.n_rec = nr;
  if (flags & PAIRS_COUNTS) {  // the aggregated multiplicity table, sorted by mer
    out.umer = to_host(umer, nu);
    out.ucnt = to_host(ucnt, nu);
    sync();
  }
  if (nr == 0) return;
  DevBuf<uint64_t> key0(nr), key1(nr), y0(nr), y1((flags & PAIRS_Y1) ? nr : 0);
  DevBuf<uint8_t> dir(nr);
  DevBuf<uint32_t> npos(nr);
  hipLaunchKernelGGL(k_records<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, chain.p, T, c, d_rlen, (uint32_t *)nullptr,
                     off.p, key0.p, key1.p, y0.p, dir.p, npos.p, y1.p);

  // ---- bucket order: stable LSD sorts (position desc, key1, key0) carrying the record index --------------------
  DevBuf<uint32_t> idx(nr), perm_a(nr), perm_b(nr), k32s(nr);
  DevBuf<uint64_t> kg(nr), kgs(nr);
  {
    if (flags & PAIRS_INSERTION_ORDER) {  // buckets keep their records in insertion order (shmr_map never sorts them)
      hipLaunchKernelGGL(k_iota, dim3(cdiv(nr, 256)), dim3(256), 0, st, perm_a.p, nr);
    } else {
      hipLaunchKernelGGL(k_iota, dim3(cdiv(nr, 256)), dim3(256), 0, st, idx.p, nr);
      bytes = 0;
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, npos.p, k32s.p, idx.p, perm_a.p, (int)nr, 0, 32, st));
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, npos.p, k32s.p, idx.p, perm_a.p, (int)nr, 0, 32, st));
    }
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, key1.p, perm_a.p, nr, kg.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kg.p, kgs.p, perm_a.p, perm_b.p, (int)nr, 0, 64, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, kg.p, kgs.p, perm_a.p, perm_b.p, (int)nr, 0, 64, st));
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, key0.p, perm_b.p, nr, kg.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kg.p, kgs.p, perm_b.p, perm_a.p, (int)nr, 0, 64, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, kg.p, kgs.p, perm_b.p, perm_a.p, (int)nr, 0, 64, st));
  }
  // perm_a = final order; kgs = sorted key0
  DevBuf<uint64_t> sk1(nr), sy0(nr);
  DevBuf<uint8_t> sdir(nr), fb(nr), fg(nr);
  hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, key1.p, perm_a.p, nr, sk1.p);
  hipLaunchKernelGGL(k_gather_out, dim3(cdiv(nr, 256)), dim3(256), 0, st, y0.p, dir.p, perm_a.p, nr, sy0.p, sdir.p);
  hipLaunchKernelGGL(k_flags, dim3(cdiv(nr, 256)), dim3(256), 0, st, kgs.p, sk1.p, nr, fb.p, fg.p);
  DevBuf<uint32_t> bstart(nr + 1), gstart(nr + 1), d_n(2);
  bytes = 0;
  PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, CountIt(0), fb.p, bstart.p, d_n.p, (int)nr, st));
  PGX_HIP(hipcub::DeviceSelect::Flagged(tmp.get(bytes), bytes, CountIt(0), fb.p, bstart.p, d_n.p, (int)nr, st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, CountIt(0), fg.p, gstart.p, d_n.p + 1, (int)nr, st));
  PGX_HIP(hipcub::DeviceSelect::Flagged(tmp.get(bytes), bytes, CountIt(0), fg.p, gstart.p, d_n.p + 1, (int)nr, st));
  uint32_t nbg[2];
  d_n.download(nbg, 2);
  sync();
  const uint32_t nbk = nbg[0], ng = nbg[1];
  PGX_HIP(hipMemcpyAsync(bstart.p + nbk, &nr, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  PGX_HIP(hipMemcpyAsync(gstart.p + ng, &nr, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  // first / last insertion (seq == original record index == perm value) per bucket and per key0 group
  DevBuf<uint32_t> bfirst(nbk), gfirst(ng), glast(ng), gbucket(ng);
  bytes = 0;
  PGX_HIP(hipcub::DeviceSegmentedReduce::Min(nullptr, bytes, perm_a.p, bfirst.p, (int)nbk, bstart.p, bstart.p + 1, st));
  PGX_HIP(hipcub::DeviceSegmentedReduce::Min(tmp.get(bytes), bytes, perm_a.p, bfirst.p, (int)nbk, bstart.p, bstart.p + 1, st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceSegmentedReduce::Min(nullptr, bytes, perm_a.p, gfirst.p, (int)ng, gstart.p, gstart.p + 1, st));
  PGX_HIP(hipcub::DeviceSegmentedReduce::Min(tmp.get(bytes), bytes, perm_a.p, gfirst.p, (int)ng, gstart.p, gstart.p + 1, st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceSegmentedReduce::Max(nullptr, bytes, perm_a.p, glast.p, (int)ng, gstart.p, gstart.p + 1, st));
  PGX_HIP(hipcub::DeviceSegmentedReduce::Max(tmp.get(bytes), bytes, perm_a.p, glast.p, (int)ng, gstart.p, gstart.p + 1, st));
  hipLaunchKernelGGL(k_group_first_bucket, dim3(cdiv(ng, 256)), dim3(256), 0, st, gstart.p, ng, bstart.p, nbk, gbucket.p);

  // insertion orders the host replays the two khash levels in: groups by first insertion, buckets by (group, first insertion)
  DevBuf<uint32_t> gord(ng), bord(nbk), iota_g(ng), iota_b(nbk), gf_sorted(ng);
  DevBuf<uint64_t> bok(nbk), bok_sorted(nbk);
  {
    hipLaunchKernelGGL(k_iota, dim3(cdiv(ng, 256)), dim3(256), 0, st, iota_g.p, ng);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, gfirst.p, gf_sorted.p, iota_g.p, gord.p, (int)ng, 0, 32, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, gfirst.p, gf_sorted.p, iota_g.p, gord.p, (int)ng, 0, 32, st));
    hipLaunchKernelGGL(k_iota, dim3(cdiv(nbk, 256)), dim3(256), 0, st, iota_b.p, nbk);
    hipLaunchKernelGGL(k_bucket_order_key, dim3(cdiv(nbk, 256)), dim3(256), 0, st, gbucket.p, ng, bfirst.p, nbk, bok.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, bok.p, bok_sorted.p, iota_b.p, bord.p, (int)nbk, 0, 64, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, bok.p, bok_sorted.p, iota_b.p, bord.p, (int)nbk, 0, 64, st));
  }

  const bool trace = getenv("PGX_TRACE") != nullptr;
  double tj0 = 0;
  if (trace) {
    sync();
    tj0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  out.y0 = to_host(sy0, nr);
  out.dir = to_host(sdir, nr);
  DevBuf<uint64_t> sy1((flags & PAIRS_Y1) ? nr : 0);
  if (flags & PAIRS_Y1) {
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, y1.p, perm_a.p, nr, sy1.p);
    out.y1 = to_host(sy1, nr);
  }
  out.gord = to_host(gord, ng);
  out.bord = to_host(bord, nbk);
  DevBuf<uint64_t> bkey1(nbk), gkey0(ng);
  hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nbk, 256)), dim3(256), 0, st, sk1.p, bstart.p, nbk, bkey1.p);
  hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(ng, 256)), dim3(256), 0, st, kgs.p, gstart.p, ng, gkey0.p);
  out.bkey1 = to_host(bkey1, nbk);
  out.gkey0 = to_host(gkey0, ng);
  out.bstart = to_host(bstart, (size_t)nbk + 1);
  out.bfirst = to_host(bfirst, nbk);
  out.gstart = to_host(gstart, (size_t)ng + 1);
  out.gfirst = to_host(gfirst, ng);
  out.glast = to_host(glast, ng);
  out.gbucket = to_host(gbucket, ng, 1);
  sync();
  out.gbucket[ng] = nbk;
  if (keep_dev) {  // the device replay reads the sorted records where they are
    keep_dev->y0 = std::move(sy0), keep_dev->dir = std::move(sdir), keep_dev->bstart = std::move(bstart);
    keep_dev->n_rec = nr, keep_dev->n_buckets = nbk, keep_dev->valid = true;
  }
  if (trace)
    fprintf(stderr, "[pgx]   join: tables downloaded in %.2f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tj0);
}

}  // namespace pgx
