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

// The multiplicity lookup of build_map (shmr_utils.c:305-320).  The aggregated table is sorted by hash, but a binary search costs
// ~10 cold sectors per lookup (3.75 GB of reads for 11 M lookups at 4.5 Gbases), and bucketing by the hash's top bits does not
// work: shimmers are MINIMA, their hashes crowd the bottom of the range.  So the lookups go through an open-addressing table
// built from the aggregated entries (16-byte slots {hash + 1, count}, load factor <= 1/2): ~1.3 probes of one sector each.
struct CSlot {
  unsigned long long key;   // hash + 1 (0 = empty)
  uint32_t cnt, pad;
};
__device__ __forceinline__ uint32_t cmix(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 29;
  return (uint32_t)h;
}
__global__ void k_count_insert(const uint64_t *__restrict__ mer, const uint32_t *__restrict__ cnt, uint32_t nu, CSlot *__restrict__ tab,
                               uint32_t mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nu) return;
  const unsigned long long want = mer[i] + 1;   // (the keys of the aggregated table are distinct)
  for (uint32_t h = cmix(mer[i]) & mask;; h = (h + 1) & mask)
    if (atomicCAS(&tab[h].key, 0ULL, want) == 0ULL) {
      tab[h].cnt = cnt[i];
      return;
    }
}
__device__ __forceinline__ bool lookup_count(const CSlot *tab, uint32_t mask, uint64_t key, uint32_t *out) {
  const unsigned long long want = key + 1;
  for (uint32_t h = cmix(key) & mask, probes = 0; probes <= mask; h = (h + 1) & mask, ++probes) {
    const uint4 v = *reinterpret_cast<const uint4 *>(&tab[h]);
    const unsigned long long k = (unsigned long long)v.y << 32 | v.x;
    if (k == want) {
      *out = v.z;
      return true;
    }
    if (k == 0) return false;
  }
  return false;
}

// keep[i] = lower <= count <= upper ; first_strict = min i with lower <= count < upper ; missing = hash not in the table
// misc[0] = first strict index, misc[1] = hashes missing from the table, misc[2] = shimmers that do not fit the seqdb (read
// id beyond the idx, or position beyond the read: the reference asserts on the read-length lookup, shmr_utils.c:374-376),
// misc[3] = one such read id
__global__ void k_keep(const pgx_mm128 *__restrict__ mm, uint32_t n,
                       const CSlot *__restrict__ tab, uint32_t tmask, uint32_t lower, uint32_t upper,
                       uint8_t *__restrict__ keep, uint32_t *__restrict__ first_strict, uint32_t *__restrict__ missing,
                       const uint32_t *__restrict__ rlen, uint32_t n_rid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (n_rid) {
    const uint64_t y = mm[i].y;
    const uint32_t rid = (uint32_t)(y >> 32);
    if (rid >= n_rid || (uint32_t)((y & 0xFFFFFFFFu) >> 1) >= rlen[rid]) {
      atomicAdd(missing + 1, 1u);
      missing[2] = rid;
      keep[i] = 0;
      return;
    }
  }
  uint32_t c = 0;
  if (!tab || !lookup_count(tab, tmask, mm[i].x >> 8, &c)) {
    atomicAdd(missing, 1u);
    keep[i] = 0;
    return;
  }
  keep[i] = (c >= lower && c <= upper);
  // (nearly every element qualifies: one atomic per wavefront, and only while it can still lower the minimum -- 11 M atomics on
  // one address were 2 ms of this kernel)
  if (c >= lower && c < upper && i < *(volatile uint32_t *)first_strict) {
    const uint64_t m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicMin(first_strict, i);
  }
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

// ---- multi-GPU scatter: the records of ONE index chunk's reads for EVERY overlap chunk ------------------------
// MODE 0: number of records shimmer i produces (0..2) over all destinations; MODE 1: write them (record + destination chunk)
template <int MODE>
__global__ void k_records_all(const pgx_mm128 *__restrict__ mm, uint32_t n, const int32_t *__restrict__ chain, uint32_t T,
                              const uint32_t *__restrict__ rlen, uint32_t *__restrict__ nrec, const uint32_t *__restrict__ off,
                              pgx_pair_rec *__restrict__ rec, uint8_t *__restrict__ dest) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool pair = false;
  pgx_mm128 a{0, 0}, b{0, 0};
  if (i > 0 && chain[i] == (int32_t)i) {
    const int32_t p = chain[i - 1];
    if (p >= 0) {
      a = mm[p], b = mm[i];
      pair = (a.y >> 32) == (b.y >> 32) && pos28(b.y) - pos28(a.y) >= 100u;
    }
  }
  if (MODE == 0) {
    nrec[i] = pair ? 2u : 0u;
  } else if (pair) {
    const uint32_t o = off[i];
    pgx_pair_rec f, r;
    memset(&f, 0, sizeof(f)), memset(&r, 0, sizeof(r));
    f.key0 = a.x, f.key1 = b.x, f.y0 = a.y, f.dir = 0, f.npos = ~pos_of(a.y);
    const uint64_t fy = flip_y(b.y, b.x, rlen);
    r.key0 = b.x, r.key1 = a.x, r.y0 = fy, r.dir = 1, r.npos = ~pos_of(fy);
    rec[o] = f, dest[o] = (uint8_t)((a.x >> 8) % T);          // owner chunk c with c % T == (key0 >> 8) % T (shmr_utils.c:337,362)
    rec[o + 1] = r, dest[o + 1] = (uint8_t)((b.x >> 8) % T);
  }
}
__global__ void k_dest_hist(const uint8_t *__restrict__ dest, uint32_t n, uint32_t *__restrict__ hist /* 256 */) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&h[dest[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_gather_rec(const pgx_pair_rec *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n,
                             pgx_pair_rec *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[perm[i]];
}
__global__ void k_unpack_rec(const pgx_pair_rec *__restrict__ rec, uint32_t n, uint64_t *__restrict__ key0,
                             uint64_t *__restrict__ key1, uint64_t *__restrict__ y0, uint8_t *__restrict__ dir,
                             uint32_t *__restrict__ npos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const pgx_pair_rec r = rec[i];
  key0[i] = r.key0, key1[i] = r.key1, y0[i] = r.y0, dir[i] = r.dir, npos[i] = r.npos;
}

// ---- first occurrence of every first key (early outer-table keys) ---------------------------------------------------------
__device__ __forceinline__ uint64_t mixk(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 33, h *= 0xc4ceb9fe1a85ec53ULL, h ^= h >> 33;
  return h;
}
__global__ void k_first_insert(const uint64_t *__restrict__ key0, uint32_t nr, unsigned long long *__restrict__ tkeys,
                               uint32_t *__restrict__ tseq, uint32_t mask) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nr) return;
  const unsigned long long k = key0[i] + 1;  // (never 0: a key is hash << 8 | span with span > 0)
  uint32_t h = (uint32_t)mixk(k) & mask;
  for (;;) {
    unsigned long long cur = tkeys[h];
    if (cur == 0) cur = atomicCAS(&tkeys[h], 0ULL, k);
    if (cur == 0 || cur == k) {
      atomicMin(&tseq[h], i);
      return;
    }
    h = (h + 1) & mask;
  }
}
__global__ void k_first_flag(const uint64_t *__restrict__ key0, uint32_t nr, const unsigned long long *__restrict__ tkeys,
                             const uint32_t *__restrict__ tseq, uint32_t mask, uint8_t *__restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nr) return;
  const unsigned long long k = key0[i] + 1;
  uint32_t h = (uint32_t)mixk(k) & mask;
  while (tkeys[h] != k) h = (h + 1) & mask;
  flag[i] = tseq[h] == i;
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

// first / last insertion per bucket and per key0 group.  The segments are short (a bucket holds ~4 records, a group ~5
// buckets), which is the worst case of a block-per-segment reduction (hipcub::DeviceSegmentedReduce: 1 ms per call for 3.9 M
// buckets); a thread per segment reads the same bytes in ~0.05 ms.
__global__ void k_bucket_minmax(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ bstart, uint32_t nbk,
                                uint32_t *__restrict__ bfirst, uint32_t *__restrict__ blast) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbk) return;
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  for (uint32_t i = bstart[b], e = bstart[b + 1]; i < e; ++i) {
    const uint32_t v = perm[i];
    mn = min(mn, v), mx = max(mx, v);
  }
  bfirst[b] = mn, blast[b] = mx;
}
__global__ void k_group_minmax(const uint32_t *__restrict__ bfirst, const uint32_t *__restrict__ blast, const uint32_t *__restrict__ gbucket,
                               uint32_t ng, uint32_t nbk, uint32_t *__restrict__ gfirst, uint32_t *__restrict__ glast) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  uint32_t mn = 0xFFFFFFFFu, mx = 0;
  for (uint32_t b = gbucket[g], e = g + 1 < ng ? gbucket[g + 1] : nbk; b < e; ++b) mn = min(mn, bfirst[b]), mx = max(mx, blast[b]);
  gfirst[g] = mn, glast[g] = mx;
}

// sort key of a bucket for the host's table replay: (its key0 group, its first insertion)
__global__ void k_bucket_order_key(const uint32_t *__restrict__ gbucket, uint32_t ng, const uint32_t *__restrict__ bfirst,
                                   uint32_t nbk, int fbits, uint64_t *__restrict__ key) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbk) return;
  uint32_t lo = 0, hi = ng;  // last group whose first bucket is <= b
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (gbucket[mid] <= b) lo = mid;
    else hi = mid;
  }
  key[b] = (uint64_t)lo << fbits | bfirst[b];
}

// bucket tables in the order the host replays the inner tables in
__global__ void k_bucket_ord(const uint32_t *__restrict__ bord, const uint32_t *__restrict__ bstart, const uint64_t *__restrict__ sk1,
                             uint32_t nbk, uint64_t *__restrict__ bkey1_o, uint32_t *__restrict__ bn_o) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nbk) return;
  const uint32_t b = bord[i], s0 = bstart[b];
  bkey1_o[i] = sk1[s0], bn_o[i] = bstart[b + 1] - s0;
}
__global__ void k_group_trail(const uint32_t *__restrict__ gbucket, const uint32_t *__restrict__ bord, const uint32_t *__restrict__ bfirst,
                              const uint32_t *__restrict__ glast, uint32_t ng, uint32_t nbk, uint8_t *__restrict__ gtrail) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const uint32_t b1 = g + 1 < ng ? gbucket[g + 1] : nbk;
  gtrail[g] = bfirst[bord[b1 - 1]] < glast[g];   // the group's last new bucket is not its last record: one more put follows
}

template <typename T>
HostArray<T> to_host(const DevBuf<T> &d, size_t n, size_t extra = 0) {  // (extra: room for a sentinel the caller appends)
  HostArray<T> h(n + extra);
  d.download(h.data(), n);
  return h;
}
}  // namespace

// the records of one overlap chunk in insertion order, structure of arrays
struct PairRecs {
  DevBuf<uint64_t> key0, key1, y0, y1;
  DevBuf<uint8_t> dir;
  DevBuf<uint32_t> npos;
  uint32_t nr = 0;
};

// aggregated multiplicities of all MC entries, sorted by mer (aggregate_mm_count, shmr_utils.c:162-176)
struct CountTable {
  DevBuf<uint64_t> umer;
  DevBuf<uint32_t> ucnt;
  uint32_t nu = 0;
  DevBuf<CSlot> tab;            // the same entries as an open-addressing table (lookup_count)
  uint32_t tmask = 0;
};
static void aggregate_counts_now(const pgx_mm_count *cin, size_t n_counts, CountTable &ct, Tmp &tmp) {
  hipStream_t st = ctx().stream;
  ct.umer.alloc(n_counts), ct.ucnt.alloc(n_counts);
  ct.nu = 0;
  if (!n_counts) return;
  DevBuf<uint64_t> mer(n_counts), mer_s(n_counts);
  DevBuf<uint32_t> cnt(n_counts), cnt_s(n_counts), d_nu(1);
  size_t bytes = 0;
  hipLaunchKernelGGL(k_split_counts, dim3(cdiv(n_counts, 256)), dim3(256), 0, st, cin, n_counts, mer.p, cnt.p);
  PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, mer.p, mer_s.p, cnt.p, cnt_s.p, (int)n_counts, 0, 56, st));
  PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, mer.p, mer_s.p, cnt.p, cnt_s.p, (int)n_counts, 0, 56, st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceReduce::ReduceByKey(nullptr, bytes, mer_s.p, ct.umer.p, cnt_s.p, ct.ucnt.p, d_nu.p, hipcub::Sum(),
                                           (int)n_counts, st));
  PGX_HIP(hipcub::DeviceReduce::ReduceByKey(tmp.get(bytes), bytes, mer_s.p, ct.umer.p, cnt_s.p, ct.ucnt.p, d_nu.p, hipcub::Sum(),
                                           (int)n_counts, st));
  d_nu.download(&ct.nu, 1);
  sync();
  ct.tmask = 0;
  if (!ct.nu) return;
  uint32_t cap = 1024;
  while (cap < 2 * (size_t)ct.nu) cap <<= 1;
  ct.tab.alloc(cap);
  ct.tmask = cap - 1;
  PGX_HIP(hipMemsetAsync(ct.tab.p, 0, (size_t)cap * sizeof(CSlot), st));
  hipLaunchKernelGGL(k_count_insert, dim3(cdiv(ct.nu, 256)), dim3(256), 0, st, ct.umer.p, ct.ucnt.p, ct.nu, ct.tab.p, ct.tmask);
}


// Every overlap chunk of a job aggregates the SAME count files (shmr_overlap.c:359-384 globs them all), and a resident pipeline runs the
// chunks one after the other in this process: the table of the last call is kept and handed out again when the entries are the same --
// same number, same order-sensitive 64-bit checksum of (mer, count), one read-only pass over the entries (~1 ms per GB) instead of the
// split + sort + reduce + insert (57 ms per full-size configs[3] chunk, profiles/r05a_chunk_timeline_c4.txt).
__global__ __launch_bounds__(256) void k_counts_checksum(const pgx_mm_count *__restrict__ in, size_t n, unsigned long long *__restrict__ sum) {
  unsigned long long h = 0, h2 = 0;
  // a CONTIGUOUS run of entries per workgroup (round 6; a grid-stride loop over the 4 GB of a full-size job's count entries jumped 32 MB per
  // iteration in every wavefront -- a new translation range each time: 8.1 ms per chunk for what is a 1 ms stream)
  const size_t per = ((n + gridDim.x - 1) / gridDim.x + 255) & ~(size_t)255, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const uint4 v = *reinterpret_cast<const uint4 *>(in + i);   // mer (x, y), count (z); the padding word is not looked at
    unsigned long long x = ((unsigned long long)v.y << 32 | v.x) + 0x9E3779B97F4A7C15ULL * (i + 1);
    x ^= x >> 33, x *= 0xff51afd7ed558ccdULL, x ^= x >> 33;
    x += v.z;
    x *= 0xc4ceb9fe1a85ec53ULL, x ^= x >> 33;
    h += x;
    // a second, independent sum (other multipliers, the fields the other way round: ADVICE r5 -- one 64-bit additive checksum was all that
    // stood between two different sets of count files and the same keep flags)
    unsigned long long y = ((unsigned long long)v.z << 32 | v.y) ^ (0xD6E8FEB86659FD93ULL * (i + 0x632BE5ABULL));
    y ^= y >> 29, y *= 0xBF58476D1CE4E5B9ULL, y ^= y >> 32;
    y += v.x;
    y *= 0x94D049BB133111EBULL, y ^= y >> 31;
    h2 += y;
  }
  for (int o = 32; o; o >>= 1) {
    h += (unsigned long long)__shfl_xor((int)(h >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)h, o, 64);
    h2 += (unsigned long long)__shfl_xor((int)(h2 >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)h2, o, 64);
  }
  __shared__ unsigned long long part[4], part2[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h, part2[threadIdx.x >> 6] = h2;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]), atomicAdd(sum + 1, part2[0] + part2[1] + part2[2] + part2[3]);
}
namespace {
struct CountCache {
  size_t n = 0;
  unsigned long long sum[2] = {0, 0};
  CountTable ct;
  bool valid = false;
};
CountCache g_counts;
ShutdownHook g_counts_reset([] { g_counts = CountCache(); });
}  // namespace
void count_cache_drop() { g_counts = CountCache(); }   // (pgx_seqdb_free: the job is over, its table's HBM goes back)
static const CountTable &aggregate_counts(const pgx_mm_count *cin, size_t n_counts, Tmp &tmp) {
  unsigned long long sum[2] = {0, 0};
  if (n_counts) {
    hipStream_t st = ctx().stream;
    DevBuf<unsigned long long> d_sum(2);
    PGX_HIP(hipMemsetAsync(d_sum.p, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_counts_checksum, dim3((unsigned)std::min<size_t>(cdiv(n_counts, 256), 8192)), dim3(256), 0, st, cin, n_counts, d_sum.p);
    d_sum.download(sum, 2);
    sync();
    if (g_counts.valid && g_counts.n == n_counts && g_counts.sum[0] == sum[0] && g_counts.sum[1] == sum[1]) return g_counts.ct;
  }
  g_counts.valid = false;
  aggregate_counts_now(cin, n_counts, g_counts.ct, tmp);
  g_counts.n = n_counts, g_counts.sum[0] = sum[0], g_counts.sum[1] = sum[1], g_counts.valid = n_counts != 0;
  return g_counts.ct;
}

// keep flags of a list against the table; returns the first strict index (0xFFFFFFFF: none)
static uint32_t keep_flags(const pgx_mm128 *mm_dev, uint32_t n, const CountTable &ct, const PairParams &pp, const uint32_t *d_rlen,
                           DevBuf<uint8_t> &keep, DevBuf<uint32_t> &d_misc) {
  hipStream_t st = ctx().stream;
  keep.alloc(n);
  d_misc.alloc(4);
  const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
  d_misc.upload(init, 4);
  hipLaunchKernelGGL(k_keep, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, ct.nu ? ct.tab.p : (const CSlot *)nullptr, ct.tmask, pp.lower, pp.upper, keep.p,
                     d_misc.p, d_misc.p + 1, d_rlen, pp.n_rid);
  uint32_t misc[4];
  d_misc.download(misc, 4);
  sync();
  PGX_REQUIRE(misc[2] == 0, PGX_EARG,
              "%u shimmers of the list do not fit the read database (e.g. read %u: not in the idx, or a position beyond its length) -- "
              "shimmer files of another seqdb?", misc[2], misc[3]);
  PGX_REQUIRE(misc[1] == 0, PGX_EARG, "%u shimmer hashes are missing from the MC files", misc[1]);
  return misc[0];
}

// chain of kept shimmers from `start` on: chain[i] == i <=> i is kept; chain[i-1] = previous kept shimmer (or -1)
static void chain_scan(const DevBuf<uint8_t> &keep, uint32_t n, const uint32_t *d_start, DevBuf<int32_t> &chain, Tmp &tmp) {
  hipStream_t st = ctx().stream;
  DevBuf<int32_t> chain_in(n);
  chain.alloc(n);
  hipLaunchKernelGGL(k_chain_in, dim3(cdiv(n, 256)), dim3(256), 0, st, keep.p, n, d_start, chain_in.p);
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(tmp.get(bytes), bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  // (chain_in goes back to the block cache on return: stream-ordered reuse on the library's one stream)
}

static void bucketize(PairRecs &R, unsigned flags, PairTables &out, DevicePairs *keep_dev, Tmp &tmp);
static void early_groups(const PairRecs &R, Tmp &tmp, const EarlyFn &early);

void dev_build_pairs(const uint32_t *d_rlen, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
                     const PairParams &pp, PairTables &out, unsigned flags, const pgx_mm128 *d_mmers, const pgx_mm_count *d_counts,
                     DevicePairs *keep_dev, const EarlyFn &early, const pgx_seqdb *locus_db) {
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
  const CountTable &ct = aggregate_counts(d_counts ? d_counts : cin_own.p, n_counts, tmp);

  // ---- keep flags, chain, records -------------------------------------------------------------------------
  DevBuf<pgx_mm128> mm_own(d_mmers ? 0 : n);
  if (!d_mmers) mm_own.upload(mmers, n);
  const pgx_mm128 *mm_dev = d_mmers ? d_mmers : mm_own.p;
  if (locus_db) locus_key_add_mm(locus_db, mm_dev, n);   // (only before the database's packs exist)
  DevBuf<uint8_t> keep;
  DevBuf<uint32_t> d_misc;
  keep_flags(mm_dev, n, ct, pp, d_rlen, keep, d_misc);
  DevBuf<int32_t> chain;
  chain_scan(keep, n, d_misc.p, chain, tmp);
  const uint32_t T = pp.total, c = pp.mychunk % T;
  DevBuf<uint32_t> nrec(n), off(n + 1);
  hipLaunchKernelGGL(k_records<0>, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, chain.p, T, c, d_rlen, nrec.p,
                     (const uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr,
                     (uint8_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr);
  PGX_HIP(hipMemsetAsync(off.p, 0, sizeof(uint32_t), st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, nrec.p, off.p + 1, (int)n, st));
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp.get(bytes), bytes, nrec.p, off.p + 1, (int)n, st));
  uint32_t nr = 0;
  PGX_HIP(hipMemcpyAsync(&nr, off.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  sync();
  out.n_rec = nr;
  if (flags & PAIRS_COUNTS) {  // the aggregated multiplicity table, sorted by mer
    out.umer = to_host(ct.umer, ct.nu);
    out.ucnt = to_host(ct.ucnt, ct.nu);
    sync();
  }
  if (nr == 0) return;
  PairRecs R;
  R.nr = nr;
  R.key0.alloc(nr), R.key1.alloc(nr), R.y0.alloc(nr), R.y1.alloc((flags & PAIRS_Y1) ? nr : 0), R.dir.alloc(nr), R.npos.alloc(nr);
  hipLaunchKernelGGL(k_records<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, mm_dev, n, chain.p, T, c, d_rlen, (uint32_t *)nullptr,
                     off.p, R.key0.p, R.key1.p, R.y0.p, R.dir.p, R.npos.p, R.y1.p);
  early_groups(R, tmp, early);
  bucketize(R, flags, out, keep_dev, tmp);
}

// ---- multi-GPU: prepare / scatter / from-records (one context per process: the state lives here between the calls) ---------
namespace {
struct ScatterState {
  const pgx_mm128 *mm = nullptr;
  uint32_t n = 0;
  DevBuf<uint8_t> keep;
  DevBuf<pgx_pair_rec> send;
  bool ready = false;
  uint64_t ix_gen = 0;   // index_generation() at prepare, when `mm` is a view of index-stage memory; 0: caller-owned
};
ScatterState g_scatter;
ShutdownHook g_scatter_reset([] { g_scatter = ScatterState(); });
}  // namespace

int64_t dev_pairs_prepare(const uint32_t *d_rlen, uint32_t n_rid, const pgx_mm128 *d_mm, size_t n_mm, const pgx_mm_count *d_counts,
                          size_t n_counts, uint32_t lower, uint32_t upper) {
  PGX_REQUIRE(n_mm < (1ULL << 31) && n_counts < (1ULL << 31), PGX_EARG, "shimmer list too long for one chunk");
  g_scatter = ScatterState();
  g_scatter.mm = d_mm, g_scatter.n = (uint32_t)n_mm;
  g_scatter.ready = true, g_scatter.ix_gen = index_owns(d_mm) ? index_generation() : 0;
  if (n_mm == 0) return -1;
  Tmp tmp;
  const CountTable &ct = aggregate_counts(d_counts, n_counts, tmp);
  DevBuf<uint32_t> d_misc;
  PairParams pp{1, 1, lower, upper, n_rid};
  const uint32_t first = keep_flags(d_mm, (uint32_t)n_mm, ct, pp, d_rlen, g_scatter.keep, d_misc);
  return first == 0xFFFFFFFFu ? -1 : (int64_t)first;
}

void dev_pairs_scatter(const uint32_t *d_rlen, uint32_t T, int64_t start, const pgx_pair_rec **d_send, uint64_t *counts) {
  PGX_REQUIRE(g_scatter.ready, PGX_ESTATE, "pgx_pairs_scatter_dev without pgx_pairs_prepare_dev");
  PGX_REQUIRE(g_scatter.ix_gen == 0 || g_scatter.ix_gen == index_generation(), PGX_ESTATE,
              "pgx_pairs_scatter_dev: an index-stage call rewrote the shimmer list since pgx_pairs_prepare_dev (the list is a view of "
              "the index workspace; copy it with pgx_copy_dev to overlap steps)");
  PGX_REQUIRE(T >= 1 && T <= 256, PGX_EARG, "the record scatter supports 1..256 overlap chunks");
  for (uint32_t c = 0; c < T; ++c) counts[c] = 0;
  *d_send = nullptr;
  const uint32_t n = g_scatter.n;
  if (n == 0 || start < 0 || start >= (int64_t)n) return;  // (no eligible anchor on this rank: shmr_utils.c:311-320)
  hipStream_t st = ctx().stream;
  KernelTimer tm("pairs_scatter", n);
  Tmp tmp;
  DevBuf<uint32_t> d_start(1);
  const uint32_t s32 = (uint32_t)start;
  d_start.upload(&s32, 1);
  DevBuf<int32_t> chain;
  chain_scan(g_scatter.keep, n, d_start.p, chain, tmp);
  DevBuf<uint32_t> nrec(n), off(n + 1);
  hipLaunchKernelGGL(k_records_all<0>, dim3(cdiv(n, 256)), dim3(256), 0, st, g_scatter.mm, n, chain.p, T, d_rlen, nrec.p,
                     (const uint32_t *)nullptr, (pgx_pair_rec *)nullptr, (uint8_t *)nullptr);
  PGX_HIP(hipMemsetAsync(off.p, 0, sizeof(uint32_t), st));
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, nrec.p, off.p + 1, (int)n, st));
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp.get(bytes), bytes, nrec.p, off.p + 1, (int)n, st));
  uint32_t nr = 0;
  PGX_HIP(hipMemcpyAsync(&nr, off.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  sync();
  if (nr == 0) return;
  PGX_REQUIRE(nr < (1u << 31), PGX_EARG, "too many pair records for one index chunk");
  DevBuf<pgx_pair_rec> rec(nr);
  DevBuf<uint8_t> dest(nr), dest_s(nr);
  DevBuf<uint32_t> idx(nr), perm(nr), hist(256);
  hipLaunchKernelGGL(k_records_all<1>, dim3(cdiv(n, 256)), dim3(256), 0, st, g_scatter.mm, n, chain.p, T, d_rlen, (uint32_t *)nullptr,
                     off.p, rec.p, dest.p);
  // stable by destination: the records of a destination stay in scan order (= insertion order of the receiving chunk)
  hipLaunchKernelGGL(k_iota, dim3(cdiv(nr, 256)), dim3(256), 0, st, idx.p, nr);
  bytes = 0;
  PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dest.p, dest_s.p, idx.p, perm.p, (int)nr, 0, 8, st));
  PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, dest.p, dest_s.p, idx.p, perm.p, (int)nr, 0, 8, st));
  g_scatter.send.alloc(nr);
  hipLaunchKernelGGL(k_gather_rec, dim3(cdiv(nr, 256)), dim3(256), 0, st, rec.p, perm.p, nr, g_scatter.send.p);
  PGX_HIP(hipMemsetAsync(hist.p, 0, 256 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_dest_hist, dim3(std::min<unsigned>(cdiv(nr, 256), 1024u)), dim3(256), 0, st, dest.p, nr, hist.p);
  uint32_t h[256];
  hist.download(h, 256);
  sync();
  // dest value v = (key0 >> 8) % T belongs to chunk c with c % T == v: chunk T for v == 0, else chunk v.  Send order = chunk
  // order 1..T = values 1, 2, ..., T-1, 0 -- but the radix sort grouped by VALUE (0 first): rotate the value-0 group to the end
  // by reporting the counts in chunk order together with a rotated buffer.
  if (T > 1 && h[0]) {
    DevBuf<pgx_pair_rec> rot(nr);
    const size_t n0 = h[0];
    PGX_HIP(hipMemcpyAsync(rot.p, g_scatter.send.p + n0, (nr - n0) * sizeof(pgx_pair_rec), hipMemcpyDeviceToDevice, st));
    PGX_HIP(hipMemcpyAsync(rot.p + (nr - n0), g_scatter.send.p, n0 * sizeof(pgx_pair_rec), hipMemcpyDeviceToDevice, st));
    sync();
    g_scatter.send = std::move(rot);
  }
  for (uint32_t c = 1; c <= T; ++c) counts[c - 1] = h[c % T];
  *d_send = g_scatter.send.p;
}

void pairs_fetch_tables(const DevicePairs &dp, PairTables &out) {
  if (out.on_host || !dp.tables) return;
  const size_t ng = dp.n_groups, nbk = dp.n_buckets;
  out.gord = to_host(dp.gord, ng);
  out.bord = to_host(dp.bord, nbk);
  out.bkey1_ord = to_host(dp.bkey1_ord, nbk);
  out.bn_ord = to_host(dp.bn_ord, nbk);
  out.gtrail = to_host(dp.gtrail, ng);
  out.gkey0 = to_host(dp.gkey0, ng);
  out.bstart = to_host(dp.bstart, nbk + 1);
  out.gstart = to_host(dp.gstart, ng + 1);
  out.gfirst = to_host(dp.gfirst, ng);
  out.glast = to_host(dp.glast, ng);
  out.gbucket = to_host(dp.gbucket, ng + 1);
  sync();
  out.on_host = true;
}

void pairs_fetch_records(const DevicePairs &dp, PairTables &out) {
  if (!dp.valid || out.y0.size() == dp.n_rec) return;
  out.y0 = to_host(dp.y0, dp.n_rec);
  out.dir = to_host(dp.dir, dp.n_rec);
  sync();
}

void dev_pairs_from_records(const pgx_pair_rec *d_rec, size_t n, PairTables &out, DevicePairs *keep_dev, unsigned flags,
                            const EarlyFn &early, const pgx_seqdb *locus_db) {
  out = PairTables();
  if (keep_dev) *keep_dev = DevicePairs();
  if (n == 0) return;
  PGX_REQUIRE(n < (1ULL << 31), PGX_EARG, "too many pair records for one overlap chunk");
  KernelTimer tm("pairs", n);
  Tmp tmp;
  const uint32_t nr = (uint32_t)n;
  PairRecs R;
  R.nr = nr;
  R.key0.alloc(nr), R.key1.alloc(nr), R.y0.alloc(nr), R.dir.alloc(nr), R.npos.alloc(nr);
  hipLaunchKernelGGL(k_unpack_rec, dim3(cdiv(nr, 256)), dim3(256), 0, ctx().stream, d_rec, nr, R.key0.p, R.key1.p, R.y0.p, R.dir.p,
                     R.npos.p);
  out.n_rec = nr;
  if (locus_db) locus_key_add_records(locus_db, R.key0.p, R.y0.p, nr);
  early_groups(R, tmp, early);
  bucketize(R, flags, out, keep_dev, tmp);
}

static void early_groups(const PairRecs &R, Tmp &tmp, const EarlyFn &early) {
  if (!early || R.nr == 0) return;
  hipStream_t st = ctx().stream;
  const uint32_t nr = R.nr;
  size_t cap = 1024;
  while (cap < (size_t)nr * 2) cap <<= 1;
  // This pass is an optimisation on top of the join's own buffers (12 B per table slot + 5 B per record: ~0.8 GB at 4.5 Gbases,
  // ~3 GB at 18 Gbases) and runs before bucketize releases anything.  With no headroom left it is skipped -- the callback stays
  // uncalled and the outer table is replayed after the join, exactly as with PGX_EARLY_OUTER=0 -- instead of failing a chunk
  // that fits without it.
  DevBuf<unsigned long long> tkeys;
  DevBuf<uint32_t> tseq, idx, d_n;
  DevBuf<uint8_t> flag;
  try {
    tkeys.alloc(cap), tseq.alloc(cap), idx.alloc(nr), d_n.alloc(1), flag.alloc(nr);
  } catch (const Fail &) {
    (void)hipGetLastError();
    if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] join: no device memory for the early outer-key pass (%zu MB); skipped\n", (cap * 12 + (size_t)nr * 5) >> 20);
    return;
  }
  PGX_HIP(hipMemsetAsync(tkeys.p, 0, cap * sizeof(unsigned long long), st));
  PGX_HIP(hipMemsetAsync(tseq.p, 0xFF, cap * sizeof(uint32_t), st));
  hipLaunchKernelGGL(k_first_insert, dim3(cdiv(nr, 256)), dim3(256), 0, st, R.key0.p, nr, tkeys.p, tseq.p, (uint32_t)(cap - 1));
  hipLaunchKernelGGL(k_first_flag, dim3(cdiv(nr, 256)), dim3(256), 0, st, R.key0.p, nr, tkeys.p, tseq.p, (uint32_t)(cap - 1), flag.p);
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, CountIt(0), flag.p, idx.p, d_n.p, (int)nr, st));
  PGX_HIP(hipcub::DeviceSelect::Flagged(tmp.get(bytes), bytes, CountIt(0), flag.p, idx.p, d_n.p, (int)nr, st));
  uint32_t nd = 0;
  d_n.download(&nd, 1);
  sync();
  EarlyGroups eg;
  eg.n = nd;
  if (nd) {
    DevBuf<uint64_t> keys(nd);
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nd, 256)), dim3(256), 0, st, R.key0.p, idx.p, nd, keys.p);
    eg.keys = to_host(keys, nd);
    PGX_HIP(hipMemcpyAsync(&eg.last_first, idx.p + nd - 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sync();
  }
  early(std::move(eg));
}

// PAIRS_DEV_TABLES: what the host wants to know about the groups (stats: [0] most buckets in a group, [1] groups beyond `lane_max`
// -- listed in `big` --, [2] first insertion of the group inserted last) and every stride-th first key in first-insertion order
__global__ void k_group_stats(const uint32_t *__restrict__ gbucket, uint32_t ng, uint32_t lane_max, uint32_t *__restrict__ stats,
                              uint32_t *__restrict__ big, const uint64_t *__restrict__ gkey0, const uint32_t *__restrict__ gord,
                              const uint32_t *__restrict__ gfirst, uint32_t stride, uint64_t *__restrict__ samp) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = 0;
  if (g < ng) {
    n = gbucket[g + 1] - gbucket[g];
    if (n > lane_max) big[atomicAdd(&stats[1], 1u)] = g;
    if (g % stride == 0) samp[g / stride] = gkey0[gord[g]];
    if (g == ng - 1) stats[2] = gfirst[gord[g]];
  }
  for (int o = 32; o; o >>= 1) n = max(n, (uint32_t)__shfl_xor((int)n, o, 64));
  // (one atomic per wavefront on ONE address is served one at a time, ~12 ns each: only while it can still raise the maximum)
  if ((threadIdx.x & 63) == 0 && n > *(volatile uint32_t *)&stats[0]) atomicMax(&stats[0], n);
}

// OR of all keys / of all negated positions' complements: the radix sorts only visit the bits that can differ
__global__ void k_key_bits(const uint64_t *__restrict__ key0, const uint64_t *__restrict__ key1, const uint32_t *__restrict__ npos, uint32_t nr,
                           unsigned long long *__restrict__ out) {
  unsigned long long k = 0, p = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nr; i += gridDim.x * blockDim.x) k |= key0[i] | key1[i], p |= (uint32_t)~npos[i];
  for (int o = 32; o; o >>= 1) k |= __shfl_xor(k, o, 64), p |= __shfl_xor(p, o, 64);
  if ((threadIdx.x & 63) == 0) {
    if (k) atomicOr(out, k);
    if (p) atomicOr(out + 1, p);
  }
}
static int bit_length(unsigned long long v) {
  int b = 0;
  while (b < 64 && (v >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

static void bucketize(PairRecs &R, unsigned flags, PairTables &out, DevicePairs *keep_dev, Tmp &tmp) {
  hipStream_t st = ctx().stream;
  size_t bytes = 0;
  const uint32_t nr = R.nr;
  DevBuf<uint64_t> &key0 = R.key0, &key1 = R.key1, &y0 = R.y0, &y1 = R.y1;
  DevBuf<uint8_t> &dir = R.dir;
  DevBuf<uint32_t> &npos = R.npos;

  // ---- bucket order: stable LSD sorts (position desc, key1, key0) carrying the record index --------------------
  DevBuf<uint32_t> idx(nr), perm_a(nr), perm_b(nr), k32s(nr);
  DevBuf<uint64_t> kg(nr), kgs(nr);
  int kbits = 64, pbits = 32;   // bits the keys / the positions occupy (hash << 8 | span is 40 bits at k = 16; positions < 64 k: 16)
  if (nr >= (1u << 16)) {       // (a 50 us round trip: not worth it for small joins)
    DevBuf<unsigned long long> d_or(2);
    PGX_HIP(hipMemsetAsync(d_or.p, 0, 2 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_key_bits, dim3(2048), dim3(256), 0, st, key0.p, key1.p, npos.p, nr, d_or.p);
    unsigned long long h_or[2] = {0, 0};
    d_or.download(h_or, 2);
    sync();
    kbits = bit_length(h_or[0]), pbits = std::min(32, bit_length(h_or[1]));
  }
  {
    if (flags & PAIRS_INSERTION_ORDER) {  // buckets keep their records in insertion order (shmr_map never sorts them)
      hipLaunchKernelGGL(k_iota, dim3(cdiv(nr, 256)), dim3(256), 0, st, perm_a.p, nr);
    } else {
      hipLaunchKernelGGL(k_iota, dim3(cdiv(nr, 256)), dim3(256), 0, st, idx.p, nr);
      bytes = 0;
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, npos.p, k32s.p, idx.p, perm_a.p, (int)nr, 0, pbits, st));
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, npos.p, k32s.p, idx.p, perm_a.p, (int)nr, 0, pbits, st));
    }
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, key1.p, perm_a.p, nr, kg.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kg.p, kgs.p, perm_a.p, perm_b.p, (int)nr, 0, kbits, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, kg.p, kgs.p, perm_a.p, perm_b.p, (int)nr, 0, kbits, st));
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, key0.p, perm_b.p, nr, kg.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, kg.p, kgs.p, perm_b.p, perm_a.p, (int)nr, 0, kbits, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, kg.p, kgs.p, perm_b.p, perm_a.p, (int)nr, 0, kbits, st));
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
  DevBuf<uint32_t> bfirst(nbk), gfirst(ng), glast(ng), gbucket((size_t)ng + 1);
  out.n_groups = ng, out.n_buckets = nbk;
  hipLaunchKernelGGL(k_group_first_bucket, dim3(cdiv(ng, 256)), dim3(256), 0, st, gstart.p, ng, bstart.p, nbk, gbucket.p);
  {
    DevBuf<uint32_t> blast(nbk);
    hipLaunchKernelGGL(k_bucket_minmax, dim3(cdiv(nbk, 256)), dim3(256), 0, st, perm_a.p, bstart.p, nbk, bfirst.p, blast.p);
    hipLaunchKernelGGL(k_group_minmax, dim3(cdiv(ng, 256)), dim3(256), 0, st, bfirst.p, blast.p, gbucket.p, ng, nbk, gfirst.p, glast.p);
  }

  // insertion orders the host replays the two khash levels in: groups by first insertion, buckets by (group, first insertion)
  DevBuf<uint32_t> gord(ng), bord(nbk), iota_g(ng), iota_b(nbk), gf_sorted(ng);
  DevBuf<uint64_t> bok(nbk), bok_sorted(nbk);
  {
    // (only the bits the keys can have are sorted: insertion indices < nr, group indices < ng)
    int fbits = 1, gbits = 1;
    while (fbits < 32 && ((uint64_t)1 << fbits) < (uint64_t)nr) ++fbits;
    while (gbits < 32 && ((uint64_t)1 << gbits) < (uint64_t)ng) ++gbits;
    hipLaunchKernelGGL(k_iota, dim3(cdiv(ng, 256)), dim3(256), 0, st, iota_g.p, ng);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, gfirst.p, gf_sorted.p, iota_g.p, gord.p, (int)ng, 0, fbits, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, gfirst.p, gf_sorted.p, iota_g.p, gord.p, (int)ng, 0, fbits, st));
    hipLaunchKernelGGL(k_iota, dim3(cdiv(nbk, 256)), dim3(256), 0, st, iota_b.p, nbk);
    hipLaunchKernelGGL(k_bucket_order_key, dim3(cdiv(nbk, 256)), dim3(256), 0, st, gbucket.p, ng, bfirst.p, nbk, fbits, bok.p);
    bytes = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, bok.p, bok_sorted.p, iota_b.p, bord.p, (int)nbk, 0, fbits + gbits, st));
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.get(bytes), bytes, bok.p, bok_sorted.p, iota_b.p, bord.p, (int)nbk, 0, fbits + gbits, st));
  }

  const bool trace = getenv("PGX_TRACE") != nullptr;
  double tj0 = 0;
  if (trace) {
    sync();
    tj0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  // the sorted records themselves are only read on the host by the host replay: a caller that keeps them on the device for the
  // device replay fetches them later if it has to fall back (pairs_fetch_records)
  const bool lazy = (flags & PAIRS_LAZY_RECORDS) && keep_dev;
  if (!lazy) {
    out.y0 = to_host(sy0, nr);
    out.dir = to_host(sdir, nr);
  }
  DevBuf<uint64_t> sy1((flags & PAIRS_Y1) ? nr : 0);
  if (flags & PAIRS_Y1) {
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nr, 256)), dim3(256), 0, st, y1.p, perm_a.p, nr, sy1.p);
    out.y1 = to_host(sy1, nr);
  }
  const bool dev_tables = (flags & PAIRS_DEV_TABLES) && (flags & PAIRS_ORD_TABLES) && keep_dev && lazy && !(flags & PAIRS_Y1);
  if (!dev_tables) {
    out.gord = to_host(gord, ng);
    out.bord = to_host(bord, nbk);
  }
  DevBuf<uint64_t> bkey1(nbk), gkey0(ng);
  hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(ng, 256)), dim3(256), 0, st, kgs.p, gstart.p, ng, gkey0.p);
  DevBuf<uint32_t> bn_o((flags & PAIRS_ORD_TABLES) ? nbk : 0);
  DevBuf<uint8_t> gtrail((flags & PAIRS_ORD_TABLES) ? ng : 0);
  if (dev_tables) {
    // the tables stay where they are: the visit order is built on the device (pgx_visit.hip).  What the host needs to decide
    // that -- the largest group, the groups a lane cannot replay, a sample of the first keys in first-insertion order -- comes
    // down with the one round trip that ends the join.
    hipLaunchKernelGGL(k_bucket_ord, dim3(cdiv(nbk, 256)), dim3(256), 0, st, bord.p, bstart.p, sk1.p, nbk, bkey1.p, bn_o.p);
    hipLaunchKernelGGL(k_group_trail, dim3(cdiv(ng, 256)), dim3(256), 0, st, gbucket.p, bord.p, bfirst.p, glast.p, ng, nbk, gtrail.p);
    PGX_HIP(hipMemcpyAsync(gbucket.p + ng, &nbk, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    const uint32_t nsamp = (ng + KEY_SAMPLE_STRIDE - 1) / KEY_SAMPLE_STRIDE;
    DevBuf<uint32_t> stats(4), big(ng);
    DevBuf<uint64_t> samp(nsamp);
    PGX_HIP(hipMemsetAsync(stats.p, 0, 4 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_group_stats, dim3(cdiv(ng, 256)), dim3(256), 0, st, gbucket.p, ng, VISIT_LANE_MAX, stats.p, big.p, gkey0.p, gord.p,
                       gfirst.p, KEY_SAMPLE_STRIDE, samp.p);
    uint32_t hst[4];
    keep_dev->key_sample.resize(nsamp);
    stats.download(hst, 4);
    samp.download(keep_dev->key_sample.data(), nsamp);
    sync();
    keep_dev->y0 = std::move(sy0), keep_dev->dir = std::move(sdir), keep_dev->bstart = std::move(bstart);
    keep_dev->n_rec = nr, keep_dev->n_buckets = nbk, keep_dev->valid = true;
    keep_dev->tables = true, keep_dev->n_groups = ng;
    keep_dev->gstart = std::move(gstart), keep_dev->gbucket = std::move(gbucket), keep_dev->gord = std::move(gord);
    keep_dev->gfirst = std::move(gfirst), keep_dev->glast = std::move(glast), keep_dev->bord = std::move(bord);
    keep_dev->bn_ord = std::move(bn_o), keep_dev->gkey0 = std::move(gkey0), keep_dev->bkey1_ord = std::move(bkey1);
    keep_dev->gtrail = std::move(gtrail), keep_dev->big_groups = std::move(big);
    keep_dev->max_group_buckets = hst[0], keep_dev->n_big_groups = hst[1], keep_dev->last_gfirst = hst[2];
    out.on_host = false;
    if (trace)
      fprintf(stderr, "[pgx]   join: tables left on the device (largest group %u buckets, %u groups beyond %u)\n", hst[0], hst[1], VISIT_LANE_MAX);
    return;
  }
  if (flags & PAIRS_ORD_TABLES) {
    hipLaunchKernelGGL(k_bucket_ord, dim3(cdiv(nbk, 256)), dim3(256), 0, st, bord.p, bstart.p, sk1.p, nbk, bkey1.p, bn_o.p);
    hipLaunchKernelGGL(k_group_trail, dim3(cdiv(ng, 256)), dim3(256), 0, st, gbucket.p, bord.p, bfirst.p, glast.p, ng, nbk, gtrail.p);
    out.bkey1_ord = to_host(bkey1, nbk);
    out.bn_ord = to_host(bn_o, nbk);
    out.gtrail = to_host(gtrail, ng);
  } else {
    hipLaunchKernelGGL(k_gather_u64, dim3(cdiv(nbk, 256)), dim3(256), 0, st, sk1.p, bstart.p, nbk, bkey1.p);
    out.bkey1 = to_host(bkey1, nbk);
    out.bfirst = to_host(bfirst, nbk);
  }
  out.gkey0 = to_host(gkey0, ng);
  out.bstart = to_host(bstart, (size_t)nbk + 1);
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
