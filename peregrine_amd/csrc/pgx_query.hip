// pgx_query.hip -- SURVEY.md 8(f) rows f3 and f4, both consumers of the shimmer-pair map that build_map
// (/root/reference/src/shmr_utils.c:295-404) builds and that pgx_pairs.hip produces on the GPU:
//   f4: the query helpers of /root/reference/src/shimmer4py.c:44-196 (build_shimmer_map4py, get_shimmers_for_read,
//       get_mmer_count, get_shimmer_hits) -- the map is built on the GPU once, lookups are binary searches on the host
//       over its sorted tables; get_shimmer_hits replays the one inner khash table it walks.
//   f3: shmr_map (/root/reference/src/shmr_map.c:48-161): the reference shimmers are chained and looked up on the GPU,
//       matching buckets are expanded to rows there; the host only formats the text.
#include <glob.h>

#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"
#include "pgx_khash.h"

namespace pgx {
namespace {
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------
// shared: shimmer / count files of a prefix (shimmer4py.c:94-116, shmr_map.c:285-345), read lengths by rid
// ---------------------------------------------------------------------------------------------------------
template <typename T>
void read_counted(const std::string &pattern, std::vector<T> &out) {
  glob_t g;
  memset(&g, 0, sizeof(g));
  if (glob(pattern.c_str(), 0, nullptr, &g) == 0) {  // name-sorted, like wordexp in the reference
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      std::vector<uint8_t> buf;
      if (!read_file(g.gl_pathv[i], buf) || buf.size() < 8) {
        const std::string bad = g.gl_pathv[i];
        globfree(&g);
        PGX_REQUIRE(false, PGX_EIO, "file '%s' open error", bad.c_str());
      }
      uint64_t n;
      memcpy(&n, buf.data(), 8);
      if (n > (buf.size() - 8) / sizeof(T)) {
        const std::string bad = g.gl_pathv[i];
        globfree(&g);
        PGX_REQUIRE(false, PGX_EIO, "file '%s' is truncated: header says %llu entries, %zu bytes follow", bad.c_str(),
                    (unsigned long long)n, buf.size() - 8);
      }
      const size_t o = out.size();
      out.resize(o + n);
      if (n) memcpy(out.data() + o, buf.data() + 8, n * sizeof(T));
    }
  }
  globfree(&g);
}

void rlen_by_rid_of(const char *seqdb_prefix, std::vector<uint32_t> &by_rid) {
  std::vector<uint32_t> rid, rlen;
  std::vector<uint64_t> roff;
  const std::string path = std::string(seqdb_prefix) + ".idx";
  PGX_REQUIRE(load_idx(path.c_str(), rid, rlen, roff) == 0, PGX_EIO, "cannot read '%s'", path.c_str());
  uint32_t mx = 0;
  for (uint32_t r : rid) mx = std::max(mx, r);
  by_rid.assign(rid.empty() ? 0 : (size_t)mx + 1, 0);
  for (size_t i = 0; i < rid.size(); ++i) by_rid[rid[i]] = rlen[i];
}

void max_rid_check(const pgx_mm128 *mm, size_t n, size_t n_rid, const char *what) {
  for (size_t i = 0; i < n; ++i)
    PGX_REQUIRE((mm[i].y >> 32) < n_rid, PGX_EARG, "%s: shimmer of read %llu, but the index knows %zu reads", what,
                (unsigned long long)(mm[i].y >> 32), n_rid);
}

// ---------------------------------------------------------------------------------------------------------
// f4
// ---------------------------------------------------------------------------------------------------------
struct ShimmerMap {
  PairTables pt;                       // records position-descending inside a bucket, with y1 and the count table
  std::vector<size_t> first, count;    // get_ridmm: first occurrence / number of occurrences of every rid
  ScratchTable scratch;
};
constexpr uint64_t MAGIC = 0x70677873686D6170ULL;  // stored in py_mmer_t.rlmap so that foreign pointers are refused
struct Handle {
  uint64_t magic;
  ShimmerMap map;
};

Handle *handle_of(py_mmer_t *m) {
  if (!m || !m->mmer0_map) return nullptr;
  Handle *h = (Handle *)m->mmer0_map;
  return h->magic == MAGIC ? h : nullptr;
}

// ---------------------------------------------------------------------------------------------------------
// f3 kernels
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pos28(uint64_t y) { return (uint32_t)((y >> 1) & 0xFFFFFFF); }
__device__ __forceinline__ uint32_t pos_of(uint64_t y) { return (uint32_t)((y & 0xFFFFFFFFu) >> 1); }

__device__ __forceinline__ bool find_u64(const uint64_t *__restrict__ a, uint32_t lo, uint32_t hi, uint64_t key, uint32_t *at) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1;
    else hi = mid;
  }
  *at = lo;
  return false;
}

// which reference shimmers take part in the chain (shmr_map.c:84-98): the first one that is a key0 of the map, then every
// later one whose hash the reads know with a multiplicity inside [lower, upper]
__global__ void k_ref_flags(const pgx_mm128 *__restrict__ ref, uint32_t n, const uint64_t *__restrict__ gkey0, uint32_t ng,
                            const uint64_t *__restrict__ umer, const uint32_t *__restrict__ ucnt, uint32_t nu, uint32_t lower,
                            uint32_t upper, uint8_t *__restrict__ keep, uint32_t *__restrict__ cnt, uint32_t *__restrict__ first_key0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t at;
  find_u64(gkey0, 0, ng, ref[i].x, &at);
  if (at < ng && gkey0[at] == ref[i].x && i < *(volatile uint32_t *)first_key0) atomicMin(first_key0, i);  // (guarded: one hot address)
  find_u64(umer, 0, nu, ref[i].x >> 8, &at);
  const bool known = at < nu && umer[at] == ref[i].x >> 8;
  const uint32_t c = known ? ucnt[at] : 0;
  cnt[i] = c;
  keep[i] = known && c >= lower && c <= upper;
}
__global__ void k_ref_chain_in(const uint8_t *__restrict__ keep, uint32_t n, const uint32_t *__restrict__ first_key0,
                               int32_t *__restrict__ v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = *first_key0;
  v[i] = (i == s || (i > s && keep[i])) ? (int32_t)i : -1;
}
struct MaxOp {
  __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; }
};
// the bucket a chained pair (previous kept shimmer, shimmer i) hits, if any (shmr_map.c:100-121)
__global__ void k_ref_hits(const pgx_mm128 *__restrict__ ref, uint32_t n, const int32_t *__restrict__ chain,
                           const uint32_t *__restrict__ first_key0, const uint64_t *__restrict__ gkey0, uint32_t ng,
                           const uint32_t *__restrict__ gbucket, const uint64_t *__restrict__ bkey1,
                           const uint32_t *__restrict__ bstart, uint32_t *__restrict__ bucket, uint32_t *__restrict__ rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t nrow = 0, bk = 0;
  const uint32_t s = *first_key0;
  if (i > s && chain[i] == (int32_t)i) {  // i is chained and not the anchor
    const pgx_mm128 a = ref[chain[i - 1]], b = ref[i];
    if ((a.y >> 32) == (b.y >> 32) && (uint64_t)pos28(b.y) - (uint64_t)pos28(a.y) >= 100ull) {  // 64-bit difference (:118)
      uint32_t g;
      find_u64(gkey0, 0, ng, a.x, &g);
      if (g < ng && gkey0[g] == a.x) {
        uint32_t at;
        find_u64(bkey1, gbucket[g], gbucket[g + 1], b.x, &at);
        if (at < gbucket[g + 1] && bkey1[at] == b.x) bk = at, nrow = bstart[at + 1] - bstart[at];
      }
    }
  }
  bucket[i] = bk, rows[i] = nrow;
}
struct MapRow {
  uint32_t ref_id, ref_bgn, ref_end, read_id, read_bgn, read_end, dir, mcount0, mcount1;
};
__global__ void k_ref_rows(const pgx_mm128 *__restrict__ ref, uint32_t n, const int32_t *__restrict__ chain,
                           const uint32_t *__restrict__ bucket, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ off,
                           const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ bstart, const uint64_t *__restrict__ y0,
                           const uint64_t *__restrict__ y1, const uint8_t *__restrict__ dir, MapRow *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || rows[i] == 0) return;
  const int32_t p = chain[i - 1];
  const pgx_mm128 a = ref[p], b = ref[i];
  const uint32_t r0 = bstart[bucket[i]];
  for (uint32_t j = 0; j < rows[i]; ++j)
    out[off[i] + j] = MapRow{(uint32_t)(a.y >> 32), pos_of(a.y), pos_of(b.y), (uint32_t)(y0[r0 + j] >> 32), pos_of(y0[r0 + j]),
                             pos_of(y1[r0 + j]), dir[r0 + j], cnt[p], cnt[i]};
}

template <typename T>
void up(DevBuf<T> &d, const HostArray<T> &h) {
  d.alloc(h.size());
  d.upload(h.data(), h.size());
}

inline char *put_u32(char *w, uint32_t v) {  // decimal, no padding
  char tmp[10];
  int n = 0;
  do tmp[n++] = (char)('0' + v % 10), v /= 10;
  while (v);
  while (n) *w++ = tmp[--n];
  return w;
}

void run_map(const pgx_mm128 *ref, size_t n_ref, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
             const uint32_t *rlen_by_rid, uint32_t n_rid, const pgx_map_params *p, std::string &text, uint64_t &n_lines) {
  text.clear();
  n_lines = 0;
  PGX_REQUIRE(p && p->total_chunk > 0 && p->mychunk > 0 && p->mychunk <= p->total_chunk, PGX_EARG,
              "need 0 < mychunk <= total_chunk (shmr_map.c:247-248)");
  PGX_REQUIRE(n_ref > 0, PGX_EARG, "no reference shimmers (shmr_map.c:83)");
  PGX_REQUIRE(n_ref < (1ULL << 31), PGX_EARG, "reference shimmer list too long");
  max_rid_check(mmers, n_mm, n_rid, "pgx_map");
  hipStream_t st = ctx().stream;
  DevBuf<uint32_t> d_rlen(n_rid);
  d_rlen.upload(rlen_by_rid, n_rid);
  PairTables pt;
  dev_build_pairs(d_rlen.p, mmers, n_mm, counts, n_counts,
                  PairParams{(uint32_t)p->total_chunk, (uint32_t)p->mychunk, (uint32_t)p->mc_lower, (uint32_t)p->mc_upper}, pt,
                  PAIRS_Y1 | PAIRS_INSERTION_ORDER | PAIRS_COUNTS);
  if (pt.n_rec == 0) return;
  KernelTimer tm("map", n_ref);
  const uint32_t n = (uint32_t)n_ref, ng = (uint32_t)pt.gkey0.size(), nu = (uint32_t)pt.umer.size();
  DevBuf<pgx_mm128> d_ref(n);
  d_ref.upload(ref, n);
  DevBuf<uint64_t> gkey0, bkey1, umer, y0, y1;
  DevBuf<uint32_t> gbucket, bstart, ucnt;
  DevBuf<uint8_t> dir;
  up(gkey0, pt.gkey0), up(bkey1, pt.bkey1), up(umer, pt.umer), up(y0, pt.y0), up(y1, pt.y1);
  up(gbucket, pt.gbucket), up(bstart, pt.bstart), up(ucnt, pt.ucnt), up(dir, pt.dir);
  DevBuf<uint8_t> keep(n);
  DevBuf<uint32_t> cnt(n), first(1), bucket(n), rows(n), off((size_t)n + 1);
  DevBuf<int32_t> chain_in(n), chain(n);
  const uint32_t none = 0xFFFFFFFFu;
  first.upload(&none, 1);
  hipLaunchKernelGGL(k_ref_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, d_ref.p, n, gkey0.p, ng, umer.p, ucnt.p, nu,
                     (uint32_t)p->mc_lower, (uint32_t)p->mc_upper, keep.p, cnt.p, first.p);
  hipLaunchKernelGGL(k_ref_chain_in, dim3(cdiv(n, 256)), dim3(256), 0, st, keep.p, n, first.p, chain_in.p);
  size_t bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(nullptr, bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  DevBuf<uint8_t> tmp(bytes);
  PGX_HIP(hipcub::DeviceScan::InclusiveScan(tmp.p, bytes, chain_in.p, chain.p, MaxOp(), (int)n, st));
  hipLaunchKernelGGL(k_ref_hits, dim3(cdiv(n, 256)), dim3(256), 0, st, d_ref.p, n, chain.p, first.p, gkey0.p, ng, gbucket.p, bkey1.p,
                     bstart.p, bucket.p, rows.p);
  PGX_HIP(hipMemsetAsync(off.p, 0, sizeof(uint32_t), st));
  bytes = 0;
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, rows.p, off.p + 1, (int)n, st));
  DevBuf<uint8_t> tmp2(bytes);
  PGX_HIP(hipcub::DeviceScan::InclusiveSum(tmp2.p, bytes, rows.p, off.p + 1, (int)n, st));
  uint32_t total = 0;
  PGX_HIP(hipMemcpyAsync(&total, off.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  sync();
  if (total == 0) return;
  DevBuf<MapRow> d_rows(total);
  hipLaunchKernelGGL(k_ref_rows, dim3(cdiv(n, 256)), dim3(256), 0, st, d_ref.p, n, chain.p, bucket.p, rows.p, off.p, cnt.p, bstart.p,
                     y0.p, y1.p, dir.p, d_rows.p);
  HostArray<MapRow> h(total);
  d_rows.download(h.data(), total);
  sync();
  // "%u %u %u %u %u %u %d %u %u\n" (shmr_map.c:152-153)
  text.resize((size_t)total * 100);
  char *w = &text[0];
  for (uint32_t i = 0; i < total; ++i) {
    const MapRow &r = h[i];
    const uint32_t f[9] = {r.ref_id, r.ref_bgn, r.ref_end, r.read_id, r.read_bgn, r.read_end, r.dir, r.mcount0, r.mcount1};
    for (int k = 0; k < 9; ++k) {
      w = put_u32(w, f[k]);
      *w++ = k == 8 ? '\n' : ' ';
    }
  }
  text.resize((size_t)(w - text.data()));
  n_lines = total;
}

}  // namespace
}  // namespace pgx

using namespace pgx;

extern "C" {

// ---- f3 ------------------------------------------------------------------------------------------------------
int pgx_map(const pgx_mm128 *ref_mmers, size_t n_ref, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
            size_t n_counts, const uint32_t *rlen_by_rid, uint32_t n_rid, const pgx_map_params *p, char **text, size_t *text_len,
            uint64_t *n_lines) {
  try {
    require_ready();
    PGX_REQUIRE(text && text_len && (n_ref == 0 || ref_mmers) && (n_mm == 0 || mmers) && (n_counts == 0 || counts) &&
                    (n_rid == 0 || rlen_by_rid),
                PGX_EARG, "pgx_map: null argument");
    std::string s;
    uint64_t nl = 0;
    run_map(ref_mmers, n_ref, mmers, n_mm, counts, n_counts, rlen_by_rid, n_rid, p, s, nl);
    timing_flush();
    char *out = (char *)malloc(s.size() + 1);
    if (!out) throw std::bad_alloc();
    memcpy(out, s.data(), s.size());
    out[s.size()] = 0;
    *text = out, *text_len = s.size();
    if (n_lines) *n_lines = nl;
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_map_chunk(const char *refdb_prefix, const char *ref_shimmer_prefix, const char *seqdb_prefix, const char *shimmer_prefix,
                  const pgx_map_params *p, char **text, size_t *text_len, uint64_t *n_lines) {
  try {
    require_ready();
    PGX_REQUIRE(ref_shimmer_prefix && seqdb_prefix && shimmer_prefix, PGX_EARG, "pgx_map_chunk: null argument");
    (void)refdb_prefix;  // the reference maps the two seqdb files but never reads them (shmr_map.c:60-78)
    std::vector<pgx_mm128> ref, mm;
    std::vector<pgx_mm_count> mc;
    std::vector<uint32_t> rl;
    read_counted(std::string(ref_shimmer_prefix) + "-[0-9]*-of-[0-9]*.dat", ref);
    rlen_by_rid_of(seqdb_prefix, rl);
    read_counted(std::string(shimmer_prefix) + "-[0-9]*-of-[0-9]*.dat", mm);
    read_counted(std::string(shimmer_prefix) + "-MC-[0-9]*-of-[0-9]*.dat", mc);
    return pgx_map(ref.data(), ref.size(), mm.data(), mm.size(), mc.data(), mc.size(), rl.data(), (uint32_t)rl.size(), p, text,
                   text_len, n_lines);
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
}

// ---- f4 ------------------------------------------------------------------------------------------------------
// The reference returns void and exit(1)s / asserts on bad input; here a failure leaves every field of *py_mmer NULL and
// the message in pgx_last_error() (and on stderr).
void build_shimmer_map4py(py_mmer_t *py_mmer, char *seqdb_prefix, char *shimmer_prefix, uint32_t mychunk, uint32_t total_chunk,
                          uint32_t lowerbound, uint32_t upperbound) {
  if (!py_mmer) return;
  memset(py_mmer, 0, sizeof(*py_mmer));
  Handle *h = nullptr;
  mm128_v *mv = nullptr;
  try {
    require_ready();
    PGX_REQUIRE(total_chunk > 0 && mychunk > 0 && mychunk <= total_chunk, PGX_EARG,
                "need 0 < mychunk <= total_chunk (shimmer4py.c:67-68)");
    const char *sp = seqdb_prefix ? seqdb_prefix : "seq_dataset";  // the defaults of shimmer4py.c:70-78
    const char *lp = shimmer_prefix ? shimmer_prefix : "shimmer-L2";
    std::vector<uint32_t> rl;
    rlen_by_rid_of(sp, rl);
    std::vector<pgx_mm128> mm;
    std::vector<pgx_mm_count> mc;
    read_counted(std::string(lp) + "-[0-9]*-of-[0-9]*.dat", mm);
    read_counted(std::string(lp) + "-MC-[0-9]*-of-[0-9]*.dat", mc);
    max_rid_check(mm.data(), mm.size(), rl.size(), "build_shimmer_map4py");
    h = new Handle{MAGIC, {}};
    DevBuf<uint32_t> d_rlen(rl.size());
    d_rlen.upload(rl.data(), rl.size());
    dev_build_pairs(d_rlen.p, mm.data(), mm.size(), mc.data(), mc.size(), PairParams{total_chunk, mychunk, lowerbound, upperbound},
                    h->map.pt, PAIRS_Y1 | PAIRS_COUNTS);
    sync();
    timing_flush();
    // get_ridmm (shmr_utils.c:415-443): a read's list starts at its first occurrence and is as long as its occurrences
    h->map.first.assign(rl.size(), 0), h->map.count.assign(rl.size(), 0);
    for (size_t i = 0; i < mm.size(); ++i) {
      const uint32_t rid = (uint32_t)(mm[i].y >> 32);
      if (h->map.count[rid]++ == 0) h->map.first[rid] = i;
    }
    mv = (mm128_v *)malloc(sizeof(mm128_v));  // caller-visible kvec, malloc'd like the reference's (:88-91)
    if (!mv) throw std::bad_alloc();
    mv->n = mv->m = mm.size();
    mv->a = (pgx_mm128 *)malloc((mm.size() ? mm.size() : 1) * sizeof(pgx_mm128));
    if (!mv->a) throw std::bad_alloc();
    if (!mm.empty()) memcpy(mv->a, mm.data(), mm.size() * sizeof(pgx_mm128));
    py_mmer->mmers = mv;
    py_mmer->mmer0_map = h, py_mmer->rlmap = h, py_mmer->mcmap = h, py_mmer->ridmm = h;
    return;
  } catch (const Fail &f) {
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
  }
  fprintf(stderr, "pgx: build_shimmer_map4py failed: %s\n", pgx_last_error());
  delete h;
  if (mv) free(mv->a);
  free(mv);
}

void pgx_shimmer_map_free(py_mmer_t *py_mmer) {
  if (!py_mmer) return;
  delete handle_of(py_mmer);
  if (py_mmer->mmers) free(py_mmer->mmers->a);
  free(py_mmer->mmers);
  memset(py_mmer, 0, sizeof(*py_mmer));
}

void get_shimmers_for_read(mm128_v *mmer, py_mmer_t *py_mmer, uint32_t rid) {
  if (!mmer) return;
  mmer->n = mmer->m = 0, mmer->a = nullptr;  // an unknown read gives the empty vector (shimmer4py.c:141-142)
  Handle *h = handle_of(py_mmer);
  if (!h || rid >= h->map.count.size() || h->map.count[rid] == 0) return;
  mmer->n = mmer->m = h->map.count[rid];
  mmer->a = py_mmer->mmers->a + h->map.first[rid];  // a view into the map's list, not owned by the caller
}

uint32_t get_mmer_count(py_mmer_t *py_mmer, uint64_t mhash) {
  Handle *h = handle_of(py_mmer);
  if (!h) return 0;
  const HostArray<uint64_t> &u = h->map.pt.umer;
  const uint64_t *e = std::lower_bound(u.begin(), u.end(), mhash);
  return (e != u.end() && *e == mhash) ? h->map.pt.ucnt[(size_t)(e - u.begin())] : 0;
}

void get_shimmer_hits(mp256_v *out, py_mmer_t *py_mmer, uint64_t mhash0, uint32_t span) {
  Handle *h = handle_of(py_mmer);
  if (!h || !out) return;
  const PairTables &pt = h->map.pt;
  const uint64_t key0 = mhash0 << 8 | span;
  const uint64_t *e = std::lower_bound(pt.gkey0.begin(), pt.gkey0.end(), key0);
  if (e == pt.gkey0.end() || *e != key0) return;
  const size_t g = (size_t)(e - pt.gkey0.begin());
  // the key0's inner table in ascending slot order (shimmer4py.c:180-181): replayed from the buckets' first-insertion order
  const uint32_t b0 = pt.gbucket[g], b1 = pt.gbucket[g + 1];
  const uint32_t *bord = pt.bord.data() + b0;
  ScratchTable &in = h->map.scratch;
  bool ab;
  in.reset();
  for (uint32_t i = 0; i < b1 - b0; ++i) in.put(pt.bkey1[bord[i]], bord[i], &ab);
  if (pt.bfirst[bord[b1 - b0 - 1]] < pt.glast[g]) in.put(pt.bkey1[bord[0]], 0, &ab);  // trailing repeat put (khash.h:298-306)
  for (uint32_t s1 = 0; s1 < in.nb; ++s1) {
    if (!in.used[s1]) continue;
    const uint32_t b = in.ids[s1];
    for (uint32_t r = pt.bstart[b]; r < pt.bstart[b + 1]; ++r) {  // already position-descending, stable (:186)
      if (out->n == out->m) {  // kv_push growth (kvec.h:78-84)
        out->m = out->m ? out->m << 1 : 2;
        out->a = (mp256_t *)realloc(out->a, sizeof(mp256_t) * out->m);
        if (!out->a) {
          out->n = out->m = 0;
          return;
        }
      }
      mp256_t &o = out->a[out->n++];
      memset(&o, 0, sizeof(o));
      o.x0 = key0, o.x1 = pt.bkey1[b], o.y0 = pt.y0[r], o.y1 = pt.y1[r], o.direction = pt.dir[r];
    }
  }
}

}  // extern "C"
