// pgx_dedup.hip -- SURVEY.md 8(f) row f2: what /root/reference/src/shmr_dedup.c:32-101 does (cat ovlp*.dat | shmr_dedup):
// the first record of every read pair wins, its coordinates are mapped to FALCON's preads.ovl text line.
// GPU: first-wins flags by a stable radix sort of (pair, stream index) + the coordinate transform of the kept records;
// host: text formatting only ("%09d %09d %d %0.1f %u %d %d %u %u %d %d %u %s\n", shmr_dedup.c:91-99).
// Note: on an EMPTY stream the reference formats one record from uninitialised stack memory (its while(!feof) loop runs
// once); this implementation writes nothing.
#include <hipcub/hipcub.hpp>

#include "pgx_internal.h"

namespace pgx {
namespace {
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

struct Row {
  uint32_t rid0, rid1;
  int32_t m_size, dist;
  uint32_t a_bgn, a_end, rlen0, strand, b_bgn, b_end, rlen1, type;
};

__global__ void k_pair_keys(const pgx_ovlp *__restrict__ in, uint32_t n, uint64_t *__restrict__ key, uint32_t *__restrict__ idx) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r0 = (uint32_t)(in[i].y0 >> 32), r1 = (uint32_t)(in[i].y1 >> 32);
  key[i] = r0 < r1 ? ((uint64_t)r0 << 32 | r1) : ((uint64_t)r1 << 32 | r0);
  idx[i] = i;
}
__global__ void k_first_flags(const uint64_t *__restrict__ skey, const uint32_t *__restrict__ sidx, uint32_t n,
                              uint8_t *__restrict__ keep) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keep[sidx[i]] = (i == 0) || skey[i] != skey[i - 1];  // stable sort: the first of a run is the earliest in the stream
}
// coordinate transform of shmr_dedup.c:44-89 (unsigned 32-bit arithmetic exactly as written there)
__global__ void k_rows(const pgx_ovlp *__restrict__ in, const uint32_t *__restrict__ sel, uint32_t m, Row *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const pgx_ovlp o = in[sel[j]];
  const uint32_t pos0 = (uint32_t)((o.y0 & 0xFFFFFFFFu) >> 1) + 1, pos1 = (uint32_t)((o.y1 & 0xFFFFFFFFu) >> 1) + 1;
  const uint32_t rlen0 = o.rl0, rlen1 = o.rl1;
  int32_t q_bgn = o.match.q_bgn, q_end = o.match.q_end, t_bgn = o.match.t_bgn, t_end = o.match.t_end;
  q_bgn -= t_bgn;
  t_bgn = 0;
  uint32_t a_bgn, a_end, b_bgn, b_end;
  if (o.strand0 == 0) {
    a_bgn = (uint32_t)((int32_t)(pos0 - pos1) + q_bgn);
    a_end = (uint32_t)((int32_t)(pos0 - pos1) + q_end);
  } else {
    a_bgn = (uint32_t)((int32_t)rlen0 - (int32_t)(pos0 - pos1) - q_end);
    a_end = (uint32_t)((int32_t)rlen0 - (int32_t)(pos0 - pos1) - q_bgn);
  }
  a_end = a_end >= rlen0 ? rlen0 : a_end;  // the "< 0" fixes of the reference are no-ops on unsigned values
  if (o.strand1 == 0) {
    b_bgn = (uint32_t)t_bgn;
    b_end = (uint32_t)t_end;
  } else {
    b_bgn = (uint32_t)((int32_t)rlen1 - t_end);
    b_end = (uint32_t)((int32_t)rlen1 - t_bgn);
  }
  b_end = b_end >= rlen1 ? rlen1 : b_end;
  Row r;
  r.rid0 = (uint32_t)(o.y0 >> 32), r.rid1 = (uint32_t)(o.y1 >> 32);
  r.m_size = o.match.m_size, r.dist = o.match.dist;
  r.a_bgn = a_bgn, r.a_end = a_end, r.rlen0 = rlen0;
  r.strand = o.strand0 == 0 ? o.strand1 : 1u - o.strand1;
  r.b_bgn = b_bgn, r.b_end = b_end, r.rlen1 = rlen1, r.type = o.ovlp_type;
  out[j] = r;
}
}  // namespace
}  // namespace pgx

using namespace pgx;

extern "C" int pgx_dedup(const pgx_ovlp *recs, size_t n, char **text, size_t *text_len, uint64_t *n_unique) {
  try {
    require_ready();
    PGX_REQUIRE(text && text_len && (n == 0 || recs), PGX_EARG, "pgx_dedup: null argument");
    PGX_REQUIRE(n < (1ULL << 31), PGX_EARG, "too many records for one call");
    std::string out;
    uint64_t m = 0;
    if (n) {
      hipStream_t st = ctx().stream;
      KernelTimer tm("dedup", n);
      pgx_ovlp *d_in = ws<pgx_ovlp>("dd.in", n);
      uint64_t *key = ws<uint64_t>("dd.key", n), *skey = ws<uint64_t>("dd.skey", n);
      uint32_t *idx = ws<uint32_t>("dd.idx", n), *sidx = ws<uint32_t>("dd.sidx", n), *sel = ws<uint32_t>("dd.sel", n);
      uint8_t *keep = ws<uint8_t>("dd.keep", n);
      uint32_t *d_m = ws<uint32_t>("dd.m", 1);
      PGX_HIP(hipMemcpyAsync(d_in, recs, n * sizeof(pgx_ovlp), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_pair_keys, dim3(cdiv(n, 256)), dim3(256), 0, st, d_in, (uint32_t)n, key, idx);
      size_t bytes = 0;
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, key, skey, idx, sidx, (int)n, 0, 64, st));
      void *tmp = ws_raw("dd.tmp", bytes);
      PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, key, skey, idx, sidx, (int)n, 0, 64, st));
      hipLaunchKernelGGL(k_first_flags, dim3(cdiv(n, 256)), dim3(256), 0, st, skey, sidx, (uint32_t)n, keep);
      bytes = 0;
      hipcub::CountingInputIterator<uint32_t, ptrdiff_t> it(0);
      PGX_HIP(hipcub::DeviceSelect::Flagged(nullptr, bytes, it, keep, sel, d_m, (int)n, st));
      tmp = ws_raw("dd.tmp", bytes);
      PGX_HIP(hipcub::DeviceSelect::Flagged(tmp, bytes, it, keep, sel, d_m, (int)n, st));
      uint32_t mm = 0;
      PGX_HIP(hipMemcpyAsync(&mm, d_m, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      pgx::sync();
      m = mm;
      Row *d_rows = ws<Row>("dd.rows", m);
      hipLaunchKernelGGL(k_rows, dim3(cdiv(m, 256)), dim3(256), 0, st, d_in, sel, (uint32_t)m, d_rows);
      std::vector<Row> rows(m);
      PGX_HIP(hipMemcpyAsync(rows.data(), d_rows, m * sizeof(Row), hipMemcpyDeviceToHost, st));
      pgx::sync();
      out.reserve(m * 96);
      char line[256];
      for (const Row &r : rows) {
        const double err_est = 100.0 - 100.0 * (double)r.dist / (double)r.m_size;
        const int len = snprintf(line, sizeof(line), "%09d %09d %d %0.1f %u %d %d %u %u %d %d %u %s\n", (int)r.rid0, (int)r.rid1,
                                 -r.m_size, err_est, 0u, (int)r.a_bgn, (int)r.a_end, r.rlen0, r.strand, (int)r.b_bgn,
                                 (int)r.b_end, r.rlen1, r.type == 0 ? "overlap" : (r.type == 1 ? "contains" : "contained"));
        out.append(line, (size_t)len);
      }
    }
    *text = (char *)malloc(out.size() + 1);
    memcpy(*text, out.data(), out.size());
    (*text)[out.size()] = 0;
    *text_len = out.size();
    if (n_unique) *n_unique = m;
    timing_flush();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}
