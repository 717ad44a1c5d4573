// pgx_pack.hip -- the 2-bit packs of a read database that the alignment kernels read (k_align_ph<8, u16, packed>, pgx_align.hip).
//
// Word w of stream s (s = 0: low nibbles = the read as stored; 1: high nibbles = its reverse complement, stored forward:
// /root/reference/src/shmr_utils.c:44-51) holds the codes of seqdb bytes 16 w .. 16 w + 15, base i in bits 2i..2i+1.  A byte whose
// nibbles are not both one-hot codes (an ambiguous base, or anything the reference's encoder never writes) has no 2-bit code: it marks
// its read in nflag[] and candidates that touch such a read take the byte-wise kernels, which compare the nibbles as DWmatch.c:136-137 does.
//
// The packs are a CACHE of the immutable seqdb bytes, kept with the pgx_seqdb for as long as it lives (round 4; round 3 rebuilt them
// in every overlap stage): built by the first large alignment launch -- 1.6 ms per 4.5 GB, at the memory roofline -- and reused by
// every later stage on the same database, on every rank of a multi-GPU job (whose seqdb replica grows with the job while its chunk
// does not).  If their HBM (seqdb / 2) cannot be had, the alignment launches stay on the byte-wise kernels.
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "pgx_internal.h"

namespace pgx {
namespace {

__device__ __forceinline__ uint32_t pack4(uint32_t n) {   // four one-hot nibbles (one per byte) -> 8 bits
  const uint32_t c = ((n >> 1) & 0x07070707u) - ((n >> 3) & 0x01010101u);   // 1,2,4,8 -> 0,1,2,3 per byte; the masks keep the shifts from
                                                                           // leaking the next byte's low bits in (no borrow then: 4 - 1, x - 0)
  return (c | (c >> 6) | (c >> 12) | (c >> 18)) & 0xFFu;
}
// bytes (one nibble each, 0..15) that are NOT exactly one of 1, 2, 4, 8: zero, or more than one bit set
__device__ __forceinline__ uint32_t not_onehot(uint32_t x) {
  const uint32_t t = x - 0x01010101u;                 // (no borrow across bytes unless a byte is zero -- and then that byte is flagged anyway)
  return ((t & ~x & 0x80808080u) | ((x & t) & 0x0F0F0F0Fu));
}
__global__ __launch_bounds__(256) void k_pack2(const uint4 *__restrict__ seq, size_t nwords, uint32_t *__restrict__ p0, uint32_t *__restrict__ p1,
                                               const uint64_t *__restrict__ roff_sorted, const uint32_t *__restrict__ rid_sorted, uint32_t nreads,
                                               size_t nbytes, uint32_t *__restrict__ nflag) {
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = seq[w];
    const uint32_t M = 0x0F0F0F0Fu;
    const uint32_t l[4] = {v.x & M, v.y & M, v.z & M, v.w & M};
    const uint32_t h[4] = {(v.x >> 4) & M, (v.y >> 4) & M, (v.z >> 4) & M, (v.w >> 4) & M};
    p0[w] = pack4(l[0]) | (pack4(l[1]) << 8) | (pack4(l[2]) << 16) | (pack4(l[3]) << 24);
    p1[w] = pack4(h[0]) | (pack4(h[1]) << 8) | (pack4(h[2]) << 16) | (pack4(h[3]) << 24);
    uint32_t any = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) any |= not_onehot(l[d]) | not_onehot(h[d]);
    if (any && w * 16 < nbytes) {   // (zero bytes past the end of the database are padding)
      for (int j = 0; j < 16; ++j) {
        const size_t pos = w * 16 + j;
        const uint32_t ln = (l[j >> 2] >> (8 * (j & 3))) & 0xFu, hn = (h[j >> 2] >> (8 * (j & 3))) & 0xFu;
        if (pos >= nbytes || (__builtin_popcount(ln) == 1 && __builtin_popcount(hn) == 1)) continue;
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

}  // namespace

size_t seq_pack_stride(const pgx_seqdb *db) { return (((db->nbytes + 1024) / 16) + 3) & ~(size_t)3; }   // dwords per stream, a multiple of 4
bool seq_packs_valid(const pgx_seqdb *db) { return db->packs_built && db->d_pack.p != nullptr; }

// [P0 | P1], seq_pack_stride dwords each; nullptr: the packs' HBM could not be allocated (the caller uses the byte-wise kernels)
const uint32_t *seq_packs(const pgx_seqdb *db) {
  if (seq_packs_valid(db)) return db->d_pack.p;
  if (db->packs_failed) return nullptr;
  hipStream_t st = ctx().stream;
  const size_t nwords = (db->nbytes + 1024) / 16;   // (the seqdb buffer carries 1 KiB of zero padding)
  const size_t stride = seq_pack_stride(db);
  try {
    MemTag mem_tag("seqdb.packs_2bit");
    if (db->d_pack.n < 2 * stride + 64) db->d_pack.alloc(2 * stride + 64);
  } catch (const Fail &) {
    db->packs_failed = true;
    (void)hipGetLastError();   // (the failed allocation's sticky error)
    if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] align: no HBM for the 2-bit packs (%.1f GB): the byte-wise kernels take every launch\n", (2 * stride + 64) * 4 / 1e9);
    return nullptr;
  }
  KernelTimer tm("pack", db->nbytes);
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
  {   // how many reads hold a byte without a 2-bit code (0 for everything the reference's encoder wrote from ACGT reads): the alignment
      // launches only start their byte-wise second launch when there is one
    uint32_t *d_cnt = ws<uint32_t>("pack.nflag_count", 1);
    size_t rb = 0;
    PGX_HIP(hipcub::DeviceReduce::Sum(nullptr, rb, db->d_nflag.p, d_cnt, (int)nr, st));
    void *rt = ws_raw("pack.red_tmp", rb);
    PGX_HIP(hipcub::DeviceReduce::Sum(rt, rb, db->d_nflag.p, d_cnt, (int)nr, st));
    PGX_HIP(hipMemcpyAsync(&db->n_flagged_reads, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    sync();
  }
  db->packs_built = true;
  return db->d_pack.p;
}

}  // namespace pgx
