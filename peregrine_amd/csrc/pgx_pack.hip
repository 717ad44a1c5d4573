// pgx_pack.hip -- the 2-bit packs of a read database that the alignment kernels read (k_align_ph<8, u16, packed>, pgx_align.hip).
//
// LAYOUT (round 6): read by read, both strands of a read next to each other, every strand starting at a dword:
//   read r:  [ forward strand: nw(r) dwords | reverse complement, stored forward: nw(r) dwords ],  nw(r) = ceil(len(r) / 16),
//   base i of a strand in bits 2(i % 16) .. +1 of its dword i / 16 (s = 0: low nibbles = the read as stored; 1: high nibbles = its
//   reverse complement: /root/reference/src/shmr_utils.c:44-51).  d_poff[rid] = dword index of the forward strand, d_prank[rid] = the
//   read's rank in the layout.
// ORDER of the reads = ascending LOCUS KEY when the overlap stage has provided one before the first build (seq_locus_key_buffer: the
// smallest top-level shimmer hash of a read -- a min-hash: reads that overlap by half share it with probability ~1/3, so the ~30 reads
// covering a locus fall into 2-3 runs instead of 30 random places), else the order of the seqdb file.  Why: at full-size configs[3] the
// packs are 47 GB and the reads of the file are in random genome order; a CU's 256 in-flight candidates walked 512 reads = 512 different
// 2 MB translation ranges and the per-CU translation cache missed 41.8 % of its requests (profiles/r05e_pmc_align_memory_c4_vs_c3.txt;
// tools/tlb_probe.hip: a dependent load costs 107 ns while its ranges fit, 242 -> 376 ns beyond).  With the reads laid out by locus
// and the REQUESTS of a launch taken in layout order by the wavefronts of one workgroup (pgx_align.hip), a CU works on one or two
// loci at a time.  Rounds 3-5: [pack of all low nibbles | pack of all high nibbles] at the seqdb's byte offsets.
// A byte whose nibbles are not both one-hot codes (an ambiguous base, or anything the reference's encoder never writes) has no 2-bit
// code: it marks its read in nflag[] and candidates that touch such a read take the byte-wise kernels, which compare the nibbles as
// DWmatch.c:136-137 does.
//
// The packs are a CACHE of the immutable seqdb bytes, kept with the pgx_seqdb for as long as it lives: built by the first large
// alignment launch (or ahead of it, beside the join: dev_align_prepare) and reused by every later stage on the same database, on every
// rank of a multi-GPU job.  If their HBM (seqdb / 2) cannot be had, the alignment launches stay on the byte-wise kernels.
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
// a workgroup per read (in layout order): thread t takes dwords t, t + 256, ... of both strands; 16 seqdb bytes per dword, loaded
// unaligned (a read starts at any byte; the seqdb buffer carries 1 KiB of zero padding behind its last byte)
__global__ __launch_bounds__(256) void k_pack_reads(const uint8_t *__restrict__ seq, const uint32_t *__restrict__ ord, uint32_t nreads,
                                                    const uint64_t *__restrict__ roff, const uint32_t *__restrict__ rlen,
                                                    const uint64_t *__restrict__ poff, uint32_t *__restrict__ pack, uint32_t *__restrict__ nflag) {
  for (uint32_t i = blockIdx.x; i < nreads; i += gridDim.x) {
    const uint32_t rid = ord[i], len = rlen[rid], nw = (len + 15u) >> 4;
    const uint8_t *src = seq + roff[rid];
    uint32_t *p0 = pack + poff[rid], *p1 = p0 + nw;
    uint32_t bad = 0;
    for (uint32_t w = threadIdx.x; w < nw; w += 256) {
      uint4 v;
      __builtin_memcpy(&v, src + (size_t)w * 16, 16);
      const uint32_t M = 0x0F0F0F0Fu;
      const uint32_t l[4] = {v.x & M, v.y & M, v.z & M, v.w & M};
      const uint32_t h[4] = {(v.x >> 4) & M, (v.y >> 4) & M, (v.z >> 4) & M, (v.w >> 4) & M};
      p0[w] = pack4(l[0]) | (pack4(l[1]) << 8) | (pack4(l[2]) << 16) | (pack4(l[3]) << 24);
      p1[w] = pack4(h[0]) | (pack4(h[1]) << 8) | (pack4(h[2]) << 16) | (pack4(h[3]) << 24);
      const uint32_t left = len - w * 16;   // bytes of the read from this dword on (>= 1); what lies behind them is the next read
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t vb = left > 4u * d ? min(left - 4u * d, 4u) : 0u;          // valid bytes of this dword of four
        const uint32_t vm = vb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * vb)) - 1u);
        bad |= (not_onehot(l[d]) | not_onehot(h[d])) & vm;
      }
    }
    if (bad) atomicOr(&nflag[rid], 1u);
  }
}
__global__ void k_iota_key(uint32_t n, const uint64_t *__restrict__ locus, const uint64_t *__restrict__ roff, const uint32_t *__restrict__ rlen,
                           uint64_t *__restrict__ key, uint32_t *__restrict__ rid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rid[i] = i;
  // a read without a key (no shimmer in the lists: ~0) sorts behind the others, in file order among themselves (the sort is stable and
  // the rids come in ascending order; the file's order is the order of the offsets)
  key[i] = locus ? locus[i] : roff[i];
  (void)rlen;
}
__global__ void k_words_of(uint32_t n, const uint32_t *__restrict__ ord, const uint32_t *__restrict__ rlen, uint64_t *__restrict__ words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) words[i] = 2ull * ((rlen[ord[i]] + 15u) >> 4);
}
__global__ void k_scatter_layout(uint32_t n, const uint32_t *__restrict__ ord, const uint64_t *__restrict__ start, uint64_t *__restrict__ poff,
                                 uint32_t *__restrict__ prank) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) poff[ord[i]] = start[i], prank[ord[i]] = i;
}
__global__ void k_locus_min_mm(const pgx_mm128 *__restrict__ mm, size_t n, uint32_t nreads, unsigned long long *__restrict__ key) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const pgx_mm128 m = mm[i];
    const uint32_t rid = (uint32_t)(m.y >> 32);
    if (rid < nreads) atomicMin(&key[rid], (unsigned long long)(m.x >> 8));
  }
}
__global__ void k_locus_min_rec(const uint64_t *__restrict__ key0, const uint64_t *__restrict__ y0, size_t n, uint32_t nreads,
                                unsigned long long *__restrict__ key) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t rid = (uint32_t)(y0[i] >> 32);
    if (rid < nreads) atomicMin(&key[rid], (unsigned long long)key0[i]);
  }
}

}  // namespace

bool seq_packs_valid(const pgx_seqdb *db) { return db->packs_built && db->d_pack.p != nullptr; }

// ---- the locus key of every read, gathered by the overlap stage's join BEFORE the packs exist ---------------------------------------
// nullptr: not wanted (the packs are built, or failed, or PGX_PACK_LOCUS=0) -- the caller then skips the pass.  The buffer is filled by
// locus_key_add_* (atomic minima over whatever the stage holds: all top-level shimmers of all reads on one GPU, the pair records it
// received on a rank of a multi-GPU job) and consumed by the first seq_packs().
uint64_t *seq_locus_key_buffer(const pgx_seqdb *db) {
  if (db->packs_built || db->packs_failed) return nullptr;
  if (getenv("PGX_PACK_LOCUS") && atoi(getenv("PGX_PACK_LOCUS")) == 0) return nullptr;
  const size_t nr = db->rlen_by_rid.size();
  if (nr == 0) return nullptr;
  if (db->d_locus_key.n < nr) {
    db->d_locus_key.alloc(nr);
    PGX_HIP(hipMemsetAsync(db->d_locus_key.p, 0xFF, nr * sizeof(uint64_t), ctx().stream));
  }
  return db->d_locus_key.p;
}
void locus_key_add_mm(const pgx_seqdb *db, const pgx_mm128 *d_mm, size_t n) {
  uint64_t *key = seq_locus_key_buffer(db);
  if (!key || !n) return;
  hipLaunchKernelGGL(k_locus_min_mm, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx().num_cu * 16)), dim3(256), 0, ctx().stream, d_mm, n,
                     (uint32_t)db->rlen_by_rid.size(), reinterpret_cast<unsigned long long *>(key));
  db->locus_key_filled = true;
}
void locus_key_add_records(const pgx_seqdb *db, const uint64_t *d_key0, const uint64_t *d_y0, size_t n) {
  uint64_t *key = seq_locus_key_buffer(db);
  if (!key || !n) return;
  hipLaunchKernelGGL(k_locus_min_rec, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx().num_cu * 16)), dim3(256), 0, ctx().stream, d_key0,
                     d_y0, n, (uint32_t)db->rlen_by_rid.size(), reinterpret_cast<unsigned long long *>(key));
  db->locus_key_filled = true;
}

// the packs (layout above); nullptr: their HBM could not be allocated (the caller uses the byte-wise kernels)
const uint32_t *seq_packs(const pgx_seqdb *db) {
  if (seq_packs_valid(db)) return db->d_pack.p;
  if (db->packs_failed || !db->d_seq.p) return nullptr;   // (no bytes to pack from: they were released, and then the packs are valid)
  hipStream_t st = ctx().stream;
  const size_t nr = db->rlen_by_rid.size();
  size_t total_words = 0;
  for (uint32_t l : db->rlen_by_rid) total_words += 2 * (((size_t)l + 15) >> 4);
  try {
    MemTag mem_tag("seqdb.packs_2bit");
    if (db->d_pack.n < total_words + 64) db->d_pack.alloc(total_words + 64);
    if (db->d_poff.n < nr + 1) db->d_poff.alloc(nr + 1);
    if (db->d_prank.n < nr + 1) db->d_prank.alloc(nr + 1);
  } catch (const Fail &) {
    db->packs_failed = true;
    db->d_locus_key.release();
    (void)hipGetLastError();   // (the failed allocation's sticky error)
    if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] align: no HBM for the 2-bit packs (%.1f GB): the byte-wise kernels take every launch\n", (total_words + 64) * 4 / 1e9);
    return nullptr;
  }
  KernelTimer tm("pack", db->nbytes);
  if (db->d_nflag.n < nr + 1) db->d_nflag.alloc(nr + 1);
  PGX_HIP(hipMemsetAsync(db->d_nflag.p, 0, (nr + 1) * sizeof(uint32_t), st));
  PGX_HIP(hipMemsetAsync(db->d_pack.p + total_words, 0, 64 * sizeof(uint32_t), st));
  const bool by_locus = db->locus_key_filled && db->d_locus_key.p != nullptr;
  if (nr) {
    const uint32_t n = (uint32_t)nr;
    DevBuf<uint64_t> k_in(nr), k_out(nr), words(nr), start(nr);
    DevBuf<uint32_t> r_in(nr), ord(nr);
    hipLaunchKernelGGL(k_iota_key, dim3((n + 255) / 256), dim3(256), 0, st, n, by_locus ? db->d_locus_key.p : (const uint64_t *)nullptr, db->d_roff.p,
                       db->d_rlen.p, k_in.p, r_in.p);
    size_t tb = 0, tb2 = 0;
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in.p, k_out.p, r_in.p, ord.p, (int)n, 0, 64, st));
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, words.p, start.p, (int)n, st));
    DevBuf<uint8_t> tmp(std::max(tb, tb2) + 256);
    PGX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, k_in.p, k_out.p, r_in.p, ord.p, (int)n, 0, 64, st));
    hipLaunchKernelGGL(k_words_of, dim3((n + 255) / 256), dim3(256), 0, st, n, ord.p, db->d_rlen.p, words.p);
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb2, words.p, start.p, (int)n, st));
    hipLaunchKernelGGL(k_scatter_layout, dim3((n + 255) / 256), dim3(256), 0, st, n, ord.p, start.p, db->d_poff.p, db->d_prank.p);
    hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)std::min<size_t>(nr, (size_t)ctx().num_cu * 64)), dim3(256), 0, st, db->d_seq.p, ord.p, n, db->d_roff.p,
                       db->d_rlen.p, db->d_poff.p, db->d_pack.p, db->d_nflag.p);
    PGX_HIP(hipGetLastError());
    {   // how many reads hold a byte without a 2-bit code (0 for everything the reference's encoder wrote from ACGT reads): the alignment
        // launches only start their byte-wise second launch when there is one
      uint32_t *d_cnt = ws<uint32_t>("pack.nflag_count", 1);
      size_t rb = 0;
      PGX_HIP(hipcub::DeviceReduce::Sum(nullptr, rb, db->d_nflag.p, d_cnt, (int)nr, st));
      void *rt = ws_raw("pack.red_tmp", rb);
      PGX_HIP(hipcub::DeviceReduce::Sum(rt, rb, db->d_nflag.p, d_cnt, (int)nr, st));
      PGX_HIP(hipMemcpyAsync(&db->n_flagged_reads, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    sync();   // (the temporaries above go back to the block cache here)
  }
  db->locus_ordered = by_locus;
  db->d_locus_key.release();
  db->packs_built = true;
  if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] packs: %.2f GB, reads in %s order\n", (total_words + 64) * 4 / 1e9, by_locus ? "locus-key" : "file");
  return db->d_pack.p;
}

}  // namespace pgx
