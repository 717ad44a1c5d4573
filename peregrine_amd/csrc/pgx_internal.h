// pgx_internal.h -- shared declarations of libpgx.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/pgx.h"

// Every kernel of this library is written for 64-lane wavefronts (ballots as 64-bit masks, `threadIdx.x & 63` lane ids, DPP row
// patterns): refuse to build for a wave32 target instead of computing wrong scan starts there.
// (neither __AMDGCN_WAVEFRONT_SIZE nor a constexpr warpSize exists on ROCm 7 / gfx950; CDNA targets have no wave32 mode, so the
// guard is on the target itself)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libpgx kernels assume 64-lane wavefronts and gfx950 instructions: build with --offload-arch=gfx950 only"
#endif

namespace pgx {

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
void set_error(const char *fmt, ...);
struct Fail {
  int code;
};
#define PGX_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      pgx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      throw pgx::Fail{PGX_EHIP};                                                              \
    }                                                                                         \
  } while (0)
#define PGX_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      pgx::set_error(__VA_ARGS__);   \
      throw pgx::Fail{code};         \
    }                                \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// context: one device, one stream
// ---------------------------------------------------------------------------------------------------------
struct Context {
  int device = -1;
  hipStream_t stream = nullptr;
  bool ready = false;
  int num_cu = 256;
};
Context &ctx();
void require_ready();
// Device state that outlives a call (plan caches, buffers handed out as library-owned views) registers a reset function:
// pgx_shutdown() runs them all BEFORE it returns the cached blocks to the driver, so nothing survives into the next pgx_init()
// (which may choose another device).  Use: static pgx::ShutdownHook h_([] { ... });
void on_shutdown(void (*fn)());
struct ShutdownHook {
  explicit ShutdownHook(void (*fn)()) { on_shutdown(fn); }
};
// bumped by every index-stage call that rewrites the index workspaces ("ix.*"): consumers of zero-copy views of those
// workspaces (pgx_pairs_prepare_dev -> pgx_pairs_scatter_dev) compare it to detect that their input was overwritten
uint64_t &index_generation();
bool ws_contains(const void *p);    // p points into one of the named workspaces
bool index_owns(const void *p);     // p points into memory an index-stage call handed out as a library-owned view

// device buffer (RAII, grow-only reuse is up to the caller)
// Device blocks come from a size-class cache: hipMalloc / hipFree are synchronous and cost tens of microseconds each, and
// the stages allocate dozens of temporaries per call.  All work is enqueued on ONE stream, so handing a released block to
// the next user is stream-ordered and safe.  pgx_shutdown() returns everything to the driver.
void *dev_alloc(size_t bytes);
void dev_release(void *p);
void dev_cache_trim();
size_t dev_cache_free_bytes();   // what the cache holds for re-use (the driver counts it as used)
void dev_cache_age();   // a stage starts: cached blocks of 64 MiB and more that two whole stages did not ask for go back to the driver
// device blocks allocated while a MemTag is in scope (on this thread) are booked under its name in the ledger (pgx_mem_ledger)
struct MemTag {
  explicit MemTag(const char *t);
  ~MemTag();
  MemTag(const MemTag &) = delete;
  MemTag &operator=(const MemTag &) = delete;
  const char *prev;
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) p = (T *)dev_alloc(count * sizeof(T));
  }
  void release() {
    if (p) dev_release(p);
    p = nullptr, n = 0;
  }
  void upload(const T *src, size_t count) {
    if (count) PGX_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx().stream));
  }
  void download(T *dst, size_t count) const {
    if (count) PGX_HIP(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, ctx().stream));
  }
};
inline void sync() { PGX_HIP(hipStreamSynchronize(ctx().stream)); }

// grow-only named device buffers that persist across calls (hipMalloc/hipFree are synchronous and slow)
void *ws_raw(const char *name, size_t bytes);
template <typename T>
T *ws(const char *name, size_t count) {
  return (T *)ws_raw(name, (count ? count : 1) * sizeof(T));
}

// ---------------------------------------------------------------------------------------------------------
// per-kernel timing with HIP events on the library stream
// ---------------------------------------------------------------------------------------------------------
struct KernelTimer {
  KernelTimer(const char *name, uint64_t units);
  ~KernelTimer();
  const char *name;
  uint64_t units;
  hipEvent_t e0, e1;
};
void timing_flush();  // resolve pending events into the totals (synchronises the stream)

// ---------------------------------------------------------------------------------------------------------
// resident read database
// ---------------------------------------------------------------------------------------------------------
struct ReadDesc {
  uint64_t off;
  uint32_t len;
  uint32_t rid;
};
}  // namespace pgx

struct pgx_seqdb {
  pgx::DevBuf<uint8_t> d_seq;      // seqdb bytes + 1 KiB of zero padding
  bool borrowed = false;           // d_seq.p belongs to the caller (pgx_seqdb_adopt_dev): never released here
  ~pgx_seqdb() {
    if (borrowed) d_seq.p = nullptr, d_seq.n = 0;
  }
  // the packed alignment kernel's view of the reads (pgx_pack.hip): a cache of the immutable seqdb bytes, built on first use
  mutable pgx::DevBuf<uint32_t> d_pack;          // 2-bit packs of the seqdb, both strands
  mutable pgx::DevBuf<uint32_t> d_nflag;         // by rid: the read holds a byte that has no 2-bit code (an ambiguous base)
  mutable pgx::DevBuf<uint64_t> d_poff;          // by rid: dword index of the read's forward strand in d_pack (its reverse complement follows)
  mutable pgx::DevBuf<uint32_t> d_prank;         // by rid: the read's rank in the packs' layout (the alignment launches take their requests in this order)
  mutable pgx::DevBuf<uint64_t> d_locus_key;     // by rid, only until the packs are built: smallest top-level shimmer hash (pgx_pack.hip)
  mutable bool locus_key_filled = false, locus_ordered = false;   // (ordered: the packs' reads are laid out by locus key, not in file order)
  mutable bool packs_built = false, packs_failed = false;   // (failed: no HBM for them -- the byte-wise kernels serve this database)
  mutable uint32_t n_flagged_reads = 0;                     // reads marked in d_nflag (known once the packs are built)
  pgx::DevBuf<uint64_t> d_roff;    // indexed by rid
  pgx::DevBuf<uint32_t> d_rlen;    // indexed by rid
  std::vector<uint32_t> rid, rlen; // idx-file order
  std::vector<uint64_t> roff;
  std::vector<uint32_t> rlen_by_rid;
  uint32_t max_rlen = 0;  // longest read (chooses the 16-bit V ring of the alignment kernel)
  std::vector<uint64_t> roff_by_rid;
  size_t nbytes = 0;
  uint64_t bases = 0;
  // the read selections of the last index calls (rid % total == mychunk % total, idx order), one per (total, chunk): the steps of a resident
  // pipeline cycle through the job's chunks, and rebuilding + re-uploading a selection of 775 k reads costs 9.5 ms of host time per stage
  // (76 ms of a full-size configs[3] step through round 4, when only the LAST selection was kept)
  struct IndexPlan {
    int total = -1, chunk = -1;
    uint64_t serial = 0, bases = 0, last_use = 0;
    std::vector<pgx::ReadDesc> reads;
  };
  std::vector<IndexPlan> plans;   // at most 32, least recently used replaced
};

namespace pgx {

// ---------------------------------------------------------------------------------------------------------
// device stages (pgx_kernels.hip).  All take/return device pointers and run on ctx().stream.
// ---------------------------------------------------------------------------------------------------------
// L0 minimizers of the given reads (device array of ReadDesc, in output order). Returns device list.
void dev_sketch(const pgx_seqdb *db, const std::vector<ReadDesc> &reads, int w, int k, DevBuf<pgx_mm128> &out,
                size_t &n_out, uint32_t *n_literal);
// fused index path: sketch (wave kernel) -> per-read reduce x levels in LDS -> ordered gather.  Returns false (and
// leaves the outputs untouched) when the chunk needs the general path (other w/k ...).
// plan_serial != 0: identifies `reads` (same serial => same list as the last call: descriptors and slab offsets are still on the device)
bool dev_index_fused(const pgx_seqdb *db, const std::vector<ReadDesc> &reads, int w, int k, int rs, int levels,
                     const pgx_mm128 **d_top, size_t *n_top, uint64_t plan_serial = 0, uint32_t *n_second = nullptr);   // n_second: reads sketched run by run (ambiguous bases)
// one mm_reduce level over a device list
void dev_reduce(const pgx_mm128 *d_in, size_t n, int rs, DevBuf<pgx_mm128> &out, size_t &n_out);
// multiplicity of x>>8, sorted by mer
void dev_count(const pgx_mm128 *d_in, size_t n, int kmer_bits, DevBuf<pgx_mm_count> &out, size_t &n_out);
// banded O(ND) confirmation of n candidate alignments (keys on device)
void dev_align(const pgx_seqdb *db, const pgx_align_key *d_keys, size_t n, int band, pgx_match *d_out,
               int tail_batch = 0);   // tail_batch: 1 = the second request batch of a stage, 2 = a later one (mostly hard candidates: pgx_align.hip)
void dev_align_prepare(const pgx_seqdb *db);   // the database's 2-bit packs, ahead of the first large launch (no-op once they exist)
// the 2-bit packs of a read database (pgx_pack.hip: read by read, [forward strand | reverse complement] at dword d_poff[rid]; d_nflag
// marks the reads with bytes that have no 2-bit code): built on first use, kept with the database; nullptr: no HBM
const uint32_t *seq_packs(const pgx_seqdb *db);
bool seq_packs_valid(const pgx_seqdb *db);
// the locus key of the reads, gathered by the overlap stage's join before the first seq_packs() (no-ops once the packs exist)
uint64_t *seq_locus_key_buffer(const pgx_seqdb *db);
void locus_key_add_mm(const pgx_seqdb *db, const pgx_mm128 *d_mm, size_t n);
void locus_key_add_records(const pgx_seqdb *db, const uint64_t *d_key0, const uint64_t *d_y0, size_t n);

// Large host arrays.  Never value-initialised (they are about to be overwritten); from 16 MiB up they are pooled anonymous
// mappings advised to use transparent huge pages, which the allocator would not do for us (THP is in "madvise" mode on
// the target hosts): first-touch faults and the final munmap are ~500x fewer than with 4 KiB pages.
void *big_alloc(size_t bytes);            // never returns nullptr (throws std::bad_alloc); content arbitrary
void big_free(void *p, size_t bytes);
void *big_alloc_zero(size_t bytes);       // all-zero mapping (fresh, or one that big_free_zero cleared)
void big_free_zero(void *p, size_t bytes);
void big_pool_trim();                     // unmap everything pooled (pgx_shutdown)
void *out_alloc(size_t bytes);            // an array for the caller (pgx_free): malloc, or a pooled mapping when large
void out_free(void *p);
template <typename T>
struct HostArray {
  T *p = nullptr;
  size_t n = 0;
  HostArray() = default;
  explicit HostArray(size_t count) { alloc(count); }
  HostArray(const HostArray &) = delete;
  HostArray &operator=(const HostArray &) = delete;
  HostArray(HostArray &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  HostArray &operator=(HostArray &&o) noexcept {
    if (this != &o) {
      clear();
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  ~HostArray() { clear(); }
  void alloc(size_t count) {
    clear();
    p = (T *)big_alloc((count ? count : 1) * sizeof(T));
    n = count;
  }
  void clear() {
    if (p) big_free(p, (n ? n : 1) * sizeof(T));
    p = nullptr, n = 0;
  }
  T *data() { return p; }
  const T *data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  T *begin() { return p; }
  T *end() { return p + n; }
  const T *begin() const { return p; }
  const T *end() const { return p + n; }
  const T &back() const { return p[n - 1]; }
};

// shimmer-pair join (pgx_pairs.hip): records sorted by (key0, key1, position desc, insertion order) plus the bucket /
// key0-group tables the host needs to replay the khash slot order on distinct keys
struct PairTables {
  size_t n_rec = 0;
  size_t n_groups = 0, n_buckets = 0;   // (always set; the arrays below only when the tables came to the host: PAIRS_DEV_TABLES)
  bool on_host = true;
  HostArray<uint64_t> y0;        // per record
  HostArray<uint8_t> dir;        // per record
  HostArray<uint64_t> bkey1;     // per bucket
  HostArray<uint32_t> bstart;    // nb+1 record offsets
  HostArray<uint32_t> bfirst;    // first insertion (record seq) of the bucket
  HostArray<uint64_t> gkey0;     // per key0 group
  HostArray<uint32_t> gstart;    // ng+1 record offsets
  HostArray<uint32_t> gfirst, glast;  // first / last insertion of the group
  HostArray<uint32_t> gbucket;   // ng+1: first bucket of the group
  HostArray<uint32_t> gord;      // groups in order of first insertion
  HostArray<uint32_t> bord;      // buckets ordered by (group, first insertion): group g's slice is [gbucket[g], gbucket[g+1])
  // PAIRS_ORD_TABLES: what the inner-table replay reads, already IN bord ORDER (sequential on the host instead of three gathers
  // per bucket): second key and size of bucket bord[i]; per group: is a put repeated after its last first-insertion (khash.h:298-306)
  HostArray<uint64_t> bkey1_ord;
  HostArray<uint32_t> bn_ord;
  HostArray<uint8_t> gtrail;
  HostArray<uint64_t> y1;        // per record, only with PAIRS_Y1
  HostArray<uint64_t> umer;      // aggregated multiplicities, sorted by mer: only with PAIRS_COUNTS
  HostArray<uint32_t> ucnt;
};
struct PairParams {
  uint32_t total, mychunk, lower, upper;  // bucket ownership (x>>8) % total == mychunk % total; multiplicity bounds
  uint32_t n_rid = 0;                     // entries of d_rlen (0: unknown, the list is not checked against the read database)
};
enum : unsigned {
  PAIRS_Y1 = 1,               // also return the second coordinate of every record (mp128_t.y1)
  PAIRS_INSERTION_ORDER = 2,  // records of a bucket in insertion order instead of position-descending
  PAIRS_COUNTS = 4,           // also return the aggregated multiplicity table
  PAIRS_ORD_TABLES = 16,      // bkey1_ord / bn_ord / gtrail instead of bkey1 / bfirst (the overlap stage's table replay)
  PAIRS_LAZY_RECORDS = 8,     // with `keep`: leave the sorted records (y0, dir) on the device only; pairs_fetch_records downloads them
  PAIRS_DEV_TABLES = 32,      // with `keep` + PAIRS_ORD_TABLES: leave the group / bucket tables on the device only (the visit order is
                              // built there, pgx_visit.hip); pairs_fetch_tables downloads them for the host paths
};
// what the join leaves in HBM for the device replay (pgx_replay.hip): the bucket-sorted records
struct DevicePairs {
  DevBuf<uint64_t> y0;       // per record: rid << 32 | lastPos << 1 | strand
  DevBuf<uint8_t> dir;       // per record
  DevBuf<uint32_t> bstart;   // n_buckets + 1 record offsets
  size_t n_rec = 0, n_buckets = 0;
  bool valid = false;
  // PAIRS_DEV_TABLES: the tables of PairTables (same names, same contents), left where the join computed them
  bool tables = false;
  size_t n_groups = 0;
  DevBuf<uint32_t> gstart, gbucket;   // n_groups + 1 each (end sentinels)
  DevBuf<uint32_t> gord, gfirst, glast, bord, bn_ord;
  DevBuf<uint64_t> gkey0, bkey1_ord;
  DevBuf<uint8_t> gtrail;
  uint32_t max_group_buckets = 0;     // most buckets a first-key group holds
  uint32_t n_big_groups = 0;          // groups with more than VISIT_LANE_MAX buckets
  DevBuf<uint32_t> big_groups;        // their indices
  std::vector<uint64_t> key_sample;   // gkey0[gord[i * KEY_SAMPLE_STRIDE]]: what the early outer-table keys are checked against
  uint32_t last_gfirst = 0;           // gfirst[gord[n_groups - 1]]
};
constexpr uint32_t VISIT_LANE_MAX = 48;     // buckets of a group one lane replays (a 64-slot table: pgx_visit.hip)
constexpr uint32_t VISIT_WAVE_MAX = 3153;   // ... a wavefront (a 4,096-slot table); larger groups: the host replay of the tables
constexpr uint32_t KEY_SAMPLE_STRIDE = 997;
void pairs_fetch_tables(const DevicePairs &dp, PairTables &out);   // (no-op when the tables are on the host already)
// the visit order on the device (pgx_visit.hip): inner tables of every first-key group replayed by a lane / a wavefront each
// (enqueued; runs while the host finishes the outer table), then the groups' visited buckets placed in outer-slot order
struct DevVisit {
  DevBuf<uint32_t> ids_all, gnb;
  DevBuf<unsigned long long> tot;
};
void dev_visit_inner(const DevicePairs &dp, uint32_t ovlp_upper, DevVisit &v);
// slots: the outer table as DistinctSlotTable leaves it (pinned host memory), ids = positions in first-insertion order
void dev_visit_place(const DevicePairs &dp, DevVisit &v, const uint64_t *slots, uint32_t n_slots, DevBuf<uint32_t> &bid, size_t *n_buckets,
                     size_t *n_entries);
// The distinct first keys of the records in the order of their first insertion -- what the host replays klib's OUTER table from
// (pgx_overlap.cpp) -- computed right after the records exist (a hash aggregation of first occurrences + an ordered select) and
// handed to `early` while the join's sorts are still to run: the outer-table replay, the longest sequential piece of host work
// of the stage, then overlaps the rest of the join.  keys[i] == gkey0[gord[i]] of the tables the join returns.
struct EarlyGroups {
  HostArray<uint64_t> keys;
  uint32_t n = 0;
  uint32_t last_first = 0;   // record index of the last key's first occurrence
};
using EarlyFn = std::function<void(EarlyGroups &&)>;
// d_rlen: read length by rid, on the device
void dev_build_pairs(const uint32_t *d_rlen, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
                     size_t n_counts, const PairParams &pp, PairTables &out, unsigned flags = 0,
                     const pgx_mm128 *d_mmers = nullptr, const pgx_mm_count *d_counts = nullptr,  // d_*: the same lists, already on the device
                     DevicePairs *keep = nullptr,   // keep: the sorted records stay on the device too
                     const EarlyFn &early = nullptr,
                     const pgx_seqdb *locus_db = nullptr);   // locus_db: its packs' locus keys are gathered from the lists on the way (pgx_pack.hip)
// ---- multi-GPU hand-over: counts all-gathered, pair records routed to their owner chunk -------------------------
// prepare: aggregate ALL chunks' counts, flag the kept shimmers of THIS index chunk's list (both on the device); returns the index
// of the first shimmer with lower <= count < upper (-1: none).  scatter: the records of every adjacent kept pair from `start` on
// (the list position the global scan starts at, shmr_utils.c:311-320), grouped by destination chunk 1..T in scan order; *d_send
// stays valid until the next prepare.  from_records: the join of one overlap chunk over the records it received, in arrival order
// (= source chunk order, then scan order: the insertion order of build_map over the concatenated lists).
int64_t dev_pairs_prepare(const uint32_t *d_rlen, uint32_t n_rid, const pgx_mm128 *d_mm, size_t n_mm, const pgx_mm_count *d_counts,
                          size_t n_counts, uint32_t lower, uint32_t upper);
void dev_pairs_scatter(const uint32_t *d_rlen, uint32_t T, int64_t start, const pgx_pair_rec **d_send, uint64_t *counts);
void dev_pairs_from_records(const pgx_pair_rec *d_rec, size_t n, PairTables &out, DevicePairs *keep_dev, unsigned flags = 0,
                            const EarlyFn &early = nullptr, const pgx_seqdb *locus_db = nullptr);
void pairs_fetch_records(const DevicePairs &dp, PairTables &out);  // the lazily kept records, to the host tables

// The greedy walk over the visit list (visit_bids: the join's bucket ids in visit order) on the GPU; the records go to the
// array alloc_out(n) returns.  false: the job does not fit the device tables' encodings or they overflowed -- nothing was
// produced and the caller runs the host replay.
void dev_place_bids(const uint32_t *ids_all, size_t n_ids, const uint32_t *psrc, const uint32_t *pcnt, const uint64_t *pdst,
                    size_t n_groups, size_t nb, DevBuf<uint32_t> &bid);   // the visit list assembled on the device
bool dev_replay(const pgx_seqdb *db, const DevicePairs &dp, const uint32_t *visit_bids, const uint32_t *d_bids, size_t nb, size_t n_entries,
                uint32_t bestn, int band, bool predict, uint32_t ovlp_upper, const std::function<pgx_ovlp *(size_t)> &alloc_out,
                size_t *n_out, pgx_overlap_stats *st, bool trace);

// what the index stage leaves in HBM for a following overlap stage of the same process (pgx_index_overlap_resident)
struct DeviceIndex {
  const pgx_mm128 *d_top = nullptr;  // lives in the index workspace: valid until the next index call
  size_t n_top = 0;
  DevBuf<pgx_mm_count> mc;
  size_t n_mc = 0;
  bool valid = false;
};
// the index stage; keep != nullptr: leave the final list + counts on the device too; host_arrays == false: do not download them
void index_stage(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *out, DeviceIndex *keep, bool host_arrays);

// Records on their way to the host while the caller already works on the next chunk (pgx_results_async, include/pgx.h): the overlap
// stage's device replay hands its record buffer to a copy on a second stream and returns; results_wait() -- called by
// pgx_results_wait, by the next stage before it allocates its own buffer, by pgx_free of that array and by every entry point that
// reads the records itself -- waits for the copy and gives the device buffer back.
bool &results_async();
void results_wait();
void results_wait_if(const void *host);   // only if `host` is the array the pending copy writes
void results_copy_async(pgx_ovlp *host, DevBuf<pgx_ovlp> &&dev, size_t n);   // after what is enqueued on ctx().stream

// ---- file-level entry points of the overlap stage (pgx_served.cpp) over the stage itself (pgx_overlap.cpp) -------------------------------
struct OvOut {  // the stage's output: one malloc'd array handed to the caller as is (a == nullptr, n set: the records went to a RecordSink)
  pgx_ovlp *a = nullptr;
  size_t n = 0;
  void alloc(size_t count) {
    out_free(a);
    a = (pgx_ovlp *)out_alloc(count ? count * sizeof(pgx_ovlp) : 1);
    n = count;
  }
  pgx_ovlp *release() {
    pgx_ovlp *p = a;
    a = nullptr, n = 0;
    return p;
  }
  ~OvOut() { out_free(a); }
};
struct DeviceLists {   // the lists as device arrays
  const pgx_mm128 *d_top = nullptr;
  const pgx_mm_count *d_mc = nullptr;
};
void overlap_check_params(const pgx_overlap_params *p);
void overlap_stage(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts, const pgx_overlap_params *p,
                   OvOut &out, pgx_overlap_stats *st, const DeviceLists *dev = nullptr);
// Where a stage's records go when they are not wanted as a host array (served commands: straight to the output file).  The device replay hands
// its record buffer over right behind k_emit -- take() returns at once, the transfer runs on the sink's own threads and streams -- and
// allocates no host array; the host replay (small sets, fall-back) delivers a host array that the caller writes itself.
struct RecordSink {
  virtual void take(DevBuf<pgx_ovlp> &&dev, size_t n) = 0;   // ordered behind what is enqueued on ctx().stream
  virtual ~RecordSink() {}
};
RecordSink *&record_sink();   // (one stage at a time per process: set around overlap_stage by the caller that owns the sink)
// Served jobs: an index command leaves a device copy of every final-level list / count file it wrote (keyed by the file's absolute path,
// size and mtime); the job's overlap commands assemble their input from those copies instead of reading the files back (pgx_served.cpp).
// d_payload (device) or h_payload (host): the file's entries, without the 8-byte count header; call AFTER the file is closed.
void list_stash_put(const std::string &path, const void *d_payload, const void *h_payload, size_t bytes);
void list_stash_clear();
void count_cache_drop();     // the aggregated count table kept across the chunks of a job (pgx_pairs.hip)
void replay_forget_sizes();  // the device replay's learned table sizes (pgx_replay.hip): another database, another job
void replay_drop_precleared();
void replay_preclear();      // tables of the last stage's sizes, allocated and cleared ahead of the replay (while the GPU waits for the host's outer table)

// pgx_overlap_stats::stream_checksum: the sum over the records of a 64-bit mix of every field (padding bytes excluded) and the record's
// position in the stream -- the same on the device (k_emit adds it up while it writes the records) and on the host
__host__ __device__ inline uint64_t checksum_mix(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 33, h *= 0xc4ceb9fe1a85ec53ULL, h ^= h >> 33;
  return h;
}
__host__ __device__ inline uint64_t record_checksum(const pgx_ovlp &o, uint64_t pos) {
  uint64_t h = checksum_mix(o.y0 + 0x9E3779B97F4A7C15ULL * (pos + 1));
  h = checksum_mix(h ^ o.y1);
  h = checksum_mix(h ^ ((uint64_t)o.rl0 | (uint64_t)o.rl1 << 32));
  h = checksum_mix(h ^ ((uint64_t)o.strand0 | (uint64_t)o.strand1 << 8 | (uint64_t)o.ovlp_type << 16));
  h = checksum_mix(h ^ ((uint64_t)(uint32_t)o.match.m_size | (uint64_t)(uint32_t)o.match.dist << 32));
  h = checksum_mix(h ^ ((uint64_t)(uint32_t)o.match.q_bgn | (uint64_t)(uint32_t)o.match.q_end << 32));
  h = checksum_mix(h ^ ((uint64_t)(uint32_t)o.match.t_bgn | (uint64_t)(uint32_t)o.match.t_end << 32));
  h = checksum_mix(h ^ ((uint64_t)(uint32_t)o.match.t_m_end | (uint64_t)(uint32_t)o.match.q_m_end << 32));
  return h;
}

// Runs fn on the library's housekeeping thread: tearing down GB-sized host tables (munmap, free) takes tens of
// milliseconds that the caller does not have to wait for.  At most a few jobs are queued; beyond that fn runs inline.
void defer_destroy(std::function<void()> fn);
void drain_deferred();  // waits until every queued job has run (pgx_shutdown)

// host helpers (pgx_api.cpp)
int load_idx(const char *path, std::vector<uint32_t> &rid, std::vector<uint32_t> &rlen, std::vector<uint64_t> &roff);
bool read_file(const std::string &path, std::vector<uint8_t> &out);
template <typename T>
T *host_copy(const std::vector<T> &v) {
  T *p = (T *)malloc(v.size() ? v.size() * sizeof(T) : 1);
  if (v.size()) memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}
}  // namespace pgx
