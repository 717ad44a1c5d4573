// pgx_api.cpp -- context, timing, resident seqdb, file formats, batch-level and shimmer4py entry points.
#include <glob.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "pgx_internal.h"

namespace pgx {

static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

Context &ctx() {
  static Context c;
  return c;
}
void require_ready() { PGX_REQUIRE(ctx().ready, PGX_ESTATE, "pgx_init() has not been called (or failed)"); }

// ---- large host arrays -------------------------------------------------------------------------------------
// Arrays of 1 MiB and more are pooled mappings: glibc serves such sizes by mmap / munmap, i.e. every stage call would map, fault in
// (4 KiB pages) and unmap its tables of a few MiB again -- ~20 k faults and ~80 MB of munmap per overlap call at 4.5 Gbases, 15-20 ms
// on a busy host, half of them in the background while the next stage runs.
static constexpr size_t HUGE = (size_t)2 << 20, BIG = (size_t)1 << 20;
// Mappings are pooled by size class instead of unmapped: munmap takes the address-space lock exclusively and stalls
// every page fault of the process for its duration -- freeing one call's tables in the background used to slow the NEXT
// stage's host side 2-3x.  Two pools: blocks with arbitrary content, and blocks known to be all zero (the lock-free
// tables need zero pages; they are cleared by whoever frees them, normally the housekeeping thread).
namespace {
struct BigPool {
  std::mutex mu;
  std::multimap<size_t, void *> any, zero;
  size_t held = 0;
  static size_t cls(size_t bytes) {  // eighths of a power of two, multiples of the huge page size
    size_t p2 = HUGE;
    while (p2 * 2 <= bytes) p2 <<= 1;
    const size_t step = std::max(p2 >> 3, HUGE);
    return (bytes + step - 1) / step * step;
  }
  static size_t cap() {
    static const size_t c = (size_t)16 << 30;
    return c;
  }
  void *take(std::multimap<size_t, void *> &pool, size_t len) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = pool.find(len);
    if (it == pool.end()) return nullptr;
    void *p = it->second;
    pool.erase(it);
    held -= len;
    return p;
  }
  bool give(std::multimap<size_t, void *> &pool, void *p, size_t len) {
    std::lock_guard<std::mutex> lk(mu);
    if (held + len > cap()) return false;
    pool.emplace(len, p);
    held += len;
    return true;
  }
  void trim() {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : any) (void)munmap(kv.second, kv.first);
    for (auto &kv : zero) (void)munmap(kv.second, kv.first);
    any.clear(), zero.clear();
    held = 0;
  }
};
BigPool &big_pool() {
  static BigPool p;
  return p;
}
void *map_fresh(size_t len) {
  void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) throw std::bad_alloc();
  (void)madvise(p, len, MADV_HUGEPAGE);
  return p;
}
}  // namespace
void *big_alloc(size_t bytes) {
  if (bytes < BIG) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
  }
  const size_t len = BigPool::cls(bytes);
  if (void *p = big_pool().take(big_pool().any, len)) return p;
  return map_fresh(len);
}
void big_free(void *p, size_t bytes) {
  if (!p) return;
  if (bytes < BIG) {
    free(p);
    return;
  }
  const size_t len = BigPool::cls(bytes);
  if (!big_pool().give(big_pool().any, p, len)) (void)munmap(p, len);
}
void *big_alloc_zero(size_t bytes) {  // zero-filled, always a mapping
  const size_t len = BigPool::cls(std::max(bytes, HUGE));
  if (void *p = big_pool().take(big_pool().zero, len)) return p;
  return map_fresh(len);
}
void big_free_zero(void *p, size_t bytes) {
  if (!p) return;
  const size_t len = BigPool::cls(std::max(bytes, HUGE));
  if (big_pool().held + len > BigPool::cap()) {
    (void)munmap(p, len);
    return;
  }
  memset(p, 0, len);  // (plain stores: no address-space lock, unlike munmap / MADV_DONTNEED)
  if (!big_pool().give(big_pool().zero, p, len)) (void)munmap(p, len);
}
void big_pool_trim() { big_pool().trim(); }

// Arrays handed to the caller (released with pgx_free): large ones come from the pooled huge-page mappings too, so that
// filling a 300 MB result does not start with 73,000 first-touch page faults; the registry tells pgx_free which is which.
// Large result arrays are PINNED (hipHostMalloc, pooled by size class like the mappings above): the 300 MB - 3 GB of ovlp_t records
// of an overlap chunk come down at the PCIe rate (~55 GB/s) instead of the ~16 GB/s a pageable destination gets (measured: 189 ms
// for the 3 GB of a human-scale chunk), and the copy is truly asynchronous.  Pinning is expensive (~0.3 ms per MB), hence the pool;
// at most PIN_CAP bytes are held back, beyond that (and when pinning fails) the array is a pageable pooled mapping as before.
namespace {
std::mutex g_out_mu;
std::map<void *, size_t> g_out_big;                  // pageable pooled mappings handed out
std::map<void *, size_t> g_out_pin;                  // pinned blocks handed out (size class)
std::multimap<size_t, void *> g_pin_free;            // pinned blocks waiting for re-use
size_t g_pin_held = 0;
// pinned result blocks kept for re-use: 16 GB per NODE -- divided among the ranks of a multi-process job (torchrun's LOCAL_WORLD_SIZE),
// as the host threads are (eight ranks used to pin up to 16 GB each)
const size_t PIN_CAP = ((size_t)16 << 30) / (size_t)std::max(1, getenv("LOCAL_WORLD_SIZE") ? atoi(getenv("LOCAL_WORLD_SIZE")) : 1);
}  // namespace
namespace {
struct PendingCopy {
  DevBuf<pgx_ovlp> dev;
  const void *host = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  hipStream_t stream = nullptr;
  bool active = false;
} g_copy;
bool g_results_async = false;
ShutdownHook g_copy_reset([] {
  if (g_copy.active) (void)hipEventSynchronize(g_copy.done);
  g_copy.dev.release();
  g_copy.active = false, g_copy.host = nullptr;
  if (g_copy.ready) (void)hipEventDestroy(g_copy.ready), g_copy.ready = nullptr;
  if (g_copy.done) (void)hipEventDestroy(g_copy.done), g_copy.done = nullptr;
  if (g_copy.stream) (void)hipStreamDestroy(g_copy.stream), g_copy.stream = nullptr;
});
}  // namespace
static std::recursive_mutex g_copy_mu;   // (out_free -> results_wait_if can run on the housekeeping thread or a Python finaliser: ADVICE r4)
bool &results_async() { return g_results_async; }
void results_wait() {
  std::lock_guard<std::recursive_mutex> lk(g_copy_mu);
  if (!g_copy.active) return;
  g_copy.active = false;
  const hipError_t e = hipEventSynchronize(g_copy.done);
  g_copy.dev.release();
  g_copy.host = nullptr;
  PGX_HIP(e);
}
void results_wait_if(const void *host) {
  std::lock_guard<std::recursive_mutex> lk(g_copy_mu);
  if (g_copy.active && g_copy.host == host) {
    try {
      results_wait();
    } catch (const Fail &) {
    }
  }
}
void results_copy_async(pgx_ovlp *host, DevBuf<pgx_ovlp> &&dev, size_t n) {
  std::lock_guard<std::recursive_mutex> lk(g_copy_mu);
  results_wait();   // (one copy in flight)
  if (!g_copy.stream) {
    PGX_HIP(hipStreamCreateWithFlags(&g_copy.stream, hipStreamNonBlocking));
    PGX_HIP(hipEventCreateWithFlags(&g_copy.ready, hipEventDisableTiming));
    PGX_HIP(hipEventCreateWithFlags(&g_copy.done, hipEventDisableTiming));
  }
  PGX_HIP(hipEventRecord(g_copy.ready, ctx().stream));
  PGX_HIP(hipStreamWaitEvent(g_copy.stream, g_copy.ready, 0));
  PGX_HIP(hipMemcpyAsync(host, dev.p, n * sizeof(pgx_ovlp), hipMemcpyDeviceToHost, g_copy.stream));
  PGX_HIP(hipEventRecord(g_copy.done, g_copy.stream));
  g_copy.dev = std::move(dev), g_copy.host = host, g_copy.active = true;
}

void *out_alloc(size_t bytes) {
  if (bytes < BIG) {
    void *p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
  }
  const size_t len = BigPool::cls(bytes);
  // (pinning pays from the SECOND result of a size class on -- a resident pipeline stepping over chunks; a one-shot process, the
  // drop-in executables, gets pageable arrays: pinning 300 MB costs more than its one transfer saves)
  static std::map<size_t, int> seen;
  bool repeat;
  {
    std::lock_guard<std::mutex> lk(g_out_mu);
    repeat = seen[len]++ > 0;
  }
  if (PIN_CAP && ctx().ready && repeat) {
    std::lock_guard<std::mutex> lk(g_out_mu);
    auto it = g_pin_free.find(len);
    void *p = nullptr;
    if (it != g_pin_free.end()) {
      p = it->second;
      g_pin_free.erase(it);
      g_pin_held -= len;
    } else if (hipHostMalloc(&p, len, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
    }
    if (p) {
      g_out_pin[p] = len;
      return p;
    }
    if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] note: no pinned block of %zu MB for a result array (held %zu MB); pageable\n", len >> 20, g_pin_held >> 20);
  }
  void *p = big_alloc(bytes);
  std::lock_guard<std::mutex> lk(g_out_mu);
  g_out_big[p] = bytes;
  return p;
}
void out_free(void *p) {
  if (!p) return;
  results_wait_if(p);   // (an array still being written by the pending record copy)
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(g_out_mu);
    auto ip = g_out_pin.find(p);
    if (ip != g_out_pin.end()) {
      const size_t len = ip->second;
      g_out_pin.erase(ip);
      if (g_pin_held + len <= PIN_CAP && ctx().ready) g_pin_free.emplace(len, p), g_pin_held += len;
      else (void)hipHostFree(p);
      return;
    }
    auto it = g_out_big.find(p);
    if (it != g_out_big.end()) bytes = it->second, g_out_big.erase(it);
  }
  if (bytes) big_free(p, bytes);
  else free(p);
}
static void pin_pool_trim() {
  std::lock_guard<std::mutex> lk(g_out_mu);
  for (auto &kv : g_pin_free) (void)hipHostFree(kv.second);
  g_pin_free.clear();
  g_pin_held = 0;
}

// ---- housekeeping thread -----------------------------------------------------------------------------------
namespace {
struct Reaper {
  std::mutex mu;
  std::condition_variable cv, idle;
  std::deque<std::function<void()>> q;
  bool busy = false, stop = false;
  std::thread th;
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return stop || !q.empty(); });
      if (q.empty()) return;
      std::function<void()> fn = std::move(q.front());
      q.pop_front();
      busy = true;
      lk.unlock();
      fn();
      fn = nullptr;
      lk.lock();
      busy = false;
      idle.notify_all();
    }
  }
  void push(std::function<void()> fn) {
    {
      std::unique_lock<std::mutex> lk(mu);
      if (q.size() < 4) {
        if (!th.joinable()) th = std::thread([this] { loop(); });
        q.push_back(std::move(fn));
        cv.notify_one();
        return;
      }
    }
    fn();  // the thread is behind: do it here rather than pile up memory
  }
  void drain() {
    std::unique_lock<std::mutex> lk(mu);
    idle.wait(lk, [&] { return q.empty() && !busy; });
  }
  ~Reaper() {
    {
      std::unique_lock<std::mutex> lk(mu);
      stop = true;
      cv.notify_all();
    }
    if (th.joinable()) th.join();
  }
};
Reaper &reaper() {
  static Reaper r;
  return r;
}
}  // namespace
void defer_destroy(std::function<void()> fn) {
  static const bool inline_only = false;
  if (inline_only) fn();
  else reaper().push(std::move(fn));
}
void drain_deferred() { reaper().drain(); }

// ---- device block cache ----------------------------------------------------------------------------------
// Released blocks are kept by size class for the next user (see pgx_internal.h).  Round 5 (the HBM ledger of a full-size configs[3] step):
//  * a request is served by a free block of its class OR UP TO ONE EIGHTH LARGER: table sizes move a little from chunk to chunk, and
//    with exact classes every multi-GB table left a sibling of the neighbouring class behind;
//  * blocks age: dev_cache_age() -- called where a stage starts, the GPU idle -- gives blocks of 64 MiB and more that two whole stages
//    did not ask for back to the driver (the tables of a first, too small attempt used to stay cached for the life of the process);
//  * every live block carries the tag of the scope that allocated it (MemTag): pgx_mem_ledger() reports the bytes by tag at the PEAK.
static std::mutex g_dev_mu;
struct FreeBlock {
  void *p;
  uint64_t age;
};
static std::multimap<size_t, FreeBlock> g_dev_free;  // size class -> cached blocks
struct LiveBlock {
  size_t cls;
  const char *tag;
};
static std::map<void *, LiveBlock> g_dev_live;       // block -> its size class, who asked for it
static uint64_t g_dev_age = 0;
static size_t g_live_bytes = 0, g_free_bytes = 0, g_peak_bytes = 0;
static std::map<std::string, size_t> g_by_tag, g_peak_by_tag;
static size_t g_peak_free = 0;
static thread_local const char *g_tag = "other";
MemTag::MemTag(const char *t) : prev(g_tag) { g_tag = t; }
MemTag::~MemTag() { g_tag = prev; }
static size_t size_class(size_t bytes) {           // powers of two up to 1 MiB, then eighths of a power of two (<= 12.5 % waste)
  size_t c = 256;
  while (c < bytes && c < (1u << 20)) c <<= 1;
  if (c >= bytes) return c;
  size_t p2 = (size_t)1 << 20;
  while (p2 * 2 <= bytes) p2 <<= 1;
  const size_t step = p2 >> 3;
  return (bytes + step - 1) / step * step;
}
static void drop_all_free_locked() {
  for (auto &kv : g_dev_free) (void)hipFree(kv.second.p);
  g_dev_free.clear();
  g_free_bytes = 0;
}
void *dev_alloc(size_t bytes) {
  const size_t c = size_class(bytes);
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_dev_free.lower_bound(c);
  void *p = nullptr;
  size_t got = c;
  if (it != g_dev_free.end() && it->first <= c + (c >> 3)) {
    p = it->second.p, got = it->first;
    g_dev_free.erase(it);
    g_free_bytes -= got;
  } else {
    hipError_t e = hipMalloc(&p, c);
    if (e != hipSuccess) {  // make room: give the cached blocks back and retry once
      drop_all_free_locked();
      (void)hipGetLastError();
      PGX_HIP(hipMalloc(&p, c));
    }
  }
  g_dev_live[p] = LiveBlock{got, g_tag};
  g_live_bytes += got, g_by_tag[g_tag] += got;
  if (g_live_bytes > g_peak_bytes) g_peak_bytes = g_live_bytes, g_peak_by_tag = g_by_tag, g_peak_free = g_free_bytes;
  return p;
}
void dev_release(void *p) {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_dev_live.find(p);
  if (it == g_dev_live.end()) return;
  g_dev_free.emplace(it->second.cls, FreeBlock{p, g_dev_age});
  g_live_bytes -= it->second.cls, g_free_bytes += it->second.cls, g_by_tag[it->second.tag] -= it->second.cls;
  g_dev_live.erase(it);
}
void dev_cache_trim() {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  drop_all_free_locked();
}
size_t dev_cache_free_bytes() {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  return g_free_bytes;
}
void dev_cache_age() {
  std::lock_guard<std::mutex> lk(g_dev_mu);
  ++g_dev_age;
  for (auto it = g_dev_free.begin(); it != g_dev_free.end();) {
    if (it->first >= ((size_t)64 << 20) && it->second.age + 2 < g_dev_age) {
      (void)hipFree(it->second.p);
      g_free_bytes -= it->first;
      it = g_dev_free.erase(it);
    } else {
      ++it;
    }
  }
}

// ---- shutdown hooks / generations --------------------------------------------------------------------------
static std::vector<void (*)()> &shutdown_hooks() {
  static std::vector<void (*)()> v;
  return v;
}
void on_shutdown(void (*fn)()) { shutdown_hooks().push_back(fn); }
uint64_t &index_generation() {
  static uint64_t g = 1;
  return g;
}

// ---- persistent workspace --------------------------------------------------------------------------------
static std::map<std::string, DevBuf<uint8_t>> g_ws;
void *ws_raw(const char *name, size_t bytes) {
  DevBuf<uint8_t> &b = g_ws[name];
  if (b.n < bytes) {
    (void)hipStreamSynchronize(ctx().stream);
    MemTag mem_tag("workspaces");
    b.alloc(bytes + (bytes >> 3) + 4096);
  }
  return b.p;
}

bool ws_contains(const void *p) {
  for (const auto &kv : g_ws)
    if (kv.second.p && (const uint8_t *)p >= kv.second.p && (const uint8_t *)p < kv.second.p + kv.second.n) return true;
  return false;
}

// ---- timing ----------------------------------------------------------------------------------------------
struct TimeAcc {
  double ms = 0;
  uint64_t launches = 0, units = 0;
};
struct Pending {
  std::string name;
  uint64_t units;
  hipEvent_t e0, e1;
};
static std::map<std::string, TimeAcc> g_time;
static std::vector<Pending> g_pending;
static std::mutex g_time_mu;   // (timers are also closed on the housekeeping thread)

KernelTimer::KernelTimer(const char *nm, uint64_t u) : name(nm), units(u) {
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, ctx().stream);
}
KernelTimer::~KernelTimer() {
  (void)hipEventRecord(e1, ctx().stream);
  std::lock_guard<std::mutex> lk(g_time_mu);
  g_pending.push_back(Pending{name, units, e0, e1});
}
void timing_flush() {
  {
    std::lock_guard<std::mutex> lk(g_time_mu);
    if (g_pending.empty()) return;
  }
  (void)hipStreamSynchronize(ctx().stream);
  std::lock_guard<std::mutex> lk(g_time_mu);
  std::vector<Pending> later;
  for (auto &p : g_pending) {
    if (hipEventQuery(p.e1) == hipErrorNotReady) {   // (an interval of the other thread's stream that is still open)
      later.push_back(p);
      continue;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      auto &a = g_time[p.name];
      a.ms += ms, a.launches += 1, a.units += p.units;
    }
    (void)hipEventDestroy(p.e0);
    (void)hipEventDestroy(p.e1);
  }
  g_pending.swap(later);
}

// ---- files -----------------------------------------------------------------------------------------------
int load_idx(const char *path, std::vector<uint32_t> &rid, std::vector<uint32_t> &rlen, std::vector<uint64_t> &roff) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char name[256];
  unsigned r, l;
  unsigned long o;
  // same tokenisation as the reference's fscanf("%u %255s %u %lu") (src/shmr_utils.c:259-260)
  while (fscanf(f, "%u %255s %u %lu", &r, name, &l, &o) == 4) rid.push_back(r), rlen.push_back(l), roff.push_back(o);
  fclose(f);
  return 0;
}
bool read_file(const std::string &path, std::vector<uint8_t> &out) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(sz > 0 ? (size_t)sz : 0);
  bool ok = sz <= 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
  fclose(f);
  return ok;
}

}  // namespace pgx

using namespace pgx;

#define PGX_GUARD_BEGIN try {
#define PGX_GUARD_END                          \
  }                                            \
  catch (const pgx::Fail &f) { return f.code; } \
  catch (const std::bad_alloc &) {             \
    pgx::set_error("out of host memory");      \
    return PGX_ENOMEM;                         \
  }                                            \
  return PGX_OK;

extern "C" {

const char *pgx_last_error(void) { return pgx::g_err.c_str(); }
const char *pgx_version(void) { return "pgx 0.1 (gfx950)"; }
void pgx_free(void *p) { pgx::out_free(p); }
int pgx_results_async(int on) {
  const int was = pgx::results_async() ? 1 : 0;
  if (!on) {
    try {   // (a failed copy must not leave the C-ABI as a C++ exception: ADVICE r4; pgx_last_error has the text, pgx_results_wait the code)
      pgx::results_wait();
    } catch (const pgx::Fail &) {
    }
  }
  pgx::results_async() = on != 0;
  return was;
}
int pgx_results_wait(void) {
  try {
    pgx::results_wait();
  } catch (const pgx::Fail &f) {
    return f.code;
  }
  return PGX_OK;
}

int pgx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static double wall_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int pgx_init(int device) {
  PGX_GUARD_BEGIN
  Context &c = ctx();
  if (c.ready && c.device == device) return PGX_OK;
  const double t_init = wall_ms();
  // one context per process: workspaces, the block cache and every resident seqdb live on the device of the first call
  PGX_REQUIRE(!c.ready, PGX_ESTATE, "pgx_init(%d): this process already runs on device %d (call pgx_shutdown first)", device, c.device);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  PGX_REQUIRE(e == hipSuccess && n > 0, PGX_EHIP, "no HIP device visible (%s)", hipGetErrorString(e));
  PGX_REQUIRE(device >= 0 && device < n, PGX_EARG, "device %d out of range (have %d)", device, n);
  PGX_HIP(hipSetDevice(device));
  if (c.stream) (void)hipStreamDestroy(c.stream);
  PGX_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  PGX_HIP(hipGetDeviceProperties(&prop, device));
  c.num_cu = prop.multiProcessorCount;
  c.device = device;
  c.ready = true;
  if (getenv("PGX_TRACE")) fprintf(stderr, "[pgx] init: device %d (HIP runtime + context + stream) in %.1f ms\n", device, wall_ms() - t_init);
  PGX_GUARD_END
}

void pgx_shutdown(void) {
  Context &c = ctx();
  drain_deferred();
  big_pool_trim();
  pin_pool_trim();
  for (auto fn : shutdown_hooks()) fn();   // plan caches / held buffers of the stages (nothing may outlive the workspaces)
  g_ws.clear();
  ++index_generation();
  if (c.stream) (void)hipStreamSynchronize(c.stream);
  dev_cache_trim();
  if (c.stream) (void)hipStreamDestroy(c.stream);
  c.stream = nullptr;
  c.ready = false;
}

int pgx_timing_get(const char *kernel, double *total_ms, uint64_t *launches, uint64_t *units) {
  timing_flush();
  std::lock_guard<std::mutex> lk(g_time_mu);
  auto it = g_time.find(kernel ? kernel : "");
  if (it == g_time.end()) {
    if (total_ms) *total_ms = 0;
    if (launches) *launches = 0;
    if (units) *units = 0;
    return PGX_EARG;
  }
  if (total_ms) *total_ms = it->second.ms;
  if (launches) *launches = it->second.launches;
  if (units) *units = it->second.units;
  return PGX_OK;
}
void pgx_timing_reset(void) {
  timing_flush();
  std::lock_guard<std::mutex> lk(g_time_mu);
  g_time.clear();
}

int pgx_mem_ledger(char *buf, size_t cap, int reset_peak) {
  std::string o = "{";
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    char t[256];
    snprintf(t, sizeof(t), "\"live_bytes\": %zu, \"cached_free_bytes\": %zu, \"peak_live_bytes\": %zu, \"cached_free_bytes_at_peak\": %zu, \"peak_by_tag\": {",
             g_live_bytes, g_free_bytes, g_peak_bytes, g_peak_free);
    o += t;
    bool first = true;
    for (const auto &kv : g_peak_by_tag) {
      if (!kv.second) continue;
      snprintf(t, sizeof(t), "%s\"%s\": %zu", first ? "" : ", ", kv.first.c_str(), kv.second);
      o += t, first = false;
    }
    o += "}, \"workspaces\": {";
    first = true;
    for (const auto &kv : g_ws) {
      if (kv.second.n < ((size_t)1 << 20)) continue;
      snprintf(t, sizeof(t), "%s\"%s\": %zu", first ? "" : ", ", kv.first.c_str(), kv.second.n);
      o += t, first = false;
    }
    o += "}";
    if (reset_peak) g_peak_bytes = g_live_bytes, g_peak_by_tag = g_by_tag, g_peak_free = g_free_bytes;
  }
  size_t fr = 0, tot = 0;
  if (ctx().ready && hipMemGetInfo(&fr, &tot) == hipSuccess) {
    char t[128];
    snprintf(t, sizeof(t), ", \"device_used_bytes\": %zu, \"device_total_bytes\": %zu", tot - fr, tot);
    o += t;
  }
  o += "}";
  if (buf && cap) {
    const size_t n = std::min(cap - 1, o.size());
    memcpy(buf, o.data(), n);
    buf[n] = 0;
  }
  return (int)o.size();
}

// ---- resident seqdb --------------------------------------------------------------------------------------
// The seqdb file straight into HBM.  One thread copying out of the page cache moves ~6 GB/s, an eighth of what the host link takes
// (round 2: 4.5 GB in 0.75 s, twice per pipeline -- each stage is its own process, as in pg_run.py), so several reader threads
// fill a ring of pinned pieces (pread, out of order) while the calling thread uploads the pieces IN order as they complete; the
// host never holds more than NBUF pieces.  PGX_LOAD_THREADS: test knob (tools/e2e.py).
static void upload_file_pieces(const char *path, uint8_t *d_dst, size_t nbytes) {
  const int fd = open(path, O_RDONLY);
  PGX_REQUIRE(fd >= 0, PGX_EIO, "cannot read %s", path);
  const size_t P = (size_t)32 << 20;
  const size_t npieces = (nbytes + P - 1) / P;
  // (6 readers move 8.3 GB/s out of the page cache -- 11.2 s for the 93 GB of a human-scale database, which is what `pgx_cli serve` starts with;
  //  from 8 GB on: 20 readers)
  const int NT = (int)std::min<size_t>(npieces, (size_t)(getenv("PGX_LOAD_THREADS") ? std::max(1, atoi(getenv("PGX_LOAD_THREADS"))) : nbytes >= ((size_t)8 << 30) ? 20 : 6));
  const int NBUF = (int)std::min<size_t>(npieces, (size_t)NT + 2);
  std::vector<uint8_t *> pin(NBUF, nullptr);
  std::vector<hipEvent_t> ev(NBUF, nullptr);
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> state(npieces, 0);   // 0: not read yet, 1: in its buffer, -1: read error
  size_t next_piece = 0, uploaded = 0;  // next piece a reader takes; pieces whose upload has been issued
  bool stop = false;
  std::vector<std::thread> readers;
  auto cleanup = [&]() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto &t : readers) t.join();
    (void)hipStreamSynchronize(ctx().stream);
    for (int k = 0; k < NBUF; ++k) {
      if (pin[k]) (void)hipHostFree(pin[k]);
      if (ev[k]) (void)hipEventDestroy(ev[k]);
    }
    close(fd);
  };
  try {
    for (int k = 0; k < NBUF; ++k) {
      PGX_HIP(hipHostMalloc((void **)&pin[k], std::min(P, std::max<size_t>(nbytes, 1)), hipHostMallocDefault));
      PGX_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    }
    // a reader: take the next piece, wait until its buffer (piece % NBUF) is free -- i.e. the piece NBUF before it has been uploaded
    // AND that upload has completed (the uploader publishes `uploaded` only after the event of that piece's copy was waited for
    // lazily: see below) --, read it, publish it
    std::vector<char> upload_done(npieces, 0);
    for (int t = 0; t < NT; ++t)
      readers.emplace_back([&]() {
        for (;;) {
          size_t p;
          {
            std::unique_lock<std::mutex> lk(mu);
            if (stop || next_piece >= npieces) return;
            p = next_piece++;
            cv.wait(lk, [&] { return stop || p < (size_t)NBUF || upload_done[p - NBUF]; });
            if (stop) return;
          }
          const size_t off = p * P, n = std::min(P, nbytes - off);
          uint8_t *dst = pin[p % NBUF];
          bool ok = true;
          for (size_t got = 0; got < n;) {
            const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
            if (r <= 0) {
              ok = false;
              break;
            }
            got += (size_t)r;
          }
          {
            std::lock_guard<std::mutex> lk(mu);
            state[p] = ok ? 1 : -1;
          }
          cv.notify_all();
        }
      });
    // the uploader (this thread): pieces in order; a piece's buffer is released to the readers once its copy has completed
    size_t released = 0, marked = 0;   // pieces [0, released) have completed their upload; [0, marked) are published to the readers
    for (size_t p = 0; p < npieces; ++p) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state[p] != 0; });
        PGX_REQUIRE(state[p] > 0, PGX_EIO, "short read from %s (piece %zu of %zu)", path, p, npieces);
      }
      const size_t off = p * P, n = std::min(P, nbytes - off);
      PGX_HIP(hipMemcpyAsync(d_dst + off, pin[p % NBUF], n, hipMemcpyHostToDevice, ctx().stream));
      PGX_HIP(hipEventRecord(ev[p % NBUF], ctx().stream));
      uploaded = p + 1;
      // release every earlier piece whose copy is known to be complete (query, do not block: the readers only need the buffers NBUF pieces ahead)
      bool any = false;
      while (released < uploaded && hipEventQuery(ev[released % NBUF]) == hipSuccess) ++released, any = true;
      if (!any && uploaded - released >= (size_t)NBUF - 1) {   // every buffer is in flight: wait for the oldest copy
        PGX_HIP(hipEventSynchronize(ev[released % NBUF]));
        ++released, any = true;
      }
      if (any) {
        {
          std::lock_guard<std::mutex> lk(mu);
          for (; marked < released; ++marked) upload_done[marked] = 1;
        }
        cv.notify_all();
      }
    }
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
}

// seqdb: host bytes, device bytes (from_device), or -- file != nullptr -- the path of the .seqdb file (nbytes = its size)
static int seqdb_upload_impl(const uint8_t *seqdb, size_t nbytes, const uint32_t *rid, const uint32_t *rlen,
                             const uint64_t *roff, uint32_t nreads, pgx_seqdb **out, bool from_device, const char *file = nullptr,
                             bool adopt = false) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(out && (nreads == 0 || (rid && rlen && roff)), PGX_EARG, "pgx_seqdb_upload: null argument");
  auto *db = new pgx_seqdb();
  db->rid.assign(rid, rid + nreads);
  db->rlen.assign(rlen, rlen + nreads);
  db->roff.assign(roff, roff + nreads);
  uint32_t max_rid = 0;
  for (uint32_t i = 0; i < nreads; ++i) {
    max_rid = std::max(max_rid, rid[i]);
    db->bases += rlen[i];
    if ((size_t)roff[i] + rlen[i] > nbytes) {
      delete db;
      PGX_REQUIRE(false, PGX_EARG, "read %u exceeds the seqdb (%zu bytes)", rid[i], nbytes);
    }
  }
  const size_t nr = nreads ? (size_t)max_rid + 1 : 0;
  db->rlen_by_rid.assign(nr, 0);
  db->roff_by_rid.assign(nr, 0);
  for (uint32_t i = 0; i < nreads; ++i)
    db->rlen_by_rid[rid[i]] = rlen[i], db->roff_by_rid[rid[i]] = roff[i], db->max_rlen = std::max(db->max_rlen, rlen[i]);
  db->nbytes = nbytes;
  try {
    if (adopt) {
      db->borrowed = true;
      db->d_seq.p = const_cast<uint8_t *>(seqdb), db->d_seq.n = nbytes + 1024;
    } else {
      MemTag mem_tag("seqdb.bytes");
      db->d_seq.alloc(nbytes + 1024);
    }
    PGX_HIP(hipMemsetAsync(db->d_seq.p + nbytes, 0, 1024, ctx().stream));
    if (adopt) {
    } else if (nbytes && file) upload_file_pieces(file, db->d_seq.p, nbytes);
    else if (nbytes) PGX_HIP(hipMemcpyAsync(db->d_seq.p, seqdb, nbytes, from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx().stream));
    db->d_roff.alloc(nr ? nr : 1);
    db->d_rlen.alloc(nr ? nr : 1);
    db->d_roff.upload(db->roff_by_rid.data(), nr);
    db->d_rlen.upload(db->rlen_by_rid.data(), nr);
    pgx::sync();
  } catch (...) {
    delete db;
    throw;
  }
  *out = db;
  PGX_GUARD_END
}

int pgx_seqdb_upload(const uint8_t *seqdb, size_t nbytes, const uint32_t *rid, const uint32_t *rlen,
                     const uint64_t *roff, uint32_t nreads, pgx_seqdb **out) {
  return seqdb_upload_impl(seqdb, nbytes, rid, rlen, roff, nreads, out, false);
}
int pgx_seqdb_upload_dev(const uint8_t *d_seqdb, size_t nbytes, const uint32_t *rid, const uint32_t *rlen,
                         const uint64_t *roff, uint32_t nreads, pgx_seqdb **out) {
  return seqdb_upload_impl(d_seqdb, nbytes, rid, rlen, roff, nreads, out, true);
}
int pgx_seqdb_adopt_dev(uint8_t *d_seqdb, size_t nbytes, size_t capacity, const uint32_t *rid, const uint32_t *rlen,
                        const uint64_t *roff, uint32_t nreads, pgx_seqdb **out) {
  if (!d_seqdb || capacity < nbytes + 1024 || ((uintptr_t)d_seqdb & 15u)) {
    pgx::set_error("pgx_seqdb_adopt_dev: need a 16-byte aligned device pointer with capacity >= nbytes + 1024 (the kernels' wide loads are "
                   "aligned 16-byte loads and read past the last read)");
    return PGX_EARG;
  }
  return seqdb_upload_impl(d_seqdb, nbytes, rid, rlen, roff, nreads, out, true, nullptr, true);
}
// event hand-over between the library's stream and another runtime's
static int stream_order(hipStream_t first, hipStream_t then) {
  PGX_GUARD_BEGIN
  require_ready();
  hipEvent_t ev;
  PGX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, first);
  if (e == hipSuccess) e = hipStreamWaitEvent(then, ev, 0);
  (void)hipEventDestroy(ev);   // (deferred by the runtime until the event has completed)
  PGX_HIP(e);
  PGX_GUARD_END
}
int pgx_stream_wait(void *other_stream) { return stream_order((hipStream_t)other_stream, ctx().stream); }
int pgx_stream_signal(void *other_stream) { return stream_order(ctx().stream, (hipStream_t)other_stream); }
int pgx_copy_dev(void *d_dst, const void *d_src, size_t nbytes) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE((d_dst && d_src) || nbytes == 0, PGX_EARG, "pgx_copy_dev: null argument");
  if (nbytes) PGX_HIP(hipMemcpyAsync(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice, ctx().stream));
  pgx::sync();
  PGX_GUARD_END
}

int pgx_seqdb_load(const char *prefix, pgx_seqdb **out) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(prefix && out, PGX_EARG, "pgx_seqdb_load: null argument");
  std::vector<uint32_t> rid, rlen;
  std::vector<uint64_t> roff;
  std::string p(prefix);
  const double t_load = wall_ms();
  PGX_REQUIRE(load_idx((p + ".idx").c_str(), rid, rlen, roff) == 0, PGX_EIO, "cannot open %s.idx", prefix);
  const double t_idx = wall_ms();
  const std::string dbpath = p + ".seqdb";
  struct stat sb;
  PGX_REQUIRE(stat(dbpath.c_str(), &sb) == 0 && S_ISREG(sb.st_mode), PGX_EIO, "cannot read %s.seqdb", prefix);
  int rc = seqdb_upload_impl(nullptr, (size_t)sb.st_size, rid.data(), rlen.data(), roff.data(), (uint32_t)rid.size(), out, false, dbpath.c_str());
  if (rc) return rc;
  if (getenv("PGX_TRACE"))
    fprintf(stderr, "[pgx] seqdb load: idx file (%zu reads) %.1f ms, %.2f GB file -> HBM %.1f ms = %.1f GB/s\n", rid.size(), t_idx - t_load,
            sb.st_size / 1e9, wall_ms() - t_idx, sb.st_size / 1e6 / (wall_ms() - t_idx));
  PGX_GUARD_END
}

// The byte seqdb out of HBM (round 6, VERDICT r5 task 5): once the 2-bit packs exist they carry the same information at a quarter of the
// size (pgx_pack.hip), and the kernels of the default path read them -- k_sketch_blk / k_sketch_wave (fused list form), k_align_ph,
// k_align1 / k_align1_list.  What cannot work from the packs keeps the bytes: a database with a read that holds an ambiguous base (no 2-bit
// code: those reads are sketched run by run and aligned nibble by nibble from the bytes) or a read beyond 65,535 bases (k_align4) is REFUSED
// (PGX_ESTATE, bytes kept, nothing changed).  After a release: pgx_index_resident* (w = 80, k = 16, levels 1 / 2, no L0 output),
// pgx_overlap_*, pgx_align_batch and the file-level forms over them work as before; entry points that need the bytes (other w / k, want_l0,
// pgx_sketch_batch) fail with PGX_ESTATE.  A buffer the library adopted (pgx_seqdb_adopt_dev) is no longer referenced: the caller may free it.
int pgx_seqdb_release_bytes(pgx_seqdb *db) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(db, PGX_EARG, "pgx_seqdb_release_bytes: null argument");
  if (!db->d_seq.p) return PGX_OK;   // (released already)
  PGX_REQUIRE(db->max_rlen <= 65535u, PGX_ESTATE, "pgx_seqdb_release_bytes: a read of %u bases (> 65,535) needs the byte-wise alignment kernel: bytes kept", db->max_rlen);
  PGX_REQUIRE(seq_packs(db) != nullptr, PGX_ESTATE, "pgx_seqdb_release_bytes: the 2-bit packs could not be built: bytes kept");
  PGX_REQUIRE(db->n_flagged_reads == 0, PGX_ESTATE, "pgx_seqdb_release_bytes: %u reads hold an ambiguous base (no 2-bit code): bytes kept", db->n_flagged_reads);
  pgx::sync();
  if (db->borrowed) db->d_seq.p = nullptr, db->d_seq.n = 0;
  else db->d_seq.release();
  PGX_GUARD_END
}
int pgx_seqdb_has_bytes(const pgx_seqdb *db) { return db && db->d_seq.p ? 1 : 0; }

void pgx_seqdb_free(pgx_seqdb *db) {
  if (db) {   // what the library kept for the chunks of a job on this database goes with it
    try {
      pgx::count_cache_drop();
      pgx::replay_forget_sizes();
      pgx::list_stash_clear();
    } catch (...) {
    }
  }
  delete db;
}
uint64_t pgx_seqdb_bases(const pgx_seqdb *db) { return db ? db->bases : 0; }
uint32_t pgx_seqdb_reads(const pgx_seqdb *db) { return db ? (uint32_t)db->rid.size() : 0; }

// ---- batch level -----------------------------------------------------------------------------------------
int pgx_sketch_batch(pgx_seqdb *db, const uint32_t *read_slots, uint32_t n, int w, int k, pgx_mm128 **out,
                     size_t *n_out) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(db && out && n_out, PGX_EARG, "pgx_sketch_batch: null argument");
  PGX_REQUIRE(w > 0 && w < 256 && k > 0 && k <= 28, PGX_EARG, "need 0<w<256, 0<k<=28 (src/mm_sketch.c:77-78)");
  std::vector<ReadDesc> reads(n);
  for (uint32_t i = 0; i < n; ++i) {
    PGX_REQUIRE(read_slots[i] < db->rid.size(), PGX_EARG, "read slot %u out of range", read_slots[i]);
    const uint32_t s = read_slots[i];
    PGX_REQUIRE(db->rlen[s] > 0, PGX_EARG, "read slot %u is empty (mm_sketch asserts len > 0)", s);
    reads[i] = ReadDesc{db->roff[s], db->rlen[s], db->rid[s]};
  }
  DevBuf<pgx_mm128> d;
  size_t m = 0;
  dev_sketch(db, reads, w, k, d, m, nullptr);
  std::vector<pgx_mm128> h(m);
  d.download(h.data(), m);
  pgx::sync();
  *out = host_copy(h);
  *n_out = m;
  PGX_GUARD_END
}

int pgx_reduce_batch(const pgx_mm128 *in, size_t n, int rs, pgx_mm128 **out, size_t *n_out) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(out && n_out && (n == 0 || in), PGX_EARG, "pgx_reduce_batch: null argument");
  PGX_REQUIRE(rs > 0 && rs < 256, PGX_EARG, "reduction factor must be 1..255");
  PGX_REQUIRE(n < (1ULL << 31), PGX_EARG, "list too long");
  DevBuf<pgx_mm128> d_in(n), d_out;
  d_in.upload(in, n);
  size_t m = 0;
  dev_reduce(d_in.p, n, rs, d_out, m);
  std::vector<pgx_mm128> h(m);
  d_out.download(h.data(), m);
  pgx::sync();
  *out = host_copy(h);
  *n_out = m;
  PGX_GUARD_END
}

int pgx_count_batch(const pgx_mm128 *in, size_t n, pgx_mm_count **out, size_t *n_out) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(out && n_out && (n == 0 || in), PGX_EARG, "pgx_count_batch: null argument");
  PGX_REQUIRE(n < (1ULL << 31), PGX_EARG, "list too long");
  DevBuf<pgx_mm128> d_in(n);
  DevBuf<pgx_mm_count> d_out;
  d_in.upload(in, n);
  size_t m = 0;
  dev_count(d_in.p, n, 56, d_out, m);
  std::vector<pgx_mm_count> h(m);
  d_out.download(h.data(), m);
  pgx::sync();
  *out = host_copy(h);
  *n_out = m;
  PGX_GUARD_END
}

int pgx_align_batch(pgx_seqdb *db, const pgx_align_key *keys, size_t n, int band, pgx_match *out) {
  PGX_GUARD_BEGIN
  require_ready();
  PGX_REQUIRE(db && (n == 0 || (keys && out)), PGX_EARG, "pgx_align_batch: null argument");
  PGX_REQUIRE(band > 0 && band < (1 << 20), PGX_EARG, "bad band");
  for (size_t i = 0; i < n; ++i) {
    const auto &k = keys[i];
    PGX_REQUIRE(k.rid0 < db->rlen_by_rid.size() && k.rid1 < db->rlen_by_rid.size() && k.q_off <= db->rlen_by_rid[k.rid0],
                PGX_EARG, "alignment key %zu out of range", i);
  }
  DevBuf<pgx_align_key> d_keys(n);
  DevBuf<pgx_match> d_out(n);
  d_keys.upload(keys, n);
  dev_align(db, d_keys.p, n, band, d_out.p);
  d_out.download(out, n);
  pgx::sync();
  PGX_GUARD_END
}

}  // extern "C"
