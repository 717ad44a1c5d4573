// pgx_overlap.cpp -- the overlap stage: what main() of /root/reference/src/shmr_overlap.c:233-419 does for one
// chunk.  Division of labour:
//   GPU  : every ovlp_match (src/DWmatch.c:66-204) -- >90 % of the reference's time -- in bulk batches (k_align)
//   host : the parts whose RESULT ORDER is defined by sequential containers in the reference and therefore has to
//          be replayed in order: shimmer-pair records (build_map, src/shmr_utils.c:295-404), the klib-khash slot
//          order that defines the bucket visit order (src/khash.h:232-336; shmr_overlap.c:206-215), the stable
//          position sort (shmr_overlap.c:46-50,217) and the greedy best-n selection with its process-global
//          seen-pair table (shmr_overlap.c:52-180).
// The greedy is order dependent but ovlp_match is a pure function of (rid0, dir0, q_off, rid1, dir1, band), so
// the host replays the greedy optimistically ("unknown alignment => assume accepted overlap"), collects the
// alignments it asked for, runs them on the GPU, and replays with the true results until a replay asks for
// nothing new.  That last replay used only true results, hence equals the reference's record sequence.
#include <glob.h>
#include <sched.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

#include <limits.h>
#include <unistd.h>

#include "pgx_internal.h"
#include "pgx_khash.h"

using namespace pgx;

namespace {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------------------------------------------------
// small open-addressing maps (host orchestration containers; layout has no observable effect)
// ---------------------------------------------------------------------------------------------------------
static inline uint64_t mix(uint64_t h) {
  h ^= h >> 33, h *= 0xff51afd7ed558ccdULL, h ^= h >> 33, h *= 0xc4ceb9fe1a85ec53ULL, h ^= h >> 33;
  return h;
}

template <typename V>
struct U64Map {
  std::vector<uint64_t> keys;
  std::vector<V> vals;
  std::vector<uint8_t> used;
  size_t size = 0, cap = 0;
  void reserve_pow2(size_t c) {
    cap = 16;
    while (cap < c) cap <<= 1;
    keys.assign(cap, 0), vals.assign(cap, V()), used.assign(cap, 0), size = 0;
  }
  void clear() {
    std::fill(used.begin(), used.end(), 0);
    size = 0;
  }
  void grow() {
    std::vector<uint64_t> ok;
    std::vector<V> ov;
    std::vector<uint8_t> ou;
    ok.swap(keys), ov.swap(vals), ou.swap(used);
    const size_t oc = cap;
    reserve_pow2(oc ? oc * 2 : 16);
    for (size_t i = 0; i < oc; ++i)
      if (ou[i]) *slot(ok[i]) = ov[i];
  }
  V *find(uint64_t k) {
    if (!cap) return nullptr;
    size_t i = mix(k) & (cap - 1);
    while (used[i]) {
      if (keys[i] == k) return &vals[i];
      i = (i + 1) & (cap - 1);
    }
    return nullptr;
  }
  V *slot(uint64_t k) {  // find or insert (value default-initialised on insert)
    if ((size + 1) * 2 > cap) grow();
    size_t i = mix(k) & (cap - 1);
    while (used[i]) {
      if (keys[i] == k) return &vals[i];
      i = (i + 1) & (cap - 1);
    }
    used[i] = 1, keys[i] = k, vals[i] = V(), ++size;
    return &vals[i];
  }
};

struct AKey {
  uint64_t a, b;  // a = rid0<<32|rid1 ; b = q_off<<2|dir0<<1|dir1
  bool operator==(const AKey &o) const { return a == o.a && b == o.b; }
};
struct AKeyMap {  // alignment memo: key -> index into the result array.  One 24-byte slot per probe (one cache miss).
  struct Slot {
    AKey k;
    uint32_t val;
    uint32_t used;
  };
  std::vector<Slot> slots;
  size_t size = 0, cap = 0;
  void init(size_t c) {
    cap = 1024;
    while (cap < c) cap <<= 1;
    slots.assign(cap, Slot{AKey{0, 0}, 0, 0});
    size = 0;
  }
  void grow() {
    AKeyMap n;
    n.init(cap * 2);
    for (size_t i = 0; i < cap; ++i)
      if (slots[i].used) *n.slot(slots[i].k, nullptr) = slots[i].val;
    *this = std::move(n);
  }
  uint32_t *slot(const AKey &k, bool *inserted) {
    if ((size + 1) * 2 > cap) grow();
    size_t i = mix(k.a ^ mix(k.b)) & (cap - 1);
    while (slots[i].used) {
      if (slots[i].k == k) {
        if (inserted) *inserted = false;
        return &slots[i].val;
      }
      i = (i + 1) & (cap - 1);
    }
    slots[i].used = 1, slots[i].k = k, slots[i].val = 0, ++size;
    if (inserted) *inserted = true;
    return &slots[i].val;
  }
};

// read pair -> dense id, 16-byte slots
struct PairMap {
  struct Slot {
    uint64_t key;   // ~0 = empty (a pair key has min rid in the high half, so ~0 cannot occur)
    uint32_t pid;
    uint32_t pad;
  };
  std::vector<Slot> slots;
  size_t size = 0, cap = 0;
  void init(size_t c) {
    cap = 1024;
    while (cap < c) cap <<= 1;
    slots.assign(cap, Slot{~0ULL, 0, 0});
    size = 0;
  }
  void grow() {
    PairMap n;
    n.init(cap * 2);
    for (size_t i = 0; i < cap; ++i)
      if (slots[i].key != ~0ULL) {
        bool f;
        *n.slot(slots[i].key, &f) = slots[i].pid;
      }
    *this = std::move(n);
  }
  uint32_t *slot(uint64_t k, bool *fresh) {
    if ((size + 1) * 2 > cap) grow();
    size_t i = mix(k) & (cap - 1);
    while (slots[i].key != ~0ULL) {
      if (slots[i].key == k) {
        *fresh = false;
        return &slots[i].pid;
      }
      i = (i + 1) & (cap - 1);
    }
    slots[i].key = k, ++size;
    *fresh = true;
    return &slots[i].pid;
  }
};

static inline uint32_t pos_of(uint64_t y) { return (uint32_t)((y & 0xFFFFFFFFu) >> 1); }

// ---------------------------------------------------------------------------------------------------------
// bucket visit list: ascending slot order of both table levels, buckets with 2 < n <= ovlp_upper, each sorted
// stably by descending position (shmr_overlap.c:206-217)
// ---------------------------------------------------------------------------------------------------------
struct Entry {
  uint32_t rid, pos1;  // pos1 = lastPos + 1
  uint64_t y0;
  uint8_t dir;
};
struct Visit {
  std::vector<uint64_t> start;  // bucket b covers entries [start[b], start[b+1])
  HostArray<Entry> entries;
  // the ids-only form (device replay): the join's bucket id of every visited bucket, in visit order
  HostArray<uint32_t> bids;
  size_t n_buckets = 0, n_entries = 0;
  // ids-only form with the placement left to the GPU (dev_place_bids): the groups in visit order, where each one's bucket ids sit in
  // `ids_all` and where they go
  bool on_device = false;
  HostArray<uint32_t> ids_all, psrc, pcnt;
  HostArray<uint64_t> pdst;
  size_t n_groups = 0;
};

// The replay threads hammer one shared table with locked operations: spread over both sockets they run ~1.7x slower than
// on one (measured, 2 x EPYC 9575F).  NodePin keeps the caller and the threads it starts on the memory node the caller
// is running on, for the lifetime of the object (PGX_PIN=0 disables it).
// the CPUs of the memory node this process works on (false: pinning is off or not possible)
static bool choose_node(cpu_set_t &saved, cpu_set_t &node) {
    if (const char *e = getenv("PGX_PIN"))
      if (atoi(e) == 0) return false;
    if (sched_getaffinity(0, sizeof(saved), &saved) != 0) return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    // the memory nodes and the CPUs of each that this process may use
    std::vector<cpu_set_t> nodes;
    int mine = -1;
    for (int nd = 0; nd < 64; ++nd) {
      char path[96];
      snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", nd);
      FILE *f = fopen(path, "r");
      if (!f) break;
      char line[4096];
      const bool ok = fgets(line, sizeof(line), f) != nullptr;
      fclose(f);
      if (!ok) continue;
      cpu_set_t set;
      CPU_ZERO(&set);
      for (char *q = line; *q && *q != '\n';) {  // "0-63,128-191"
        char *end;
        const long a = strtol(q, &end, 10);
        long b = a;
        if (end == q) break;
        if (*end == '-') b = strtol(end + 1, &end, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
          if (CPU_ISSET(c, &saved)) {
            CPU_SET(c, &set);
            if (c == cpu) mine = (int)nodes.size();
          }
        q = *end == ',' ? end + 1 : end;
      }
      if (CPU_COUNT(&set) >= 2) nodes.push_back(set);
      else if (mine == (int)nodes.size()) mine = -1;
    }
    if (nodes.empty()) return false;
    // one process per GPU (torchrun exports LOCAL_RANK / LOCAL_WORLD_SIZE): spread the ranks over the nodes evenly instead
    // of wherever their main threads happen to run; a single process stays where it is
    int pick = mine;
    const char *lr = getenv("LOCAL_RANK"), *lw = getenv("LOCAL_WORLD_SIZE");
    if (lr && lw && atoi(lw) > 1) pick = (int)((long)atoi(lr) * (long)nodes.size() / std::max(1, atoi(lw))) % (int)nodes.size();
    if (pick < 0) return false;
    node = nodes[(size_t)pick];
    return true;
}
struct NodePin {
  cpu_set_t saved, node;
  bool active = false;
  NodePin() {
    if (!choose_node(saved, node)) return;
    active = sched_setaffinity(0, sizeof(node), &node) == 0;  // threads created from here on inherit the mask
  }
  ~NodePin() {
    if (active) sched_setaffinity(0, sizeof(saved), &saved);
  }
  NodePin(const NodePin &) = delete;
  NodePin &operator=(const NodePin &) = delete;
};

// A persistent team of host threads.  The stage runs dozens of short parallel regions per call (replay rounds of a
// fraction of a millisecond, settle, collect, the table replays); creating 15 threads for each costs more than the work
// at the small end.  Workers spin briefly for the next region and then sleep; every region starts by adopting the
// caller's CPU affinity (see NodePin).
class WorkTeam {
 public:
  template <typename F>
  void run(unsigned nthr, F &&fn) {  // fn(thread index) on nthr threads, the caller being thread 0
    if (nthr <= 1) {
      fn(0);
      return;
    }
    std::lock_guard<std::mutex> serial(run_mu_);
    cpu_set_t mask;
    const bool have_mask = sched_getaffinity(0, sizeof(mask), &mask) == 0;
    {
      std::lock_guard<std::mutex> lk(mu_);
      while (th_.size() < nthr - 1) {
        const unsigned id = (unsigned)th_.size();
        th_.emplace_back([this, id] { worker(id); });
      }
      job_.call = [](void *c, unsigned ti) { (*static_cast<std::remove_reference_t<F> *>(c))(ti); };
      job_.ctx = (void *)&fn;
      job_.workers = nthr - 1;
      job_.mask = mask, job_.have_mask = have_mask;
      remaining_.store(nthr - 1, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
      if (sleepers_) cv_.notify_all();
    }
    fn(0);
    for (unsigned spins = 0; remaining_.load(std::memory_order_acquire); ++spins)
      if (spins < 4096) __builtin_ia32_pause();
      else std::this_thread::yield();
  }
  ~WorkTeam() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_.fetch_add(1, std::memory_order_release);
      cv_.notify_all();
    }
    for (auto &t : th_) t.join();
  }

 private:
  struct Job {
    void (*call)(void *, unsigned) = nullptr;
    void *ctx = nullptr;
    unsigned workers = 0;
    cpu_set_t mask;
    bool have_mask = false;
  };
  void worker(unsigned id) {
    uint64_t seen = 0;
    cpu_set_t mine;
    CPU_ZERO(&mine);
    for (;;) {
      for (unsigned spins = 0; gen_.load(std::memory_order_acquire) == seen && spins < 20000; ++spins) __builtin_ia32_pause();
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (gen_.load(std::memory_order_acquire) == seen) {
          ++sleepers_;
          cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
          --sleepers_;
        }
        if (stop_) return;
        seen = gen_.load(std::memory_order_acquire);
        j = job_;
      }
      if (id >= j.workers) continue;
      if (j.have_mask && !CPU_EQUAL(&j.mask, &mine)) {
        (void)sched_setaffinity(0, sizeof(j.mask), &j.mask);
        mine = j.mask;
      }
      j.call(j.ctx, id + 1);
      remaining_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::mutex run_mu_, mu_;
  std::condition_variable cv_;
  std::vector<std::thread> th_;
  Job job_;
  std::atomic<uint64_t> gen_{0};
  std::atomic<unsigned> remaining_{0};
  unsigned sleepers_ = 0;
  bool stop_ = false;
};
WorkTeam &team() {
  static WorkTeam t;
  return t;
}
template <typename F>
void par_run(unsigned nthr, F &&fn) {
  team().run(nthr, std::forward<F>(fn));
}

// The GPU join delivers every (key0,key1) bucket contiguous and internally ordered, plus the first/last insertion of every
// bucket and key0 group.  klib-khash's final slot layout depends only on the order in which DISTINCT keys are first
// inserted, plus one detail: a put of an already-present key still runs the load-factor check (khash.h:298-306), so if any
// put follows the last first-insertion the table may grow once more.  Both levels are replayed on distinct keys only.
// ids_only: leave the visit list as bucket ids (the device replay reads the records where the join left them)
// The outer table replayed AHEAD of the join's end (EarlyFn of dev_build_pairs): ids are positions in first-insertion order,
// i.e. id i stands for group gord[i] of the tables the join returns later.
// pinned blocks for the outer table's slot array (the device visit uploads it, pgx_visit.hip): a few, kept while the library is up
// Each block is a transparent-huge-page mapping registered with the HIP runtime (hipHostRegister) rather than hipHostMalloc
// memory: the table is probed at random by the host thread that replays it, and at 8 M slots (67 MB) every probe of 4 KiB
// pages is a TLB miss on top of the cache miss.
struct PinBlocks {
  struct B {
    void *p;
    size_t n;
    bool used;
    bool mapped;   // mmap + hipHostRegister (else hipHostMalloc)
  };
  std::mutex mu;
  std::vector<B> b;
};
PinBlocks &pin_blocks() {
  static PinBlocks z;
  return z;
}
void pin_block_release(void *p, size_t n, bool mapped) {
  if (mapped) {
    (void)hipHostUnregister(p);
    (void)munmap(p, n);
  } else {
    (void)hipHostFree(p);
  }
}
ShutdownHook g_pin_blocks_reset([] {
  PinBlocks &z = pin_blocks();
  std::lock_guard<std::mutex> lk(z.mu);
  for (auto &x : z.b)
    if (!x.used) pin_block_release(x.p, x.n, x.mapped);   // (a block still in use belongs to a table that is being torn down: leaked, not freed under it)
  z.b.clear();
});
void *pin_slot_alloc(size_t bytes) {
  PinBlocks &z = pin_blocks();
  std::lock_guard<std::mutex> lk(z.mu);
  for (auto &x : z.b)
    if (!x.used && x.n >= bytes && x.n <= 4 * bytes + (1u << 20)) {
      x.used = true;
      return x.p;
    }
  for (size_t i = 0; i < z.b.size(); ++i)   // the wrong size: let go of it
    if (!z.b[i].used) {
      pin_block_release(z.b[i].p, z.b[i].n, z.b[i].mapped);
      z.b.erase(z.b.begin() + i);
      break;
    }
  const size_t len = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p != MAP_FAILED) {
    (void)madvise(p, len, MADV_HUGEPAGE);
    memset(p, 0, len);   // (faulted in as huge pages before the runtime pins them)
    if (hipHostRegister(p, len, hipHostRegisterDefault) == hipSuccess) {
      z.b.push_back({p, len, true, true});
      return p;
    }
    (void)hipGetLastError();
    (void)munmap(p, len);
  }
  p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
    (void)hipGetLastError();
    throw std::bad_alloc();
  }
  z.b.push_back({p, bytes, true, false});
  return p;
}
void pin_slot_free(void *p, size_t) {
  PinBlocks &z = pin_blocks();
  std::lock_guard<std::mutex> lk(z.mu);
  for (auto &x : z.b)
    if (x.p == p) {
      x.used = false;
      return;
    }
  // (allocated before a pgx_shutdown: the registry is gone and with it the block's kind -- leaked rather than guessed)
}

struct PreOuter {
  DistinctSlotTable table;
  EarlyGroups eg;
  std::thread th;
  bool started = false;
  double ms = 0, t_start = 0, t_end = 0;
  void start(EarlyGroups &&g, size_t n_rec) {
    eg = std::move(g);
    const uint32_t min_n = getenv("PGX_EARLY_OUTER_MIN") ? (uint32_t)atol(getenv("PGX_EARLY_OUTER_MIN")) : 4096u;
    if (eg.n < min_n || eg.n >= (1u << 30)) return;   // (small sets: nothing to hide)
    started = true;
    table.reserve(eg.n, big_alloc, big_free, pin_slot_alloc, pin_slot_free);
    cpu_set_t saved, node;   // (the memory node the stage's other host threads will be pinned to: chosen from the caller's CPU)
    const bool pin = choose_node(saved, node);
    th = std::thread([this, n_rec, pin, node] {
      if (pin) (void)sched_setaffinity(0, sizeof(node), &node);
      const double t0 = now_ms();
      t_start = t0;
      const uint64_t *k = eg.keys.data();
      const size_t n = eg.n;
      // look-ahead of the put loop: the skip count of a key's home, then the slot it will take.  24 / 8 puts ahead while the table
      // lives in the caches (0.75 M keys at c3: 6.6 ms whatever the distances); a table far beyond them (c5s: 4.4 M keys, 8.4 M slots
      // = 67 MB + 34 MB of skip counts) needs the misses started ~100 ns x the puts per ns earlier: tools/khash_bench.cpp with the
      // reference's key shape (KB_REAL=1) on the GPU box's host: 82.5 ms at 24 / 8, 58.7 at 96 / 32, 56.1 at 200 / 64
      const size_t far = n >= ((size_t)3 << 19) ? 128 : 24, near = n >= ((size_t)3 << 19) ? 48 : 8;
      for (size_t i = 0; i < n; ++i) {
        if (i + far < n) table.prefetch_home(k[i + far]);
        if (i + near < n) table.prefetch(k[i + near]);
        table.put_new(k[i], (uint32_t)i);
      }
      if ((size_t)eg.last_first + 1 < n_rec) table.touch();  // a put after the last first-insertion (khash.h:298-306)
      t_end = now_ms();
      ms = t_end - t0;
    });
  }
  void join() {
    if (th.joinable()) th.join();
  }
  ~PreOuter() { join(); }
};

void build_visit(const PairTables &pt, uint32_t ovlp_upper, Visit &v, bool ids_only = false, PreOuter *pre = nullptr) {
  v.start.assign(1, 0), v.entries.clear(), v.bids.clear();
  v.on_device = false, v.n_groups = 0;
  v.n_buckets = v.n_entries = 0;
  const size_t ng = pt.gkey0.size();
  if (!ng) return;
  const bool trace = getenv("PGX_TRACE") != nullptr;
  if (trace && atoi(getenv("PGX_TRACE")) >= 3) {   // buckets per first-key group (log2 classes): groups, buckets
    uint64_t hg[33] = {0}, hb[33] = {0};
    for (size_t g = 0; g < ng; ++g) {
      const uint32_t n = pt.gbucket[g + 1] - pt.gbucket[g];
      int c = 0;
      while ((1u << c) < n) ++c;
      ++hg[c], hb[c] += n;
    }
    fprintf(stderr, "[pgx]   groups by buckets (<= 2^c: groups / buckets):");
    for (int c = 0; c < 33; ++c)
      if (hg[c]) fprintf(stderr, " 2^%d: %llu / %llu", c, (unsigned long long)hg[c], (unsigned long long)hb[c]);
    fprintf(stderr, "\n");
  }
  const double tv0 = now_ms();
  // The two levels are independent until the very end: the outer table only decides the ORDER in which the key0 groups
  // are visited, an inner table only the order of one group's buckets.  So one thread replays the outer table (a
  // sequential process with long probe chains: key0 = small hash << 8 | span is a poor input for khash's integer hash)
  // while the others replay the inner tables, group range by group range; then the groups' fragments are moved to their
  // final places in outer-slot order.
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  // inner-table workers: 48 for a lone process (2 x 64 cores), fewer per rank when several ranks share the host
  static const unsigned nin_cap = [] {
    const char *lw = getenv("LOCAL_WORLD_SIZE");
    const int world = lw ? std::max(1, atoi(lw)) : 1;
    return (unsigned)std::max(12, 48 / world);
  }();
  const unsigned nin = (unsigned)std::min<size_t>(std::min(nin_cap, hw), std::max<size_t>(1, ng / 2048));  // inner workers
  struct GroupOut {
    uint64_t eoff;      // offset of the group's entries in its worker's fragment
    uint32_t boff;      // offset of its bucket sizes
    uint32_t ne, nb;    // entries, buckets (0: nothing to visit)
    uint32_t worker;
  };
  HostArray<GroupOut> go(ng);   // (every element is assigned by its group's worker)
  struct Frag {  // sized up front from the group range (no growth, no copies)
    HostArray<uint32_t> own;    // bucket sizes
    uint32_t *sizes = nullptr;  // -> own, or (ids_only: the bucket ids) this worker's range of the shared array
    uint32_t base = 0;          // ids_only: offset of that range
    HostArray<Entry> entries;
    size_t ns = 0, ne = 0;
  };
  std::vector<Frag> frag(nin);
  if (ids_only) v.ids_all.alloc(pt.gbucket[ng]);   // every worker writes the ids of its group range into its own slice
  // the outer table: already being replayed by the early thread (ids = insertion positions), or replayed here
  bool pre_ok = pre && pre->started && pre->eg.n == ng;
  for (size_t i = 0; pre_ok && i < ng; i += 997) pre_ok = pre->eg.keys[i] == pt.gkey0[pt.gord[i]];
  if (pre && pre->started && !pre_ok) {
    pre->join();
    fprintf(stderr, "[pgx] note: the early outer-table keys do not match the join's group tables; replaying the outer table again\n");
  }
  DistinctSlotTable local_outer;
  DistinctSlotTable &outer = pre_ok ? pre->table : local_outer;
  PGX_REQUIRE(ng < ((size_t)1 << 30), PGX_EARG, "too many first-key groups for one overlap chunk");   // (DistinctSlotTable: 30-bit ids)
  if (!pre_ok) outer.reserve(ng, big_alloc, big_free);
  auto gid = [&](uint32_t s0) { return pre_ok ? pt.gord[outer.id_at(s0)] : outer.id_at(s0); };
  auto outer_work = [&] {
    if (pre_ok) {
      pre->join();
      return;
    }
    const HostArray<uint32_t> &gord = pt.gord;  // groups by first insertion (sorted on the GPU)
    // three dependent misses per put on a table that has outgrown the caches -- the key (gkey0 is indexed through the
    // permutation), the home slot's skip count, the slot the probe sequence resumes at -- each started a stage earlier
    for (size_t i = 0; i < ng; ++i) {
      if (i + 48 < ng) __builtin_prefetch(&pt.gkey0[gord[i + 48]]);
      if (i + 24 < ng) outer.prefetch_home(pt.gkey0[gord[i + 24]]);
      if (i + 8 < ng) outer.prefetch(pt.gkey0[gord[i + 8]]);
      outer.put_new(pt.gkey0[gord[i]], gord[i]);
    }
    if ((size_t)pt.gfirst[gord.back()] + 1 < pt.n_rec) outer.touch();  // a put after the last first-insertion (khash.h:298-306)
  };
  auto inner_work = [&](unsigned ti) {
    // group range with ~1/nin of the records
    auto split = [&](unsigned t) {
      if (t == 0) return (size_t)0;
      if (t >= nin) return ng;
      const uint32_t want = (uint32_t)((uint64_t)pt.n_rec * t / nin);
      return (size_t)(std::lower_bound(pt.gstart.begin(), pt.gstart.begin() + ng, want) - pt.gstart.begin());
    };
    const size_t g_lo = split(ti), g_hi = split(ti + 1);
    Frag &f = frag[ti];
    if (g_lo >= g_hi) return;
    if (!ids_only) f.entries.alloc(pt.gstart[g_hi] - pt.gstart[g_lo]);   // (gstart / gbucket carry an end sentinel)
    if (ids_only) f.sizes = v.ids_all.data() + pt.gbucket[g_lo], f.base = pt.gbucket[g_lo];
    else f.own.alloc(pt.gbucket[g_hi] - pt.gbucket[g_lo]), f.sizes = f.own.data();
    ScratchTable in;
    bool ab;
    for (size_t g = g_lo; g < g_hi; ++g) {
      GroupOut &o = go[g];
      o = GroupOut{f.ne, (uint32_t)f.ns, 0, 0, ti};
      if (pt.gstart[g + 1] - pt.gstart[g] <= 2) continue;  // no bucket of this key0 can hold more than 2 records
      const uint32_t b0 = pt.gbucket[g], b1 = pt.gbucket[g + 1];
      const uint32_t *bord = pt.bord.data() + b0;  // this group's buckets by first insertion (sorted on the GPU)
      const uint64_t *k1 = pt.bkey1_ord.data() + b0;   // their second keys and sizes, in that order (PAIRS_ORD_TABLES)
      const uint32_t *bsz = pt.bn_ord.data() + b0;
      in.reset();
      for (uint32_t i = 0; i < b1 - b0; ++i) in.put(k1[i], i, &ab);   // (id = position in the group's slice)
      if (pt.gtrail[g]) in.put(k1[0], 0, &ab);  // trailing repeat put
      for (uint32_t s1 = 0; s1 < in.nb; ++s1) {
        if (!in.used[s1]) continue;
        const uint32_t bi = in.ids[s1];
        const uint32_t bn = bsz[bi], b = bord[bi];
        if (bn <= 2 || bn > ovlp_upper) continue;  // shmr_overlap.c:216
        if (ids_only) {
          f.sizes[f.ns++] = b, f.ne += bn;
        } else {
          for (uint32_t r = pt.bstart[b]; r < pt.bstart[b + 1]; ++r) {
            const uint64_t y = pt.y0[r];
            f.entries[f.ne++] = Entry{(uint32_t)(y >> 32), pos_of(y) + 1, y, pt.dir[r]};
          }
          f.sizes[f.ns++] = bn;
        }
        o.ne += bn, ++o.nb;
      }
    }
  };
  double t_outer = 0, t_inner = 0, t_inner_only = 0;
  std::atomic<int> inner_left{(int)nin};
  if (nin == 1) {
    outer_work();
    t_outer = now_ms() - tv0;
    inner_work(0);
    t_inner = now_ms() - tv0 - t_outer;
  } else {
    par_run(nin + 1, [&](unsigned ti) {
      if (ti == 0) outer_work(), t_outer = now_ms() - tv0;
      else {
        inner_work(ti - 1);
        if (inner_left.fetch_sub(1) == 1) t_inner_only = now_ms() - tv0;
      }
    });
    t_inner = now_ms() - tv0;
  }
  const double tv2 = now_ms();
  if (ids_only && nin > 1 && outer.nb >= (1u << 16)) {
    // ids-only: one random pass over the groups' results (slot range per worker), then a sequential one that writes the copy
    // descriptors the GPU assembles the visit list from (dev_place_bids)
    struct P2 {
      std::vector<uint32_t> src, cnt;
      uint64_t ne = 0, nb = 0;
    };
    std::vector<P2> piece(nin);
    par_run(nin, [&](unsigned ti) {
      P2 &pc = piece[ti];
      const uint32_t lo = (uint32_t)((uint64_t)outer.nb * ti / nin), hi = (uint32_t)((uint64_t)outer.nb * (ti + 1) / nin);
      pc.src.reserve((hi - lo) / 2 + 16), pc.cnt.reserve((hi - lo) / 2 + 16);
      for (uint32_t s0 = lo; s0 < hi; ++s0) {
        // (two dependent random reads per used slot -- the group of the slot's key, then its result -- each started ahead)
        if (pre_ok && s0 + 32 < hi && outer.is_used(s0 + 32)) __builtin_prefetch(&pt.gord[outer.id_at(s0 + 32)]);
        if (s0 + 12 < hi && outer.is_used(s0 + 12)) __builtin_prefetch(&go[gid(s0 + 12)]);
        if (outer.is_used(s0)) {
          const GroupOut &o = go[gid(s0)];
          if (o.nb) pc.src.push_back(frag[o.worker].base + o.boff), pc.cnt.push_back(o.nb), pc.ne += o.ne, pc.nb += o.nb;
        }
      }
    });
    const double tv3 = now_ms();
    std::vector<size_t> first(nin + 1, 0);
    std::vector<uint64_t> b0(nin + 1, 0);
    uint64_t ne = 0;
    for (unsigned t = 0; t < nin; ++t)
      first[t + 1] = first[t] + piece[t].src.size(), b0[t + 1] = b0[t] + piece[t].nb, ne += piece[t].ne;
    const size_t no = first[nin];
    v.n_buckets = b0[nin], v.n_entries = ne, v.on_device = true, v.n_groups = no;
    v.psrc.alloc(no), v.pcnt.alloc(no), v.pdst.alloc(no);
    const double tv4 = now_ms();
    par_run(nin, [&](unsigned ti) {
      uint64_t b = b0[ti];
      size_t at = first[ti];
      const P2 &pc = piece[ti];
      for (size_t k = 0; k < pc.src.size(); ++k, ++at) v.psrc[at] = pc.src[k], v.pcnt[at] = pc.cnt[k], v.pdst[at] = b, b += pc.cnt[k];
    });
    if (trace) {
      fprintf(stderr, "[pgx]   visit: outer table %.2f ms%s alongside %u inner-table workers (done at %.2f ms), slot scan %.2f ms\n",
              pre_ok ? pre->ms : t_outer, pre_ok ? " (started during the join)" : "", nin, t_inner, now_ms() - tv2);
      if (pre_ok) fprintf(stderr, "[pgx]   visit: the early outer table ran from %.2f ms before to %.2f ms after the join's end; inner workers alone %.2f ms; "
                          "slot scan: pieces %.2f ms, descriptor arrays %.2f ms, descriptors %.2f ms\n",
                          tv0 - pre->t_start, pre->t_end - tv0, t_inner_only, tv3 - tv2, tv4 - tv3, now_ms() - tv4);
    }
    return;
  }
  // final places: groups in ascending outer slot order
  std::vector<uint32_t> order;
  std::vector<uint64_t> eat, bat;
  if (nin == 1 || outer.nb < (1u << 16)) {
    order.reserve(ng);
    for (uint32_t s0 = 0; s0 < outer.nb; ++s0)
      if (outer.is_used(s0) && go[gid(s0)].nb) order.push_back(gid(s0));
    eat.assign(order.size() + 1, 0), bat.assign(order.size() + 1, 0);
    for (size_t i = 0; i < order.size(); ++i) eat[i + 1] = eat[i] + go[order[i]].ne, bat[i + 1] = bat[i] + go[order[i]].nb;
  } else {
    // the slot scan touches one GroupOut per used slot at random: every worker takes a slot range, the pieces are joined in
    // range order, and the running totals are carried over the pieces
    struct Piece {
      std::vector<uint32_t> ids;
      uint64_t ne = 0, nb = 0;
    };
    std::vector<Piece> piece(nin);
    par_run(nin, [&](unsigned ti) {
      Piece &pc = piece[ti];
      const uint32_t lo = (uint32_t)((uint64_t)outer.nb * ti / nin), hi = (uint32_t)((uint64_t)outer.nb * (ti + 1) / nin);
      for (uint32_t s0 = lo; s0 < hi; ++s0)
        if (outer.is_used(s0)) {
          const uint32_t g = gid(s0);
          const GroupOut &o = go[g];
          if (o.nb) pc.ids.push_back(g), pc.ne += o.ne, pc.nb += o.nb;
        }
    });
    std::vector<size_t> first(nin + 1, 0);
    std::vector<uint64_t> e0(nin + 1, 0), b0(nin + 1, 0);
    for (unsigned t = 0; t < nin; ++t)
      first[t + 1] = first[t] + piece[t].ids.size(), e0[t + 1] = e0[t] + piece[t].ne, b0[t + 1] = b0[t] + piece[t].nb;
    order.resize(first[nin]);
    eat.assign(first[nin] + 1, 0), bat.assign(first[nin] + 1, 0);
    par_run(nin, [&](unsigned ti) {
      uint64_t e = e0[ti], b = b0[ti];
      size_t at = first[ti];
      for (uint32_t id : piece[ti].ids) {
        order[at] = id, eat[at] = e, bat[at] = b;
        e += go[id].ne, b += go[id].nb, ++at;
      }
    });
    eat[first[nin]] = e0[nin], bat[first[nin]] = b0[nin];
  }
  const size_t no = order.size();
  const uint64_t ne = eat[no], nbk = bat[no];
  v.n_buckets = nbk, v.n_entries = ne;
  if (ids_only) {
    // the bucket ids are assembled in visit order on the GPU (dev_place_bids): the host only says which slice goes where
    v.on_device = true, v.n_groups = no;
    v.psrc.alloc(no), v.pcnt.alloc(no), v.pdst.alloc(no);
    auto desc = [&](unsigned ti, unsigned nt) {
      for (size_t i = no * ti / nt, ie = no * (ti + 1) / nt; i < ie; ++i) {
        const GroupOut &o = go[order[i]];
        v.psrc[i] = frag[o.worker].base + o.boff, v.pcnt[i] = o.nb, v.pdst[i] = bat[i];
      }
    };
    if (nin == 1) desc(0, 1);
    else par_run(nin, [&](unsigned ti) { desc(ti, nin); });
  } else {
    v.entries.alloc(ne), v.start.resize(nbk + 1);
    auto place = [&](unsigned ti, unsigned nt) {
      for (size_t i = no * ti / nt, ie = no * (ti + 1) / nt; i < ie; ++i) {
        const GroupOut &o = go[order[i]];
        const Frag &f = frag[o.worker];
        memcpy(v.entries.data() + eat[i], f.entries.data() + o.eoff, (size_t)o.ne * sizeof(Entry));
        uint64_t at = eat[i];
        for (uint32_t j = 0; j < o.nb; ++j) v.start[bat[i] + j] = at, at += f.sizes[o.boff + j];
      }
    };
    if (nin == 1) place(0, 1);
    else par_run(nin, [&](unsigned ti) { place(ti, nin); });
  }
  if (!ids_only) v.start[nbk] = ne;
  if (trace)
    fprintf(stderr, "[pgx]   visit: outer table %.2f ms%s alongside %u inner-table workers (done at %.2f ms), placement %.2f ms\n",
            pre_ok ? pre->ms : t_outer, pre_ok ? " (started during the join; waited for" : "", nin, t_inner, now_ms() - tv2);
}

// ---------------------------------------------------------------------------------------------------------
// greedy replay (shimmer_to_overlap, shmr_overlap.c:52-180) over the visit list with an alignment memo
// ---------------------------------------------------------------------------------------------------------
enum { T_OVERLAP = 0, T_CONTAINS = 1, T_CONTAINED = 2 };
constexpr int END_FUZZ = 48;              // READ_END_FUZZINESS, shmr_overlap.c:36

struct Verdict {
  bool accepted;
  uint8_t type;
};

// ---------------------------------------------------------------------------------------------------------
// Incremental greedy replay.
//
// The reference walks the buckets once, in order, sharing one seen-pair table (shmr_overlap.c:194-228).  Here every
// bucket's evaluation is a pure function of (a) the seen-pair entries OWNED BY EARLIER BUCKETS for the pairs it examines
// and (b) the alignment results it looks up.  Each pair remembers which bucket inserted it ("owner") and which buckets
// examined it ("readers").  A round scans the buckets in order and (re)evaluates only the dirty ones; when a bucket's
// insertions change, the later readers of those pairs become dirty, and a later owner displaced by an earlier insertion
// becomes dirty too.  Unknown alignments are requested and GUESSED (accepted; type predicted from the geometry); after
// the GPU batch a wrong guess makes its bucket dirty, a right guess only has its record patched.  At the fixed point every
// bucket was last evaluated against final inputs, which is exactly the sequential process.
// ---------------------------------------------------------------------------------------------------------
struct Replay {
  static constexpr uint32_t NONE = 0xFFFFFFFFu;
  const Visit &v;
  const std::vector<uint32_t> &rlen;
  uint32_t bestn;
  bool predict = true;  // PGX_PREDICT=0: guess "plain overlap" always

  AKeyMap memo;                     // alignment key -> global request number (its result is pending while >= req_base)
  std::vector<pgx_match> results;   // indexed by global request number
  std::vector<pgx_align_key> requests;  // this sweep's requests: global number = req_base + index
  uint32_t req_base = 0;

  PairMap pair_id;                  // read pair -> dense id
  struct PState {
    uint32_t owner;                 // owning bucket or NONE
    uint32_t rhead;                 // head of the pair's reader list in rlog
    uint32_t type;
    uint32_t last_reader;           // bucket of the newest reader-list node (avoids touching rlog on the hot path)
  };
  std::vector<PState> ps;
  struct RNode {
    uint32_t next, bucket;
  };
  std::vector<RNode> rlog;

  struct BState {
    uint32_t rec0 = 0, nrec = 0;    // range in recs
    uint32_t own0 = 0, nown = 0;    // range in owned (pair id, type)
    uint32_t lookups = 0, skips = 0;
  };
  std::vector<BState> bs;
  std::vector<pgx_ovlp> recs;       // arena; re-evaluated buckets append a fresh range
  struct Own {
    uint32_t pid;
    uint8_t type;
  };
  std::vector<Own> owned;           // arena
  std::vector<uint8_t> dirty;
  struct Guess {
    uint32_t bucket, req, rec, rlen0, rlen1, q_off;
    uint8_t type;
  };
  std::vector<Guess> guesses;
  std::vector<uint8_t> contained;
  std::vector<Own> old_own;
  uint64_t n_eval = 0;

  Replay(const Visit &vv, const std::vector<uint32_t> &rl, uint32_t bn) : v(vv), rlen(rl), bestn(bn) {
    pair_id.init(std::max<size_t>(1 << 16, v.entries.size()));
    memo.init(std::max<size_t>(1 << 16, v.entries.size()));
    const size_t nb = v.start.size() - 1;
    bs.assign(nb, BState());
    dirty.assign(nb, 1);
  }

  static inline int64_t iabs(int64_t x) { return x < 0 ? -x : x; }

  // acceptance test and classification of shimmer_to_overlap (shmr_overlap.c:134-160)
  static Verdict classify(const pgx_match &m, uint32_t rlen0, uint32_t rlen1, uint32_t q_off) {
    const uint32_t slen0 = rlen0 - q_off, slen1 = rlen1;
    Verdict r{false, T_OVERLAP};
    if (m.q_bgn < END_FUZZ && m.t_bgn < END_FUZZ &&
        (iabs((int64_t)slen0 - m.q_end) < END_FUZZ || iabs((int64_t)slen1 - m.t_end) < END_FUZZ) && m.q_end > 500 &&
        m.t_end > 500) {
      r.accepted = true;
      if (iabs((int64_t)rlen0 - ((int64_t)m.q_end - m.q_bgn)) < END_FUZZ * 2 ||
          iabs((int64_t)rlen1 - ((int64_t)m.t_end - m.t_bgn)) < END_FUZZ * 2)
        r.type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
    }
    return r;
  }

  uint32_t pid_of(uint64_t pair) {
    bool fresh;
    uint32_t *p = pair_id.slot(pair, &fresh);
    if (fresh) {
      *p = (uint32_t)ps.size();
      ps.push_back(PState{NONE, NONE, 0, NONE});
    }
    return *p;
  }
  void mark_readers_after(uint32_t pid, uint32_t b) {
    for (uint32_t n = ps[pid].rhead; n != NONE; n = rlog[n].next)
      if (rlog[n].bucket > b) dirty[rlog[n].bucket] = 1;
  }

  // shimmer_to_overlap (shmr_overlap.c:52-180) for bucket b against the entries owned by earlier buckets
  void eval(uint32_t b) {
    ++n_eval;
    BState &st = bs[b];
    // withdraw what the previous evaluation of this bucket inserted
    old_own.assign(owned.begin() + st.own0, owned.begin() + st.own0 + st.nown);
    for (const Own &o : old_own)
      if (ps[o.pid].owner == b) ps[o.pid].owner = NONE;
    st.rec0 = (uint32_t)recs.size(), st.nrec = 0, st.own0 = (uint32_t)owned.size(), st.nown = 0;
    st.lookups = st.skips = 0;
    const Entry *e = v.entries.data() + v.start[b];
    const size_t n = v.start[b + 1] - v.start[b];
    contained.assign(n, 0);
    for (size_t hi = n - 1; hi > 0; --hi) {
      const size_t ai = hi - 1;
      if (contained[ai]) continue;
      const uint32_t rid0 = e[ai].rid, pos0 = e[ai].pos1, rlen0 = rlen[rid0];
      size_t got = 0;
      for (size_t pi = ai + 1; pi < n && got < bestn; ++pi) {
        if (contained[pi]) continue;
        const uint32_t rid1 = e[pi].rid;
        if (rid0 == rid1) continue;
        const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
        const uint32_t pid = pid_of(pair);
        PState &pst = ps[pid];
        if (pst.last_reader != b) {  // register as a reader (once per evaluation run)
          rlog.push_back(RNode{pst.rhead, b});
          pst.rhead = (uint32_t)rlog.size() - 1;
          pst.last_reader = b;
        }
        if (pst.owner != NONE && pst.owner <= b) {  // present in the table as this bucket sees it
          if (pst.type == T_OVERLAP) ++got;
          ++st.skips;
          continue;
        }
        const uint32_t pos1 = e[pi].pos1, rlen1 = rlen[rid1];
        const uint32_t q_off = pos0 - pos1;
        const AKey key{(uint64_t)rid0 << 32 | rid1, (uint64_t)q_off << 2 | (uint64_t)e[ai].dir << 1 | e[pi].dir};
        bool fresh;
        uint32_t *mv = memo.slot(key, &fresh);
        ++st.lookups;
        Verdict vd;
        const pgx_match *mm = nullptr;
        if (fresh) {
          *mv = req_base + (uint32_t)requests.size();
          requests.push_back(pgx_align_key{rid0, rid1, q_off, e[ai].dir, e[pi].dir, {0, 0}});
        }
        if (*mv >= req_base) {
          // guess: accepted; the type follows from the geometry the shimmer pair implies (read1 starts q_off bases into
          // read0): if read1 fits inside the rest of read0, or read0 starts (almost) where read1 starts, a containment
          vd.accepted = true;
          vd.type = T_OVERLAP;
          if (predict && (rlen1 <= rlen0 - q_off || q_off < (uint32_t)(END_FUZZ * 2 - 8)))
            vd.type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
          guesses.push_back(Guess{b, *mv, (uint32_t)recs.size(), rlen0, rlen1, q_off, vd.type});
        } else {
          mm = &results[*mv];
          vd = classify(*mm, rlen0, rlen1, q_off);
        }
        if (vd.accepted) {
          if (vd.type == T_OVERLAP) ++got;
          else if (vd.type == T_CONTAINS) contained[pi] = 1;
          else contained[ai] = 1;
          PState &pw = ps[pid];  // (ps may have been reallocated by pid_of? no: no insertion since `pst`)
          if (pw.owner != NONE && pw.owner > b) dirty[pw.owner] = 1;  // a later bucket had inserted it
          pw.owner = b, pw.type = vd.type;
          owned.push_back(Own{pid, vd.type});
          ++st.nown;
          pgx_ovlp o;
          memset(&o, 0, sizeof(o));
          o.y0 = e[ai].y0, o.y1 = e[pi].y0, o.rl0 = rlen0, o.rl1 = rlen1;
          o.strand0 = e[ai].dir, o.strand1 = e[pi].dir, o.ovlp_type = vd.type;
          if (mm) o.match = *mm;
          recs.push_back(o);
          ++st.nrec;
        }
        if (contained[ai]) break;
      }
    }
    // what changed for later buckets?  (skipped while everything behind is dirty anyway: first sweep)
    if (!first_sweep) {
      for (const Own &o : old_own)
        if (ps[o.pid].owner != b || (ps[o.pid].type == T_OVERLAP) != (o.type == T_OVERLAP)) mark_readers_after(o.pid, b);
      for (uint32_t i = 0; i < st.nown; ++i) {
        const Own &o = owned[st.own0 + i];
        bool same = false;
        for (const Own &q : old_own)
          if (q.pid == o.pid && (q.type == T_OVERLAP) == (o.type == T_OVERLAP)) {  // readers only observe presence
            same = true;                                                           // and "is a plain overlap"
            break;
          }
        if (!same) mark_readers_after(o.pid, b);
      }
    }
  }
  bool first_sweep = true;

  // one round: evaluate the dirty buckets in order; returns the number of alignments requested
  size_t sweep() {
    req_base = (uint32_t)results.size();
    requests.clear();
    guesses.clear();
    const size_t nb = bs.size();
    for (size_t b = 0; b < nb; ++b)
      if (dirty[b]) {
        dirty[b] = 0;
        eval((uint32_t)b);
      }
    first_sweep = false;
    return requests.size();
  }

  // room for this sweep's results (the GPU batch writes them in place)
  pgx_match *result_slots() {
    results.resize((size_t)req_base + requests.size());
    return results.data() + req_base;
  }
  // after the GPU batch: right guesses get their record patched, wrong ones make their bucket dirty
  bool settle() {
    bool any = false;
    for (const Guess &g : guesses) {
      const pgx_match &m = results[g.req];
      const Verdict vd = classify(m, g.rlen0, g.rlen1, g.q_off);
      if (!vd.accepted || vd.type != g.type) dirty[g.bucket] = 1, any = true;
      else recs[g.rec].match = m;
    }
    return any;
  }

  void collect(OvOut &out, uint64_t &lookups, uint64_t &skips) const {
    size_t total = 0;
    for (const BState &b : bs) total += b.nrec;
    out.alloc(total);
    lookups = skips = 0;
    pgx_ovlp *w = out.a;
    for (const BState &b : bs) {
      if (b.nrec) memcpy(w, recs.data() + b.rec0, (size_t)b.nrec * sizeof(pgx_ovlp)), w += b.nrec;
      lookups += b.lookups, skips += b.skips;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// Parallel form of the incremental replay (same fixed point, many host threads).
//
// Buckets are evaluated concurrently, roughly in visit order, against ONE shared pair table.  The owner/reader
// protocol that makes the sequential replay incremental also makes a premature evaluation harmless:
//   reader  (bucket b examines pair P):  push b on P's reader list, THEN load P's owner;
//   writer  (bucket a inserts/withdraws P): store P's owner, THEN scan P's reader list and dirty the readers > a;
// both with sequentially consistent atomics, so either the writer sees the reader or the reader sees the new owner
// (Dekker).  An insertion never overwrites an earlier owner and dirties a displaced later owner.  A bucket is
// evaluated at most once per round (rounds are separated by a barrier), so per-bucket state needs no locking.  Rounds
// repeat until no bucket is dirty; the unique fixed point is the sequential process, whatever the interleaving.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct BlockArena {  // append-only, never moves what it handed out (other threads may still read old ranges)
  size_t block = 1 << 12;  // elements per block; set_block() scales it with the job (big jobs: huge-page sized blocks)
  std::vector<HostArray<T>> blocks;
  size_t used = 0, cap = 0;
  void set_block(size_t expected_elements) {
    block = std::min<size_t>(std::max<size_t>(expected_elements / 4, 1 << 12), ((size_t)32 << 20) / sizeof(T));
  }
  T *alloc(size_t n) {
    if (used + n > cap) {
      cap = std::max(block, n);
      blocks.emplace_back(cap);
      used = 0;
    }
    T *p = blocks.back().data() + used;
    used += n;
    return p;
  }
};

struct ParReplay {
  // every field of the shared pair table encodes "nothing" as 0, so the table is plain zero-filled pages
  static constexpr uint64_t NOOWN = 0;
  static constexpr uint64_t EMPTY = 0;
  static constexpr uint32_t NIL = 0;
  static constexpr uint32_t NIN = 11;
  static constexpr uint32_t NO_CHUNK = 0xFFFFFFFFu, ALLOCATING = 0xFFFFFFFEu;
  struct Overflow {};

  const Visit &v;
  const std::vector<uint32_t> &rlen;
  uint32_t bestn;
  bool predict = true;
  bool trace = false;
  unsigned nthr;
  size_t block = 64;  // buckets a worker takes at a time: neighbours in visit order share a key0 group, hence reads and
                      // pairs, so they are best evaluated in order by one thread (measured: 16 -> 12.6 k conflicts in the
                      // first round at 4.5 Gbases, 64 -> 4.6 k, 1024 -> 22 k because the in-flight window grows)

  struct alignas(64) PSlot {       // one cache line per read pair
    std::atomic<uint64_t> key;     // pair + 1, or EMPTY
    std::atomic<uint64_t> own;     // (owner bucket << 8 | type) + 1, or NOOWN
    std::atomic<uint32_t> rhead;   // overflow reader list: index into rlog, or NIL
    std::atomic<uint32_t> in[NIN]; // the first readers, bucket + 1, filled front to back (0: free)
  };
  static_assert(sizeof(PSlot) == 64, "pair slot must be one cache line");
  PSlot *ptab = nullptr;           // mmap'd: zero pages, transparent huge pages where the kernel grants them
  size_t pcap = 0;
  struct RNode {
    uint32_t next, bucket;
  };
  HostArray<RNode> rlog;
  // (the shared counters live on cache lines of their own, below: a fetch_add next to the read-mostly pointers would
  //  evict those from every other core each time)
  uint32_t rcap = 0;

  // alignment memo: insert-only, lock-free.  A slot is claimed by a CAS on `a` (rid0 << 32 | rid1, never 0 because the two
  // reads differ); `bv` = (q_off << 2 | dir0 << 1 | dir1) << 32 | (request number + 1) follows with a release store, and
  // a thread that meets a claimed slot whose `bv` is still 0 waits the few nanoseconds until it appears.
  struct MSlot {
    std::atomic<uint64_t> a, bv;
  };
  MSlot *mtab = nullptr;  // mmap'd zero pages
  size_t mcap = 0;
  // request r's result lives in results[r]; it is pending while r >= settled (settled only moves between sweeps)
  HostArray<pgx_match> results;
  size_t settled = 0;
  HostArray<pgx_align_key> requests;
  uint32_t reqcap = 0;

  struct Own {
    uint32_t pid;
    uint8_t type;
  };
  struct Guess {
    uint32_t bucket, epoch, req, rlen0, rlen1, q_off;
    pgx_ovlp *rec;
    uint8_t type;
  };
  struct BState {
    pgx_ovlp *recs = nullptr;
    Own *own = nullptr;
    uint32_t nrec = 0, nown = 0, lookups = 0, skips = 0, epoch = 0;
  };
  std::vector<BState> bs;
  std::unique_ptr<std::atomic<uint8_t>[]> dirty;
  struct alignas(128) TL {  // per-thread state on its own cache lines (no false sharing between neighbours)
    BlockArena<pgx_ovlp> recs;
    BlockArena<Own> owned;
    std::vector<Guess> guesses;
    std::vector<uint8_t> contained;
    std::vector<pgx_ovlp> tmp_recs;
    std::vector<Own> tmp_own;
    uint64_t n_eval = 0;
    uint32_t rnext = 0, rend = 0;      // private chunk of reader-node indices
    uint32_t qnext = 0, qend = 0;      // private chunk of request slots
    std::atomic<uint32_t> cur_chunk{NO_CHUNK};  // first slot of the chunk being filled (what the submitter may not ship yet)
  };
  static constexpr uint32_t RCHUNK = 4096, QCHUNK = 32;
  static constexpr size_t PREFETCH = 3;
  struct TLArray {  // (TL holds an atomic, so it cannot live in a std::vector)
    std::unique_ptr<TL[]> p;
    size_t n = 0;
    void resize(size_t count) { p.reset(new TL[count]), n = count; }
    TL &operator[](size_t i) { return p[i]; }
    const TL &operator[](size_t i) const { return p[i]; }
    TL *begin() { return p.get(); }
    TL *end() { return p.get() + n; }
    const TL *begin() const { return p.get(); }
    const TL *end() const { return p.get() + n; }
    size_t size() const { return n; }
  } tl;
  alignas(128) std::atomic<size_t> cursor{0};
  alignas(128) std::atomic<uint32_t> nreq{0};
  alignas(128) std::atomic<uint32_t> rcount{1};  // node 0 is NIL
  alignas(128) std::atomic<bool> overflow{false};
  alignas(128) char tail_pad = 0;

  static uint64_t enc(uint32_t owner, uint8_t type) { return ((uint64_t)owner << 8 | type) + 1; }
  static uint32_t owner_of(uint64_t o) { return (uint32_t)((o - 1) >> 8); }
  static uint8_t type_of(uint64_t o) { return (uint8_t)((o - 1) & 0xFF); }

  ParReplay(const Visit &vv, const std::vector<uint32_t> &rl, uint32_t bn, unsigned threads)
      : v(vv), rlen(rl), bestn(bn), nthr(threads) {
    const size_t ne = std::max<size_t>(v.entries.size(), 1024);
    pcap = 1024;
    while (pcap < ne - ne / 4) pcap <<= 1;  // distinct pairs ~ 0.25-0.3 x entries; > 70 % load -> Overflow -> sequential replay
    ptab = (PSlot *)big_alloc_zero(pcap * sizeof(PSlot));  // huge pages: random probes over a GB-sized table
    rcap = (uint32_t)std::min<size_t>(ne * 10 + (size_t)nthr * RCHUNK, 0xFFFFFFF0u);
    rlog.alloc(rcap);
    reqcap = (uint32_t)std::min<size_t>(ne * 2 + (size_t)nthr * QCHUNK * 8 + 1024, 0x7FFFFFF0u);
    requests.alloc(reqcap);
    results.alloc(reqcap);  // untouched pages cost nothing
    mcap = 1024;
    while (mcap < ne) mcap <<= 1;  // distinct alignments ~ 0.3 x entries
    mtab = (MSlot *)big_alloc_zero(mcap * sizeof(MSlot));
    const size_t nb = v.start.size() - 1;
    bs.assign(nb, BState());
    dirty.reset(new std::atomic<uint8_t>[nb ? nb : 1]);
    for (size_t i = 0; i < nb; ++i) dirty[i].store(1, std::memory_order_relaxed);
    tl.resize(nthr);
    for (TL &t : tl) {  // ~0.3 alignments (records, insertions, first-sweep guesses) per entry
      t.guesses.reserve(ne / nthr / 3 + 1024);
      t.recs.set_block(ne / nthr / 3);
      t.owned.set_block(ne / nthr / 3);
    }
  }
  ~ParReplay() {
    big_free_zero((void *)ptab, pcap * sizeof(PSlot));  // (cleared here, i.e. on the housekeeping thread)
    big_free_zero((void *)mtab, mcap * sizeof(MSlot));
  }
  ParReplay(const ParReplay &) = delete;
  ParReplay &operator=(const ParReplay &) = delete;

  // A seq_cst load is a plain load on x86, a seq_cst store a locked exchange that drains the store buffer and ends all
  // memory-level parallelism: most marks hit a flag that is already set, so look first.  (If the flag reads 1 the
  // bucket's next evaluation starts after this point in the seq_cst order and therefore sees the caller's update.)
  void mark_dirty(uint32_t b) {
    if (dirty[b].load(std::memory_order_seq_cst) == 0) dirty[b].store(1, std::memory_order_seq_cst);
  }

  uint32_t pid_of(uint64_t pair) {
    size_t i = mix(pair) & (pcap - 1);
    const uint64_t want = pair + 1;
    unsigned probes = 0;
    for (;;) {
      uint64_t k = ptab[i].key.load(std::memory_order_acquire);
      if (k == want) return (uint32_t)i;
      if (k == EMPTY) {
        if (ptab[i].key.compare_exchange_strong(k, want, std::memory_order_acq_rel)) return (uint32_t)i;
        if (k == want) return (uint32_t)i;
      }
      i = (i + 1) & (pcap - 1);
      if (++probes > 512) {  // the table is far fuller than sized for: give up (sequential replay takes over)
        overflow.store(true);
        return (uint32_t)i;
      }
    }
  }
  // Register bucket b as a reader of the pair BEFORE it loads the owner.  The first NIN readers live in the slot's own
  // cache line; a bucket that is already listed (an earlier evaluation) is not added again.
  void add_reader(PSlot &ps, uint32_t b, TL &t) {
    // ONE locked operation: a compare-exchange on the first free inline entry both publishes the reader and orders the
    // publication before the owner load that follows (a listing by an earlier evaluation needs nothing: the writer's scan
    // finds it, and this run's owner load follows the seq_cst exchange that cleared dirty[b])
    for (uint32_t i = 0; i < NIN; ++i) {
      uint32_t cur = ps.in[i].load(std::memory_order_relaxed);
      if (cur == b + 1) return;
      if (cur == 0) {
        if (ps.in[i].compare_exchange_strong(cur, b + 1, std::memory_order_seq_cst)) return;
        if (cur == b + 1) return;  // (cannot happen: a bucket is evaluated by one thread at a time)
      }
    }
    if (t.rnext == t.rend) {  // a shared counter per node would serialise the threads on one cache line
      t.rnext = rcount.fetch_add(RCHUNK, std::memory_order_relaxed);
      t.rend = t.rnext + RCHUNK;
    }
    const uint32_t n = t.rnext++;
    if (n >= rcap) {
      overflow.store(true);
      return;
    }
    rlog[n].bucket = b;
    uint32_t h = ps.rhead.load(std::memory_order_seq_cst);
    do {
      rlog[n].next = h;
    } while (!ps.rhead.compare_exchange_weak(h, n, std::memory_order_seq_cst));
  }
  void mark_readers_after(PSlot &ps, uint32_t b) {
    for (uint32_t i = 0; i < NIN; ++i) {
      const uint32_t x = ps.in[i].load(std::memory_order_seq_cst);
      if (x == 0) break;  // entries fill front to back; a reader that lists itself later loads the owner after this point
      if (x > b + 1) mark_dirty(x - 1);
    }
    for (uint32_t n = ps.rhead.load(std::memory_order_seq_cst); n != NIL; n = rlog[n].next)
      if (rlog[n].bucket > b) mark_dirty(rlog[n].bucket);
  }

  // The pair table is far larger than the caches and every examination is a random probe into it: the worker starts
  // the misses of the NEXT bucket's likely probes (the first PREFETCH partners of every row) before evaluating this one.
  void prefetch_bucket(uint32_t b) const {
    const Entry *e = v.entries.data() + v.start[b];
    const size_t n = v.start[b + 1] - v.start[b];
    for (size_t ai = 0; ai + 1 < n; ++ai) {
      const uint32_t rid0 = e[ai].rid;
      for (size_t pi = ai + 1, pe = std::min(n, ai + 1 + PREFETCH); pi < pe; ++pi) {
        const uint32_t rid1 = e[pi].rid;
        const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
        __builtin_prefetch(&ptab[mix(pair) & (pcap - 1)], 1);
        if (memo_prefetch && pi == ai + 1 && rid0 != rid1) {  // the row's first partner is the likeliest memo lookup
          const uint64_t ka = (uint64_t)rid0 << 32 | rid1;
          const uint64_t kb = (uint64_t)(e[ai].pos1 - e[pi].pos1) << 2 | (uint64_t)e[ai].dir << 1 | e[pi].dir;
          __builtin_prefetch(&mtab[mix(ka ^ mix(kb)) & (mcap - 1)], 1);
        }
      }
    }
  }
  bool memo_prefetch = true;  // (measured: first round 160 -> 156 ms at 4.5 Gbases)

  // The alignment memo: the request number of (rid0, rid1, q_off, dir0, dir1), filing the request if it is new.
  uint32_t request_of(TL &t, uint32_t rid0, uint32_t rid1, uint32_t q_off, uint8_t dir0, uint8_t dir1) {
    if (q_off >= (1u << 30)) overflow.store(true);  // (a Gbase-long read: the sequential replay's wider keys take over)
    const AKey key{(uint64_t)rid0 << 32 | rid1, (uint64_t)q_off << 2 | (uint64_t)dir0 << 1 | dir1};
    const uint64_t b32 = (uint64_t)q_off << 2 | (uint64_t)dir0 << 1 | dir1;
    uint32_t mval = 0;
    {
      size_t i = mix(key.a ^ mix(key.b)) & (mcap - 1);
      for (unsigned probes = 0;; i = (i + 1) & (mcap - 1)) {
        MSlot &ms = mtab[i];
        uint64_t a = ms.a.load(std::memory_order_acquire);
        if (a == 0 && ms.a.compare_exchange_strong(a, key.a, std::memory_order_acq_rel)) {  // ours: file the request
          if (t.qnext == t.qend) {
            t.cur_chunk.store(ALLOCATING, std::memory_order_seq_cst);  // (between the fetch_add and the publication
            t.qnext = nreq.fetch_add(QCHUNK, std::memory_order_seq_cst);  //  the submitter must not count the chunk)
            t.qend = t.qnext + QCHUNK;
            // unused slots of a chunk must still hold a valid key: pre-fill with this one
            for (uint32_t z = t.qnext; z < t.qend && z < reqcap; ++z)
              requests[z] = pgx_align_key{rid0, rid1, q_off, dir0, dir1, {0, 0}};
            t.cur_chunk.store(t.qnext, std::memory_order_seq_cst);
          }
          const uint32_t r = t.qnext++;
          if (r >= reqcap) overflow.store(true);
          else requests[r] = pgx_align_key{rid0, rid1, q_off, dir0, dir1, {0, 0}};
          if (t.qnext == t.qend) t.cur_chunk.store(NO_CHUNK, std::memory_order_release);  // chunk complete
          ms.bv.store(b32 << 32 | ((uint64_t)r + 1), std::memory_order_release);
          mval = r;
          break;
        }
        if (a == key.a) {  // (after a lost CAS `a` holds the winner's key)
          uint64_t bv = ms.bv.load(std::memory_order_acquire);
          while (bv == 0) {
            __builtin_ia32_pause();
            bv = ms.bv.load(std::memory_order_acquire);
          }
          if (bv >> 32 == b32) {
            mval = (uint32_t)bv - 1;
            break;
          }
        }
        if (++probes > 512) {
          overflow.store(true);
          mval = 0xFFFFFFFFu;
          break;
        }
      }
    }
    return mval;
  }

  void eval(uint32_t b, TL &t) {
    ++t.n_eval;
    BState &st = bs[b];
    const Own *old_own = st.own;
    const uint32_t n_old = st.nown;
    // The previous evaluation's insertions are NOT withdrawn up front: a transient "absent" would be visible to buckets
    // evaluated concurrently and nobody would tell them if the pair is simply re-inserted.  Instead this evaluation
    // ignores its own stale entries (owner == b but not inserted in this run) and withdraws the leftovers at the end.
    ++st.epoch;
    t.tmp_recs.clear(), t.tmp_own.clear();
    uint32_t lookups = 0, skips = 0;
    const size_t g0 = t.guesses.size();
    const Entry *e = v.entries.data() + v.start[b];
    const size_t n = v.start[b + 1] - v.start[b];
    t.contained.assign(n, 0);
    for (size_t hi = n - 1; hi > 0; --hi) {
      const size_t ai = hi - 1;
      if (t.contained[ai]) continue;
      const uint32_t rid0 = e[ai].rid, pos0 = e[ai].pos1, rlen0 = rlen[rid0];
      size_t got = 0;
      for (size_t pi = ai + 1; pi < n && got < bestn; ++pi) {
        if (t.contained[pi]) continue;
        const uint32_t rid1 = e[pi].rid;
        if (rid0 == rid1) continue;
        const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
        const uint32_t pid = pid_of(pair);
        PSlot &ps = ptab[pid];
        add_reader(ps, b, t);
        const uint64_t cur = ps.own.load(std::memory_order_seq_cst);
        bool present = cur != NOOWN && owner_of(cur) <= b;
        if (present && owner_of(cur) == b) {  // ours: only counts if inserted during THIS evaluation
          present = false;
          for (const Own &o : t.tmp_own)
            if (o.pid == pid) {
              present = true;
              break;
            }
        }
        if (present) {  // present in the table as this bucket sees it
          if (type_of(cur) == T_OVERLAP) ++got;
          ++skips;
          continue;
        }
        const uint32_t pos1 = e[pi].pos1, rlen1 = rlen[rid1];
        const uint32_t q_off = pos0 - pos1;
        ++lookups;
        const uint32_t mval = request_of(t, rid0, rid1, q_off, e[ai].dir, e[pi].dir);
        Verdict vd;
        const pgx_match *mm = nullptr;
        bool guessed = false;
        if (mval >= settled) {
          vd.accepted = true;
          vd.type = T_OVERLAP;
          if (predict && (rlen1 <= rlen0 - q_off || q_off < (uint32_t)(END_FUZZ * 2 - 8)))
            vd.type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
          guessed = true;
        } else {
          mm = &results[mval];
          vd = Replay::classify(*mm, rlen0, rlen1, q_off);
        }
        if (vd.accepted) {
          if (vd.type == T_OVERLAP) ++got;
          else if (vd.type == T_CONTAINS) t.contained[pi] = 1;
          else t.contained[ai] = 1;
          // take ownership unless an earlier bucket got in first (then this evaluation is stale and will be redone)
          uint64_t c2 = ps.own.load(std::memory_order_seq_cst);
          for (;;) {
            if (c2 != NOOWN && owner_of(c2) < b) {
              mark_dirty(b);
              break;
            }
            if (ps.own.compare_exchange_weak(c2, enc(b, vd.type), std::memory_order_seq_cst)) {
              if (c2 != NOOWN && owner_of(c2) > b) mark_dirty(owner_of(c2));
              break;
            }
          }
          t.tmp_own.push_back(Own{pid, vd.type});
          pgx_ovlp o;
          memset(&o, 0, sizeof(o));
          o.y0 = e[ai].y0, o.y1 = e[pi].y0, o.rl0 = rlen0, o.rl1 = rlen1;
          o.strand0 = e[ai].dir, o.strand1 = e[pi].dir, o.ovlp_type = vd.type;
          if (mm) o.match = *mm;
          if (guessed)
            t.guesses.push_back(Guess{b, st.epoch, mval, rlen0, rlen1, q_off,
                                      (pgx_ovlp *)(uintptr_t)t.tmp_recs.size(), vd.type});
          t.tmp_recs.push_back(o);
        }
        if (t.contained[ai]) break;
      }
    }
    // publish this evaluation's output (stable storage; the previous ranges stay valid for whoever still reads them)
    st.nrec = (uint32_t)t.tmp_recs.size(), st.nown = (uint32_t)t.tmp_own.size();
    st.lookups = lookups, st.skips = skips;
    st.recs = st.nrec ? t.recs.alloc(st.nrec) : nullptr;
    st.own = st.nown ? t.owned.alloc(st.nown) : nullptr;
    if (st.nrec) memcpy(st.recs, t.tmp_recs.data(), st.nrec * sizeof(pgx_ovlp));
    if (st.nown) memcpy(st.own, t.tmp_own.data(), st.nown * sizeof(Own));
    for (size_t g = g0; g < t.guesses.size(); ++g) t.guesses[g].rec = st.recs + (uintptr_t)t.guesses[g].rec;
    // what changed for later buckets?  readers only observe presence and "is a plain overlap"
    for (uint32_t i = 0; i < n_old; ++i) {
      PSlot &ps = ptab[old_own[i].pid];
      bool again = false, same = false;
      for (uint32_t j = 0; j < st.nown; ++j)
        if (st.own[j].pid == old_own[i].pid) {
          again = true;
          same = (st.own[j].type == T_OVERLAP) == (old_own[i].type == T_OVERLAP);
          break;
        }
      if (!again) {  // no longer inserted by this bucket: withdraw (unless somebody else owns it by now)
        uint64_t expect = enc(b, old_own[i].type);
        ps.own.compare_exchange_strong(expect, NOOWN, std::memory_order_seq_cst);
      }
      if (!again || !same) mark_readers_after(ps, b);
    }
    for (uint32_t i = 0; i < st.nown; ++i) {
      bool was = false;
      for (uint32_t j = 0; j < n_old; ++j)
        if (old_own[j].pid == st.own[i].pid) {
          was = true;
          break;
        }
      if (!was) mark_readers_after(ptab[st.own[i].pid], b);
    }
  }

  // Requests below this index sit in completely filled chunks: they can go to the GPU while the sweep continues.
  size_t complete_prefix() const {
    size_t m = std::min<size_t>(nreq.load(std::memory_order_seq_cst), reqcap);
    for (const TL &t : tl) {
      const uint32_t c = t.cur_chunk.load(std::memory_order_seq_cst);
      if (c == ALLOCATING) return 0;
      if (c != NO_CHUNK) m = std::min<size_t>(m, c);
    }
    return m;
  }
  std::function<void(size_t, size_t)> submit;  // ships requests [first, upto) to the GPU without waiting (thread 0 only)
  size_t submitted = 0;
  void maybe_submit() {
    const size_t p = complete_prefix();
    if (p > submitted && p - submitted >= std::max<size_t>(16384, (submitted - sweep_first) / 2)) {
      submit(submitted, p);
      submitted = p;
    }
  }
  size_t sweep_first = 0;

  void worker(unsigned ti) {
    const size_t nb = bs.size();
    TL &t = tl[ti];
    for (;;) {
      if (ti == 0 && submit) maybe_submit();
      const size_t c0 = cursor.fetch_add(block, std::memory_order_relaxed);
      if (c0 >= nb || overflow.load(std::memory_order_relaxed)) return;
      const size_t c1 = std::min(nb, c0 + block);
      if (dirty[c0].load(std::memory_order_relaxed)) prefetch_bucket((uint32_t)c0);
      for (size_t b = c0; b < c1; ++b) {
        if (b + 1 < c1 && dirty[b + 1].load(std::memory_order_relaxed)) prefetch_bucket((uint32_t)(b + 1));
        if (dirty[b].load(std::memory_order_relaxed) && dirty[b].exchange(0, std::memory_order_seq_cst)) eval((uint32_t)b, t);
      }
    }
  }

  // evaluate until no bucket is dirty; returns the number of alignments requested since the last settle()
  // dirty buckets, found eight flags at a time (the tail rounds of a GB-scale job have a handful of dirty buckets among
  // millions); at most `keep` of them are listed
  size_t scan_dirty(std::vector<uint32_t> &list, size_t keep) const {
    const size_t nb = bs.size();
    const uint8_t *f = reinterpret_cast<const uint8_t *>(dirty.get());  // (std::atomic<uint8_t> is one plain byte)
    size_t nd = 0, b = 0;
    list.clear();
    for (; b + 8 <= nb; b += 8) {
      uint64_t w;
      memcpy(&w, f + b, 8);
      if (!w) continue;
      for (size_t j = b; j < b + 8; ++j)
        if (f[j]) {
          if (nd < keep) list.push_back((uint32_t)j);
          ++nd;
        }
    }
    for (; b < nb; ++b)
      if (f[b]) {
        if (nd < keep) list.push_back((uint32_t)b);
        ++nd;
      }
    return nd;
  }

  size_t sweep(uint64_t *n_evals, unsigned *n_rounds) {
    std::vector<uint32_t> few;
    for (;;) {
      const size_t nd = scan_dirty(few, 48);
      if (!nd) break;
      const double r0 = now_ms();
      if (nd < 48 && nthr > 1) {
        // a handful of buckets: one thread, straight from the list (what they dirty in turn is found by the next scan)
        TL &t = tl[0];
        for (uint32_t b : few) {
          if (submit) maybe_submit();
          if (dirty[b].exchange(0, std::memory_order_seq_cst)) eval(b, t);
        }
      } else {
        cursor.store(0);
        if (nthr == 1) worker(0);
        else par_run(nthr, [&](unsigned ti) { worker(ti); });
      }
      if (trace) fprintf(stderr, "[pgx]   round: %zu dirty buckets, %.2f ms\n", nd, now_ms() - r0);
      if (overflow.load()) throw Overflow();
      if (n_rounds) ++*n_rounds;
    }
    if (n_evals) {
      *n_evals = 0;
      for (const TL &t : tl) *n_evals += t.n_eval;
    }
    for (TL &t : tl) {  // the rest of every private chunk stays filled with a duplicate key
      t.qnext = t.qend = 0;
      t.cur_chunk.store(NO_CHUNK, std::memory_order_relaxed);
    }
    return std::min<size_t>(nreq.load(), reqcap);
  }

  // results[first_req, upto) have been written by the GPU batch: right guesses get their record patched, wrong ones
  // make their bucket dirty.  Every thread settles the guesses it made itself.
  bool settle(size_t first_req, size_t upto) {
    settled = upto;
    std::atomic<bool> any{false};
    auto one = [&](unsigned ti) {
      TL &t = tl[ti];
      bool mine = false;
      for (const Guess &g : t.guesses) {
        if (g.req < first_req) continue;
        const pgx_match &m = results[g.req];
        const Verdict vd = Replay::classify(m, g.rlen0, g.rlen1, g.q_off);
        if (!vd.accepted || vd.type != g.type) dirty[g.bucket].store(1), mine = true;
        else if (bs[g.bucket].epoch == g.epoch) g.rec->match = m;  // (a newer evaluation has its own guesses)
      }
      t.guesses.clear();
      if (mine) any.store(true);
    };
    size_t ng = 0;
    for (const TL &t : tl) ng += t.guesses.size();
    if (ng < 4096) for (unsigned ti = 0; ti < nthr; ++ti) one(ti);
    else par_run(nthr, one);
    return any.load();
  }

  void collect(OvOut &out, uint64_t &lookups, uint64_t &skips) const {
    const size_t nb = bs.size();
    std::vector<size_t> first(nthr + 1, 0);  // output offset of each thread's slice of the bucket order
    lookups = skips = 0;
    for (unsigned ti = 0; ti < nthr; ++ti) {
      size_t c = 0;
      for (size_t b = nb / nthr * ti, e = ti + 1 == nthr ? nb : nb / nthr * (ti + 1); b < e; ++b)
        c += bs[b].nrec, lookups += bs[b].lookups, skips += bs[b].skips;
      first[ti + 1] = first[ti] + c;
    }
    out.alloc(first[nthr]);
    auto one = [&](unsigned ti) {
      pgx_ovlp *w = out.a + first[ti];
      for (size_t b = nb / nthr * ti, e = ti + 1 == nthr ? nb : nb / nthr * (ti + 1); b < e; ++b)
        if (bs[b].nrec) memcpy(w, bs[b].recs, (size_t)bs[b].nrec * sizeof(pgx_ovlp)), w += bs[b].nrec;
    };
    if (first[nthr] < (1u << 16)) for (unsigned ti = 0; ti < nthr; ++ti) one(ti);
    else par_run(nthr, one);
  }
};

void check_params(const pgx_overlap_params *p) {
  PGX_REQUIRE(p, PGX_EARG, "null params");
  PGX_REQUIRE(p->total_chunk > 0 && p->mychunk > 0 && p->mychunk <= p->total_chunk, PGX_EARG,
              "need 0 < mychunk <= total_chunk (shmr_overlap.c:328-329)");
  PGX_REQUIRE(p->align_bandwidth > 0 && p->align_bandwidth < (1 << 20), PGX_EARG, "bad align_bandwidth");
  PGX_REQUIRE(p->ovlp_upper >= 0 && p->mc_lower >= 0 && p->mc_upper >= 0, PGX_EARG, "negative bound");
}

// the lists either as host arrays (mmers / counts), as device arrays (dev), or -- a rank of a multi-GPU job -- as the pair
// records this chunk received from all index chunks (d_recs: device pointer, arrival order = insertion order)
// The FRONT of an overlap stage: count table, join, visit order -- everything up to the greedy walk; it reads only the lists and the
// parameters.  (Round 5 ran the front of chunk c + 1 on a second stream and host thread beside chunk c's walk -- pgx_overlap_prefetch_dev,
// commit 503bb51: bit-exact, and 7.12 s per c4 step against 7.06 without: the walk's small launches and the front's sorts
// time-share the GPU, only the 18 ms wait for the host's outer table was there to win.  Removed again; HISTORY.md "Round 5".)
struct Scratch {   // the big host tables of a stage: torn down on the housekeeping thread once the results are out
  PairTables pt;
  Visit visit;
  PreOuter pre;   // (its destructor joins the thread)
};
struct StageFront {
  Scratch *scratch = nullptr;
  DevicePairs dpairs;
  DevBuf<uint32_t> d_bids;
  bool placed = false, gpu_replay = false;
  pgx_overlap_stats s;
  double gpu_ms = 0, t0 = 0, t1 = 0;
  StageFront() { memset(&s, 0, sizeof(s)); }
  StageFront(const StageFront &) = delete;
  StageFront &operator=(const StageFront &) = delete;
  ~StageFront() {
    if (scratch) {
      Scratch *z = scratch;
      defer_destroy([z] { delete z; });
    }
  }
};
void overlap_front(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
                   const pgx_overlap_params *p, const DeviceLists *dev, const pgx_pair_rec *d_recs, size_t n_recs, StageFront &f) {
  pgx_overlap_stats &s = f.s;
  const double t0 = f.t0 = now_ms();
  double &gpu_ms = f.gpu_ms;
  MemTag mem_tag("overlap.join");
  Scratch *scratch = f.scratch = new Scratch;
  PairTables &pt = scratch->pt;
  // The greedy walk itself runs on the GPU (pgx_replay.hip) from 0.2 M pair records on, where it is as fast as or faster than
  // the multi-threaded host replay below and does not lean on the host cores, which the ranks of a multi-GPU job share
  // (overlap stage, 30x sets: E. coli-size 10.2 vs 10.1 ms per step; 5 Mb 18.7 vs 19.7 ms; 20 Mb 44 vs 53 ms; 80 Mb 134 vs 217 ms;
  // 150 Mb 0.23 vs 0.45 s; below that a sweep is bound by the latency of single bucket evaluations and kernel launches:
  // 1 Mb 11.8 vs 9.8 ms, 0.3 Mb 10.9 vs 6.1 ms).  PGX_GPU_REPLAY=1 / 0 forces either one; the host replay is also the
  // fallback for jobs the device tables' encodings do not hold.
  const int gpu_replay_env = getenv("PGX_GPU_REPLAY") ? atoi(getenv("PGX_GPU_REPLAY")) : -1;
  static const size_t gpu_replay_min = 200000;   // pair records from which the device replay wins (tools/crossover.py)
  bool &gpu_replay = f.gpu_replay;
  gpu_replay = gpu_replay_env != 0;  // (decided once the join has counted the records)
  const bool trace = getenv("PGX_TRACE") != nullptr;
  DevicePairs &dpairs = f.dpairs;
  static const bool early_outer = true;   // the outer khash table is replayed by a host thread DURING the join
  // the visit order on the device (pgx_visit.hip): the join's tables stay in HBM, the inner khash tables are replayed there, the
  // host only replays the outer one.  PGX_DEV_VISIT=0: the round-2 form (tables downloaded, inner tables by host threads).
  const bool dev_visit = gpu_replay && early_outer && !(getenv("PGX_DEV_VISIT") && atoi(getenv("PGX_DEV_VISIT")) == 0);
  const unsigned jflags = PAIRS_ORD_TABLES | (gpu_replay ? PAIRS_LAZY_RECORDS : 0u) | (dev_visit ? PAIRS_DEV_TABLES : 0u);
  EarlyFn early;
  if (early_outer) early = [scratch, &pt](EarlyGroups &&g) { scratch->pre.start(std::move(g), pt.n_rec); };
  if (d_recs)
    dev_pairs_from_records(d_recs, n_recs, pt, gpu_replay ? &dpairs : nullptr, jflags, early, db);
  else
    dev_build_pairs(db->d_rlen.p, mmers, n_mm, counts, n_counts,
                    PairParams{(uint32_t)p->total_chunk, (uint32_t)p->mychunk, (uint32_t)p->mc_lower, (uint32_t)p->mc_upper,
                               (uint32_t)db->rlen_by_rid.size()},
                    pt, jflags, dev ? dev->d_top : nullptr, dev ? dev->d_mc : nullptr, gpu_replay ? &dpairs : nullptr, early, db);
  pgx::sync();
  s.n_pair_records = pt.n_rec;
  if (gpu_replay_env < 0) gpu_replay = pt.n_rec >= gpu_replay_min;
  if (!gpu_replay) {
    pairs_fetch_tables(dpairs, pt);
    pairs_fetch_records(dpairs, pt);   // (kept on the device in case the device replay ran: the host replay reads them)
    dpairs = DevicePairs();
  }
  const double t1 = f.t1 = now_ms();
  gpu_ms += t1 - t0;
  NodePin pin;  // from here on this thread and its helper threads stay on one memory node
  Visit &visit = scratch->visit;
  if (trace) fprintf(stderr, "[pgx]   pinned to a memory node at +%.2f ms after the join\n", now_ms() - t1);
  if (gpu_replay && dpairs.valid) {
    DevBuf<uint32_t> &d_bids = f.d_bids;
    bool &placed = f.placed;
    if (dpairs.tables) {
      PreOuter &pre = scratch->pre;
      const uint32_t wave_max = getenv("PGX_VISIT_WAVE_MAX") ? (uint32_t)std::min<long>(atol(getenv("PGX_VISIT_WAVE_MAX")), VISIT_WAVE_MAX) : VISIT_WAVE_MAX;   // (tests: force the fall-back)
      bool ok = pre.started && pre.eg.n == dpairs.n_groups && dpairs.max_group_buckets <= wave_max;
      if (ok) {
        DevVisit dv;
        dev_visit_inner(dpairs, (uint32_t)p->ovlp_upper, dv);            // (enqueued: the GPU replays the inner tables ...
        if (pt.n_rec >= ((size_t)2 << 20)) dev_align_prepare(db);        //  ... and packs the reads for the alignments ...
        pre.join();                                                      //  ... while the outer table finishes here)
        const double tw = now_ms();
        for (size_t i = 0; ok && i < dpairs.key_sample.size(); ++i) ok = pre.eg.keys[i * KEY_SAMPLE_STRIDE] == dpairs.key_sample[i];
        ok = ok && ((size_t)pre.eg.last_first == (size_t)dpairs.last_gfirst);
        if (ok) {
          size_t nbv = 0, nev = 0;
          dev_visit_place(dpairs, dv, pre.table.slot, pre.table.nb, d_bids, &nbv, &nev);
          visit.n_buckets = nbv, visit.n_entries = nev, visit.on_device = true, visit.n_groups = 0;
          placed = true;
          s.device_visit = 1 + dpairs.n_big_groups;   // (the tables stay until the device replay has succeeded: its fall-back, the host replay, fetches them)
          if (trace)
            fprintf(stderr, "[pgx]   visit on the device: waited %.2f ms for the outer table (host thread %.2f ms, %u slots), placed in %.2f ms\n", tw - t1,
                    pre.ms, pre.table.nb, now_ms() - tw);
        } else {
          fprintf(stderr, "[pgx] note: the early outer-table keys do not match the join's group tables; the host builds the visit order\n");
        }
      } else if (trace) {
        fprintf(stderr, "[pgx]   visit: host path (early outer table %s, largest group %u buckets)\n", pre.started ? "running" : "not started",
                dpairs.max_group_buckets);
      }
      if (!placed) pairs_fetch_tables(dpairs, pt);
    }
    if (!placed) build_visit(pt, (uint32_t)p->ovlp_upper, visit, true, &scratch->pre);
    s.n_buckets = visit.n_buckets;
    if (trace)
      fprintf(stderr, "[pgx] GPU join: %zu records, %zu buckets, %zu key0 groups in %.2f ms; visit order (%llu buckets, ids only) in %.2f ms\n",
              pt.n_rec, pt.n_buckets, pt.n_groups, t1 - t0, (unsigned long long)s.n_buckets, now_ms() - t1);
    if (trace && atoi(getenv("PGX_TRACE")) >= 2 && visit.n_buckets && !visit.on_device && pt.on_host) {  // bucket sizes: a pass of the device replay lasts as long as its largest bucket
      std::vector<uint32_t> sz(visit.n_buckets);
      for (size_t i = 0; i < visit.n_buckets; ++i) sz[i] = pt.bstart[visit.bids[i] + 1] - pt.bstart[visit.bids[i]];
      std::sort(sz.begin(), sz.end());
      fprintf(stderr, "[pgx]   bucket sizes: median %u, 90 %% %u, 99 %% %u, 99.9 %% %u, max %u\n", sz[sz.size() / 2], sz[sz.size() * 9 / 10],
              sz[sz.size() * 99 / 100], sz[sz.size() * 999 / 1000], sz.back());
    }
    if (visit.on_device && !placed)
      dev_place_bids(visit.ids_all.data(), visit.ids_all.size(), visit.psrc.data(), visit.pcnt.data(), visit.pdst.data(), visit.n_groups,
                     visit.n_buckets, d_bids);
  }
}

void run_overlap(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts,
                 const pgx_overlap_params *p, OvOut &out, pgx_overlap_stats *st, const DeviceLists *dev = nullptr,
                 const pgx_pair_rec *d_recs = nullptr, size_t n_recs = 0) {
  StageFront front;
  dev_cache_age();
  overlap_front(db, mmers, n_mm, counts, n_counts, p, dev, d_recs, n_recs, front);
  pgx_overlap_stats s = front.s;
  const double t0 = front.t0, t1 = front.t1;
  double gpu_ms = front.gpu_ms;
  MemTag mem_tag("overlap.join");
  Scratch *scratch = front.scratch;
  PairTables &pt = scratch->pt;
  bool gpu_replay = front.gpu_replay;
  const bool trace = getenv("PGX_TRACE") != nullptr;
  const bool predict = true;   // the type of a pending alignment is guessed from the geometry (predict_contained)
  DevicePairs &dpairs = front.dpairs;
  NodePin pin;  // from here on the caller and its helper threads stay on one memory node
  Visit &visit = scratch->visit;
  if (gpu_replay && dpairs.valid) {
    DevBuf<uint32_t> &d_bids = front.d_bids;
    const double r0 = now_ms();
    size_t nrec = 0;
    pgx_overlap_stats rs;
    memset(&rs, 0, sizeof(rs));
    if (dev_replay(db, dpairs, visit.on_device ? nullptr : visit.bids.data(), visit.on_device ? d_bids.p : nullptr, visit.n_buckets,
                   visit.n_entries, (uint32_t)(uint8_t)p->bestn,
                   p->align_bandwidth, predict, (uint32_t)p->ovlp_upper,
                   [&](size_t n) -> pgx_ovlp * {
                     if (record_sink()) {   // (the records go from the device to the sink: no host array)
                       out_free(out.a), out.a = nullptr, out.n = n;
                       return nullptr;
                     }
                     out.alloc(n);
                     return out.a;
                   }, &nrec, &rs, trace)) {
      s.n_align_needed = rs.n_align_needed, s.n_seen_skip = rs.n_seen_skip, s.n_align_gpu = rs.n_align_gpu, s.rounds = rs.rounds;
      s.n_evaluations = rs.n_evaluations, s.device_replay = 1;
      s.replay_attempts = rs.replay_attempts, s.stream_checksum = rs.stream_checksum;
      gpu_ms += now_ms() - r0;
      timing_flush();
      if (trace) fprintf(stderr, "[pgx] stage total %.2f ms\n", now_ms() - t0);
      s.n_records = out.n;
      s.gpu_ms = gpu_ms;
      s.host_ms = now_ms() - t0 - gpu_ms;
      if (st) *st = s;
      return;
    }
    pairs_fetch_tables(dpairs, pt);    // the device replay gave up: the host replay needs the tables and the records
    pairs_fetch_records(dpairs, pt);
    dpairs = DevicePairs();
  }
  build_visit(pt, (uint32_t)p->ovlp_upper, visit, false, &scratch->pre);
  s.n_buckets = visit.start.size() - 1;
  if (trace)
    fprintf(stderr, "[pgx] GPU join: %zu records, %zu buckets, %zu key0 groups in %.2f ms; visit order (%llu buckets) in %.2f ms\n",
            pt.n_rec, pt.n_buckets, pt.n_groups, t1 - t0, (unsigned long long)s.n_buckets, now_ms() - t1);
  auto align_batch = [&](const pgx_align_key *keys, size_t nreq, pgx_match *res) {  // results land in the replay's table
    const double g0 = now_ms();
    pgx_align_key *d_keys = ws<pgx_align_key>("ov.keys", nreq);
    pgx_match *d_res = ws<pgx_match>("ov.res", nreq);
    PGX_HIP(hipMemcpyAsync(d_keys, keys, nreq * sizeof(pgx_align_key), hipMemcpyHostToDevice, ctx().stream));
    dev_align(db, d_keys, nreq, p->align_bandwidth, d_res);
    PGX_HIP(hipMemcpyAsync(res, d_res, nreq * sizeof(pgx_match), hipMemcpyDeviceToHost, ctx().stream));
    pgx::sync();
    gpu_ms += now_ms() - g0;
    s.n_align_gpu += nreq;
  };
  // 24 threads measured best on a 64-core node at both ends (E. coli set: 8 -> 15.1 ms, 16 -> 12.3, 24 -> 10.3, 48 -> 10.0,
  // 64 -> 16.6 per step; 4.5 Gbases: 16 -> 620 ms, 24 -> 539, 32 -> 571); the ranks of a multi-process job share the host
  unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  {
    cpu_set_t allowed;  // (a container may grant far fewer CPUs than the machine has)
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0 && CPU_COUNT(&allowed) > 0)
      threads = std::min(threads, (unsigned)CPU_COUNT(&allowed));
  }
  if (const char *lw = getenv("LOCAL_WORLD_SIZE")) threads = std::max(4u, threads / (2u * (unsigned)std::max(1, atoi(lw))));
  threads = std::min(24u, threads);
  if (const char *tv = getenv("PGX_THREADS")) threads = (unsigned)std::max(1, atoi(tv));
  // the shared-table protocol costs a locked operation per examination (plus one per insertion) and a thread team per round; measured against
  // the sequential replay with 16 threads: 4.2 s -> 0.25 s for the first sweep at 4.5 Gbases, 11.9 -> 7 ms of sweeps at
  // 75 Mbases (200 k entries); below ~50 k entries the team start-up dominates
  size_t par_min = 50000;
  if (const char *pm = getenv("PGX_PAR_MIN")) par_min = (size_t)atoll(pm);
  if (visit.entries.size() < par_min) threads = 1;
  bool done = false;
  if (threads > 1) {
    try {
      const double c0 = now_ms();
      // (the replay tables too: but they refer to the visit list, so they go first)
      ParReplay *rpp = new ParReplay(visit, db->rlen_by_rid, (uint32_t)(uint8_t)p->bestn, threads);
      struct DeferReplay {
        ParReplay *r;
        ~DeferReplay() {
          ParReplay *z = r;
          defer_destroy([z] { delete z; });
        }
      } defer_replay{rpp};
      ParReplay &rp = *rpp;
      rp.predict = predict;
      rp.trace = trace;
      if (const char *bv = getenv("PGX_BLOCK")) rp.block = (size_t)std::max(1, atoi(bv));
      if (trace) fprintf(stderr, "[pgx] parallel replay tables set up in %.2f ms; t = +%.2f ms\n", now_ms() - c0, now_ms() - t0);
      size_t first_req = 0;
      double settle_ms = 0;
      // alignment batches go to the GPU while the sweep that files them is still running; the results come back once,
      // after the sweep
      struct Batch {
        DevBuf<pgx_align_key> keys;
        DevBuf<pgx_match> res;
        size_t first, n;
      };
      std::vector<Batch> inflight;
      rp.submit = [&](size_t first, size_t upto) {
        const double g0 = now_ms();
        Batch b{DevBuf<pgx_align_key>(upto - first), DevBuf<pgx_match>(upto - first), first, upto - first};
        PGX_HIP(hipMemcpyAsync(b.keys.p, rp.requests.data() + first, b.n * sizeof(pgx_align_key), hipMemcpyHostToDevice,
                               ctx().stream));
        dev_align(db, b.keys.p, b.n, p->align_bandwidth, b.res.p);
        inflight.push_back(std::move(b));
        s.n_align_gpu += upto - first;
        gpu_ms += now_ms() - g0;
        if (trace) fprintf(stderr, "[pgx]   submitted %zu requests in %.2f ms at t = +%.2f ms\n", upto - first, now_ms() - g0, now_ms() - t0);
      };
      for (;;) {
        const double p0 = now_ms();
        uint64_t ev = 0;
        unsigned rounds = 0;
        rp.sweep_first = rp.submitted = first_req;
        const size_t upto = rp.sweep(&ev, &rounds);
        ++s.rounds;
        s.n_evaluations = ev;
        if (trace)
          fprintf(stderr, "[pgx] parallel sweep %u (%u threads): %u rounds, %llu evaluations so far, %.2f ms, %zu requests (%zu already on the GPU)\n",
                  s.rounds, threads, rounds, (unsigned long long)ev, now_ms() - p0, upto - first_req, rp.submitted - first_req);
        if (upto == first_req) break;
        const double g0 = now_ms();
        if (upto > rp.submitted) rp.submit(rp.submitted, upto);
        for (Batch &b : inflight)
          PGX_HIP(hipMemcpyAsync(rp.results.data() + b.first, b.res.p, b.n * sizeof(pgx_match), hipMemcpyDeviceToHost, ctx().stream));
        pgx::sync();
        inflight.clear();
        gpu_ms += now_ms() - g0;
        if (trace) fprintf(stderr, "[pgx]   waited %.2f ms for the GPU after the sweep\n", now_ms() - g0);
        const double s0 = now_ms();
        const bool any = rp.settle(first_req, upto);
        settle_ms += now_ms() - s0;
        first_req = upto;
        if (!any) break;
      }
      const double k0 = now_ms();
      rp.collect(out, s.n_align_needed, s.n_seen_skip);
      if (trace) fprintf(stderr, "[pgx] settle %.2f ms total, collect %.2f ms; t = +%.2f ms\n", settle_ms, now_ms() - k0, now_ms() - t0);
      done = true;
    } catch (const ParReplay::Overflow &) {
      fprintf(stderr, "[pgx] note: parallel replay tables overflowed; falling back to the sequential replay\n");
      s.rounds = 0, s.n_align_gpu = 0;
    }
  }
  if (!done) {
    Replay rp(visit, db->rlen_by_rid, (uint32_t)(uint8_t)p->bestn);  // bestn is a uint8_t in the reference (:245)
    rp.predict = predict;
    for (;;) {
      const double p0 = now_ms();
      const uint64_t ev0 = rp.n_eval;
      const size_t nreq = rp.sweep();
      ++s.rounds;
      if (trace)
        fprintf(stderr, "[pgx] replay sweep %u: %llu buckets evaluated in %.2f ms, %zu requests\n", s.rounds,
                (unsigned long long)(rp.n_eval - ev0), now_ms() - p0, nreq);
      if (nreq == 0) break;
      align_batch(rp.requests.data(), nreq, rp.result_slots());
      if (!rp.settle()) break;  // every guess was right: the replay is exact
    }
    rp.collect(out, s.n_align_needed, s.n_seen_skip);
    s.n_evaluations = rp.n_eval;
  }
  const double tf0 = now_ms();
  timing_flush();
  if (trace) fprintf(stderr, "[pgx] stage total %.2f ms (timing flush %.2f ms)\n", now_ms() - t0, now_ms() - tf0);
  s.n_records = out.n;
  for (size_t i = 0; i < out.n; ++i) s.stream_checksum += record_checksum(out.a[i], i);   // (the host replay serves small sets)
  s.gpu_ms = gpu_ms;
  s.host_ms = now_ms() - t0 - gpu_ms;
  if (st) *st = s;
}

}  // namespace

// what the file-level entry points (pgx_served.cpp) see of the stage
namespace pgx {
void overlap_check_params(const pgx_overlap_params *p) { check_params(p); }
void overlap_stage(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts, size_t n_counts, const pgx_overlap_params *p,
                   OvOut &out, pgx_overlap_stats *st, const DeviceLists *dev) {
  run_overlap(db, mmers, n_mm, counts, n_counts, p, out, st, dev);
}
RecordSink *&record_sink() {
  static RecordSink *s = nullptr;
  return s;
}
}  // namespace pgx

extern "C" {

int pgx_khash_slot_order(const uint64_t *keys, size_t n, uint64_t *out) {
  try {
    PGX_REQUIRE((keys && out) || n == 0, PGX_EARG, "pgx_khash_slot_order: null argument");
    PGX_REQUIRE(n < (1ULL << 30), PGX_EARG, "pgx_khash_slot_order: too many keys");
    DistinctSlotTable t;
    for (size_t i = 0; i < n; ++i) {
      if (i + 8 < n) t.prefetch(keys[i + 8]);
      t.put_new(keys[i], (uint32_t)i);
    }
    size_t m = 0;
    for (uint32_t s0 = 0; s0 < t.nb; ++s0)
      if (t.is_used(s0)) out[m++] = keys[t.id_at(s0)];
  } catch (const Fail &f) {
    return f.code;
  }
  return PGX_OK;
}

int pgx_khash_slot_order_ex(const uint64_t *keys, size_t n, int touch, uint64_t *out) {
  try {
    PGX_REQUIRE((keys && out) || n == 0, PGX_EARG, "pgx_khash_slot_order_ex: null argument");
    PGX_REQUIRE(n < (1ULL << 30), PGX_EARG, "pgx_khash_slot_order_ex: too many keys");
    if (n == 0) return PGX_OK;
    size_t m = 0;
    DistinctSlotTable t;
    for (size_t i = 0; i < n; ++i) {
      if (i + 8 < n) t.prefetch(keys[i + 8]);
      t.put_new(keys[i], (uint32_t)i);
    }
    if (touch) t.touch();
    for (uint32_t s0 = 0; s0 < t.nb; ++s0)
      if (t.is_used(s0)) out[m++] = keys[t.id_at(s0)];
    PGX_REQUIRE(m == n, PGX_ESTATE, "pgx_khash_slot_order_ex: %zu of %zu keys placed (are the keys distinct?)", m, n);
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    return PGX_EARG;
  }
  return PGX_OK;
}

int pgx_overlap_resident(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
                         size_t n_counts, const pgx_overlap_params *p, pgx_ovlp **out, size_t *n_out,
                         pgx_overlap_stats *stats) {
  try {
    require_ready();
    PGX_REQUIRE(db && out && n_out && (n_mm == 0 || mmers) && (n_counts == 0 || counts), PGX_EARG,
                "pgx_overlap_resident: null argument");
    check_params(p);
    OvOut v;
    run_overlap(db, mmers, n_mm, counts, n_counts, p, v, stats);
    *n_out = v.n;
    *out = v.release();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

// index + overlap of ONE chunk with the shimmer list and the counts handed over in HBM (no download + upload between the
// stages); anything the fused index path does not cover falls back to the two-stage hand-over through host arrays
int pgx_index_overlap_resident(pgx_seqdb *db, const pgx_index_params *ip, const pgx_overlap_params *op, int want_index_arrays,
                               pgx_index_result *index_out, pgx_ovlp **out, size_t *n_out, pgx_overlap_stats *stats) {
  try {
    require_ready();
    PGX_REQUIRE(db && ip && op && index_out && out && n_out, PGX_EARG, "pgx_index_overlap_resident: null argument");
    PGX_REQUIRE(ip->total_chunk == 1 && ip->mychunk == 1, PGX_EARG,
                "pgx_index_overlap_resident is the single-index-chunk pipeline (other chunks' lists would be missing)");
    check_params(op);
    DeviceIndex dev;
    index_stage(db, ip, index_out, &dev, want_index_arrays != 0);
    OvOut v;
    if (dev.valid) {
      const DeviceLists dl{dev.d_top, dev.mc.p};
      run_overlap(db, nullptr, dev.n_top, nullptr, dev.n_mc, op, v, stats, &dl);
    } else {  // (want_l0, ambiguous parameters ...: the general index path has already produced host arrays)
      run_overlap(db, index_out->top, index_out->n_top, index_out->top_mc, index_out->n_top_mc, op, v, stats);
    }
    *n_out = v.n;
    *out = v.release();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

// ---- multi-GPU hand-over on device pointers (include/pgx.h) -----------------------------------------------------
int pgx_overlap_resident_dev(pgx_seqdb *db, const pgx_mm128 *d_mmers, size_t n_mm, const pgx_mm_count *d_counts,
                             size_t n_counts, const pgx_overlap_params *p, pgx_ovlp **out, size_t *n_out,
                             pgx_overlap_stats *stats) {
  try {
    require_ready();
    PGX_REQUIRE(db && out && n_out && (n_mm == 0 || d_mmers) && (n_counts == 0 || d_counts), PGX_EARG,
                "pgx_overlap_resident_dev: null argument");
    check_params(p);
    OvOut v;
    const DeviceLists dl{d_mmers, d_counts};
    run_overlap(db, nullptr, n_mm, nullptr, n_counts, p, v, stats, &dl);
    *n_out = v.n;
    *out = v.release();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_pairs_prepare_dev(pgx_seqdb *db, const pgx_mm128 *d_top, size_t n_top, const pgx_mm_count *d_counts_all,
                          size_t n_counts_all, int mc_lower, int mc_upper, int64_t *first_strict) {
  try {
    require_ready();
    PGX_REQUIRE(db && first_strict && (n_top == 0 || d_top) && (n_counts_all == 0 || d_counts_all) && mc_lower >= 0 && mc_upper >= 0,
                PGX_EARG, "pgx_pairs_prepare_dev: bad argument");
    *first_strict = dev_pairs_prepare(db->d_rlen.p, (uint32_t)db->rlen_by_rid.size(), d_top, n_top, d_counts_all, n_counts_all,
                                      (uint32_t)mc_lower, (uint32_t)mc_upper);
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_pairs_scatter_dev(pgx_seqdb *db, int total_chunk, int64_t start, const pgx_pair_rec **d_send, uint64_t *send_counts) {
  try {
    require_ready();
    PGX_REQUIRE(db && d_send && send_counts && total_chunk > 0, PGX_EARG, "pgx_pairs_scatter_dev: bad argument");
    dev_pairs_scatter(db->d_rlen.p, (uint32_t)total_chunk, start, d_send, send_counts);
    timing_flush();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_overlap_records_dev(pgx_seqdb *db, const pgx_pair_rec *d_records, size_t n_records, const pgx_overlap_params *p,
                            pgx_ovlp **out, size_t *n_out, pgx_overlap_stats *stats) {
  try {
    require_ready();
    PGX_REQUIRE(db && out && n_out && (n_records == 0 || d_records), PGX_EARG, "pgx_overlap_records_dev: null argument");
    check_params(p);
    OvOut v;
    static const pgx_pair_rec none{};   // (an empty record set still takes the records path)
    run_overlap(db, nullptr, 0, nullptr, 0, p, v, stats, nullptr, n_records ? d_records : &none, n_records);
    *n_out = v.n;
    *out = v.release();
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

}  // extern "C"
