// pgx_host_tables.h -- host side of the overlap stage (included by pgx_overlap.cpp only, inside its unnamed namespace): the hash maps of the host
// replay, the visit list (Entry / Visit) and its construction from the GPU join's tables (build_visit: klib-khash slot order of both table levels,
// /root/reference/src/khash.h:232-336, src/shmr_overlap.c:206-217), the thread team and the pinned block pools.
#pragma once


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

