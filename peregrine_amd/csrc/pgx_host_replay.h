// pgx_host_replay.h -- the greedy best-n walk (shimmer_to_overlap + the seen-pair table, /root/reference/src/shmr_overlap.c:52-228) on the HOST
// (included by pgx_overlap.cpp only, inside its unnamed namespace, behind pgx_host_tables.h): Replay, the sequential fixed point, and ParReplay, the same
// with a thread team over shared lock-free tables.  Sets below 200,000 pair records and the fall-back of the device replay (pgx_replay.hip) run here.
#pragma once

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

