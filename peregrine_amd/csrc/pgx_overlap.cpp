// pgx_overlap.cpp -- the overlap stage (driver: front = count table, join, visit order; then the greedy walk on the device, pgx_replay.hip, or on
// the host, pgx_host_replay.h; the resident C entry points): what main() of /root/reference/src/shmr_overlap.c:233-419 does for one
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

#include "pgx_host_tables.h"   // maps, visit list + build_visit, thread team, pinned pools
#include "pgx_host_replay.h"   // Replay / ParReplay: the greedy walk on the host

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
        if (pt.n_rec >= ((size_t)2 << 20)) replay_preclear();            //  ... and clears the replay's tables (round 6: 7-8 ms of a full-size chunk) ...
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
  struct DropPre {   // (tables the front cleared ahead of time for a device replay that then did not run: an exception, the host path)
    ~DropPre() { replay_drop_precleared(); }
  } drop_pre;
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
