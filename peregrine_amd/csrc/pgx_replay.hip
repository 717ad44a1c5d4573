// pgx_replay.hip -- the greedy best-n walk over the shimmer-pair buckets (shimmer_to_overlap + the seen-pair table of
// build_ovlp, /root/reference/src/shmr_overlap.c:52-228) on the GPU, one lane per bucket.
//
// The reference visits the buckets once, in order, sharing one seen-pair table.  A bucket's evaluation is a pure function of
// (a) which of the pairs it examines were inserted by EARLIER buckets (and with which type) and (b) the alignment results it
// looks up.  So the walk is solved as a fixed point (the same formulation as the host replay in pgx_overlap.cpp, which stays
// as the fallback), in phases separated by kernel boundaries -- no intra-kernel ordering is needed anywhere:
//
//   k_eval    every dirty bucket of a window re-runs shimmer_to_overlap against the pair table AS IT STANDS (read-only in
//             this kernel): pair present <=> owner < bucket.  It registers itself as a reader of every pair it examines
//             (lock-free push), guesses unknown alignments (accepted; type predicted from the geometry) and leaves the
//             list of pairs it would insert.
//   k_update  applies the evaluated buckets' lists to the table (compare-and-swap, lowest bucket wins), withdraws what a
//             bucket no longer inserts, and marks dirty every later reader of a pair whose state changed, the displaced
//             owner, and the bucket itself when an earlier bucket got in first.
//   ... repeated window by window until no bucket is dirty; then
//   k_file    files the alignments the lists still need in the memo table (insert-only) and numbers the requests,
//   dev_align the banded O(ND) kernel (pgx_align.hip) on the new requests,
//   k_settle  checks every guess against its result; a wrong guess makes the bucket dirty again
//   ... until a sweep files nothing or every guess was right; k_emit writes the ovlp_t records in bucket order.
//
// At the fixed point every bucket was last evaluated against a table that has not changed since in any way it could
// observe, and the owner of a pair is the lowest bucket inserting it: by induction over the bucket order that is the
// sequential walk.  Changes only ever propagate to LATER buckets, so the lowest unstable bucket rises monotonically.
#include <chrono>
#include <optional>

#include <hipcub/hipcub.hpp>

#include "pgx_replay.h"

namespace pgx {
using namespace rp;
namespace {

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
uint32_t pow2_at_least(size_t x) {
  size_t c = 1024;
  while (c < x) c <<= 1;
  return (uint32_t)c;
}
unsigned cdiv256(size_t n) { return (unsigned)((n + 255) / 256); }

// Round 6: the three hash tables of a stage (hot pair table, reader lists, memo: 38.6 GB at full-size configs[3]) must start all zero, and clearing them
// takes 7-8 ms -- on a GPU that sits idle for ~18 ms of every chunk while the host finishes the outer khash table (pgx_overlap.cpp: overlap_front).
// So the front allocates and clears tables of the sizes the LAST stage used inside that window (replay_preclear); replay_attempt takes them over when
// its own sizes come out the same (the chunks of a job are alike, the sizes are powers of two) and clears its own otherwise.
struct PreCleared {
  DevBuf<PHot> ph;
  DevBuf<PCold> pc;
  DevBuf<MSlot> mt;
  uint32_t pcap = 0, mcap = 0;
  size_t ccap = 0;
  bool valid = false;
  void drop() {
    ph.release(), pc.release(), mt.release();
    valid = false;
  }
};
PreCleared g_pre;
uint32_t g_last_pcap = 0, g_last_mcap = 0;   // what the last successful stage of the process used
size_t g_last_ccap = 0;
ShutdownHook h_pre([] { g_pre.drop(), g_last_pcap = g_last_mcap = 0, g_last_ccap = 0; });

// one attempt with the given table sizes (multiples of the defaults); returns 0, or the OV_* bits of what overflowed
uint32_t replay_attempt(const pgx_seqdb *db, const DevicePairs &dp, const uint32_t *visit_bids, const uint32_t *d_bids, size_t nb, size_t n_entries,
                        uint32_t bestn, int band, bool predict, const std::function<pgx_ovlp *(size_t)> &alloc_out,
                        size_t *n_out, pgx_overlap_stats *st, bool trace, const double *mult, double *usage) {
  const double t0 = now_ms();
  hipStream_t s = ctx().stream;
  const size_t ne = std::max<size_t>(n_entries, 1024);
  MemTag mem_tag("replay.other");
  R r;
  memset(&r, 0, sizeof(r));
  r.nb = (uint32_t)nb;
  DevBuf<uint32_t> bid(d_bids ? 0 : nb);   // (d_bids: the visit list was assembled on the device, dev_place_bids)
  if (!d_bids) bid.upload(visit_bids, nb);
  r.bid = d_bids ? d_bids : bid.p, r.bstart = dp.bstart.p, r.y0 = dp.y0.p, r.dir = dp.dir.p, r.rlen = db->d_rlen.p;
  const uint32_t pcap = pow2_at_least((size_t)(ne * mult[3])), mcap = pow2_at_least((size_t)(ne * mult[4]));
  DevBuf<PHot> ph;
  DevBuf<PCold> pc;
  DevBuf<MSlot> mt;
  // (what the pre-cleared tables hold was free memory when round 5 measured the rule below: counted as free here too)
  const size_t pre_hold = g_pre.valid ? (size_t)g_pre.pcap * sizeof(PHot) + g_pre.ccap * sizeof(PCold) + (size_t)g_pre.mcap * sizeof(MSlot) : 0;
  bool precleared = false;
  if (g_pre.valid && g_pre.pcap == pcap && g_pre.mcap == mcap && !getenv("PGX_REPLAY_COLD_SHIFT")) {   // (its reader lists: checked below, once the shift is known)
    ph = std::move(g_pre.ph), mt = std::move(g_pre.mt);
    precleared = true;
  } else {
    MemTag t1("replay.pair_table_hot");
    ph.alloc(pcap);
  }
  // The hot table wants a load of <= 0.4 (dev_replay) and the reader lists are 16 x as large per slot (34 GB for a full-size configs[3]
  // chunk).  Where that is more than a third of the free device memory, 2 (4, 8) neighbouring hot slots share one list: all sharing costs is
  // that a change of one pair also marks the other's later readers dirty -- a spurious re-evaluation, never a missed one (measured at c4:
  // + 19 % evaluations, + 1.3 % step time for 17 GB less).  PGX_REPLAY_COLD_SHIFT forces the shift (tests: 0 .. 3).
  r.cshift = 0;
  if (getenv("PGX_REPLAY_COLD_SHIFT")) {
    r.cshift = (uint32_t)std::min(3, std::max(0, atoi(getenv("PGX_REPLAY_COLD_SHIFT"))));
  } else {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      while (r.cshift < 3 && ((size_t)pcap >> r.cshift) * sizeof(PCold) > (free_b + dev_cache_free_bytes() + pre_hold) / 3) ++r.cshift;   // (the last stage's tables are in the cache)
  }
  const size_t ccap = ((size_t)pcap >> r.cshift) + 1;
  const bool pc_pre = precleared && g_pre.ccap == ccap;
  if (pc_pre) {
    pc = std::move(g_pre.pc);
  } else {
    MemTag t2("replay.pair_table_readers");
    g_pre.pc.release();
    pc.alloc(ccap);
  }
  if (!precleared) {
    MemTag t3("replay.memo_table");
    mt.alloc(mcap);
  }
  g_pre.drop();   // (whatever was not taken over goes back to the block cache)
  r.ph = ph.p, r.pc = pc.p, r.pmask = pcap - 1, r.mt = mt.p, r.mmask = mcap - 1;
  r.item_cap = (uint32_t)std::min<size_t>((size_t)(((size_t)(ne * 6) + nb * (size_t)64) * mult[0]) + (size_t)(nb / GPW + 2 + SPARSE_CAP + 8 + BIG_WG * BIG_NW) * ICH + (1u << 20), 0x7FFFFFF0u);
  r.rn_cap = (uint32_t)std::min<size_t>((size_t)(ne * 8 * mult[1]) + (1u << 20), 0x7FFFFFF0u);
  r.req_cap = (uint32_t)std::min<size_t>((size_t)(ne * 1 * mult[2]) + 65536, 0x7FFFFFF0u);
  DevBuf<Item> items;
  DevBuf<RNode> rn;
  DevBuf<pgx_align_key> rq_key;
  DevBuf<pgx_match> rq_res;
  {
    MemTag t4("replay.items");
    items.alloc(r.item_cap);
  }
  {
    MemTag t5("replay.reader_nodes");
    rn.alloc(r.rn_cap);
  }
  {
    MemTag t6("replay.requests");
    rq_key.alloc(r.req_cap), rq_res.alloc(r.req_cap);
  }
  r.items = items.p, r.rn = rn.p, r.rq_key = rq_key.p, r.rq_res = rq_res.p;
  DevBuf<uint8_t> bytes(nb * 5);
  DevBuf<uint32_t> words(nb * 5);
  r.dirty = bytes.p, r.evaluated = bytes.p + nb, r.parity = bytes.p + 2 * nb, r.bflags = bytes.p + 3 * nb, r.ever = bytes.p + 4 * nb;
  r.ihead = words.p, r.inum = words.p + nb, r.ohead = words.p + 2 * nb, r.lookups = words.p + 3 * nb, r.skips = words.p + 4 * nb;
  // big buckets (>= big_min entries, no read twice) are evaluated by a workgroup each: k_eval_big beside every evaluation launch
  // (measured at C4 scale: the buckets that hold a read twice -- 7 k of 2.3 M, up to 128 entries, one partner at a time in the narrow
  // kernels -- were what every sparse pass waited for; big buckets WITHOUT a repeated read are rare (40 of 2.3 M beyond 48 entries:
  // the multiplicity cut-off removes the repeat families' shimmers) and stay with the narrow kernels by default)
  // (round 4: 48 -- with k_eval_big's walk a third shorter the long buckets WITHOUT a repeated read are better off there too: k_eval_rows 38 -> 22 ms
  //  per c4s step, hidden behind k_eval_big as it is: 360 -> 354 ms; from 24 entries on k_eval_big doubles, 411 ms.  Sixteen wavefronts = eight
  //  rows a step (-DPGX_BIG_NW=16): k_eval_big 56 -> 186 ms.)
  r.big_min = getenv("PGX_REPLAY_BIG") ? (uint32_t)std::max(0, atoi(getenv("PGX_REPLAY_BIG"))) : 48u;
  r.dup_min = getenv("PGX_REPLAY_DUP") ? (uint32_t)std::max(0, atoi(getenv("PGX_REPLAY_DUP"))) : 12u;
  DevBuf<uint4> wcur(nb + 2 + SPARSE_CAP + 1 + (size_t)BIG_WG * BIG_NW);  // (one slot per wavefront of k_eval: GPW buckets each; list mode; k_eval_big)
  r.wcur = wcur.p, r.wlist0 = (uint32_t)(nb + 2), r.wbig0 = (uint32_t)(nb + 2 + SPARSE_CAP + 1);
  DevBuf<uint32_t> dlist(LIST_CAP), blist(LIST_CAP + 64);
  r.dlist = dlist.p, r.blist = blist.p;
  DevBuf<Counters> dc(1);
  r.c = dc.p;
  DevBuf<unsigned long long> spread(SPREAD * 8);
  r.spread = spread.p;
  PGX_HIP(hipMemsetAsync(spread.p, 0, SPREAD * 8 * sizeof(unsigned long long), s));
  {
    const int mq = END_FUZZ * 2 - 8, mt = END_FUZZ * 2 - 8;
    r.predict = predict ? std::max(mq, 1) : 0, r.predict2 = mt;
  }
  r.bestn = bestn, r.settled = 0;
  const bool timed_misc = getenv("PGX_REPLAY_TIMING") && atoi(getenv("PGX_REPLAY_TIMING")) != 0;   // "replay_misc" / "replay_emit" in pgx_timing_get
  std::optional<KernelTimer> tm_setup;
  if (timed_misc) tm_setup.emplace("replay_misc", nb);
  if (!precleared) PGX_HIP(hipMemsetAsync(ph.p, 0, (size_t)pcap * sizeof(PHot), s));
  if (!pc_pre) PGX_HIP(hipMemsetAsync(pc.p, 0, ccap * sizeof(PCold), s));
  if (!precleared) PGX_HIP(hipMemsetAsync(mt.p, 0, (size_t)mcap * sizeof(MSlot), s));
  if (trace && precleared) fprintf(stderr, "[pgx]   replay tables taken over pre-cleared (pair table%s, memo)\n", pc_pre ? ", reader lists" : "");
  PGX_HIP(hipMemsetAsync(bytes.p, 0, nb * 5, s));
  PGX_HIP(hipMemsetAsync(words.p, 0, nb * 5 * sizeof(uint32_t), s));
  PGX_HIP(hipMemsetAsync(dc.p, 0, sizeof(Counters), s));
  hipLaunchKernelGGL(k_init_slots, dim3(cdiv256(wcur.n)), dim3(256), 0, s, wcur.p, (uint32_t)wcur.n, (uint32_t)(nb / GPW + 2), r.wlist0, dc.p);
  if (trace) {
    DevBuf<uint32_t> hist(32);
    PGX_HIP(hipMemsetAsync(hist.p, 0, 32 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_setup, dim3(cdiv256(nb)), dim3(256), 0, s, r, hist.p);
    uint32_t h[32];
    hist.download(h, 32);
    sync();
    fprintf(stderr, "[pgx]   buckets by entries / 8 (last class: >= 120):");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", h[i]);
    fprintf(stderr, "\n[pgx]   ... of those holding a read twice (one partner at a time):");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %u", h[16 + i]);
    fprintf(stderr, "\n");
  } else {
    hipLaunchKernelGGL(k_setup, dim3(cdiv256(nb)), dim3(256), 0, s, r, (uint32_t *)nullptr);
  }
  tm_setup.reset();

  bool use_big = r.big_min || r.dup_min;   // (decided below, once k_setup has counted the big buckets)
  static Counters *hc = nullptr;  // pinned mirror of the device counters
  static unsigned long long *hs = nullptr;  // pinned mirror of the spread totals
  static uint32_t *reset3 = nullptr;  // {ndirty, min_dirty, max_dirty} before a count (pinned, constant)
  if (!hc) {
    PGX_HIP(hipHostMalloc((void **)&hc, sizeof(Counters), hipHostMallocDefault));
    PGX_HIP(hipHostMalloc((void **)&hs, SPREAD * 8 * sizeof(unsigned long long), hipHostMallocDefault));
    PGX_HIP(hipHostMalloc((void **)&reset3, 3 * sizeof(uint32_t), hipHostMallocDefault));
    reset3[0] = 0, reset3[1] = 0xFFFFFFFFu, reset3[2] = 0;
  }
  auto fetch = [&](bool totals) {  // counters (and, for the trace / the statistics, the spread totals) to the host; synchronises
    PGX_HIP(hipMemcpyAsync(hc, dc.p, sizeof(Counters), hipMemcpyDeviceToHost, s));
    if (totals) PGX_HIP(hipMemcpyAsync(hs, spread.p, SPREAD * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    sync();
    if (totals) {
      hc->evals = hc->lookups = hc->skips = 0;
      hc->records = 0;   // (re-used as the stream checksum k_emit adds up in word 7 of every line)
      for (uint32_t i = 0; i < SPREAD; ++i) hc->evals += hs[i * 8], hc->lookups += hs[i * 8 + 1], hc->skips += hs[i * 8 + 2], hc->records += hs[i * 8 + 7];
#ifdef PGX_BIG_STATS
      {
        unsigned long long b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
        for (uint32_t i = 0; i < SPREAD; ++i) b3 += hs[i * 8 + 3], b4 += hs[i * 8 + 4], b5 += hs[i * 8 + 5], b6 += hs[i * 8 + 6], b7 += hs[i * 8 + 7];
        fprintf(stderr, "[pgx]   k_eval_big so far: %u evaluations, %llu steps, %llu rows committed; steps cut at a duplicate %llu, ended by a containment %llu, all rows %llu; longest %u steps, %u evaluations of >= 64 steps (entries %u)\n",
                hc->big_evals, b3, b4, b5, b6, b7, hc->big_max_steps, hc->big_long, hc->big_long_n);
      }
#endif
#ifdef PGX_SETTLE_STATS
      unsigned long long c3 = 0, c4 = 0, c5 = 0, c6 = 0;
      for (uint32_t i = 0; i < SPREAD; ++i) c3 += hs[i * 8 + 3], c4 += hs[i * 8 + 4], c5 += hs[i * 8 + 5], c6 += hs[i * 8 + 6];
      fprintf(stderr, "[pgx]   wrong guesses so far: type %llu, rejected short overlap %llu, rejected no match %llu, rejected other %llu\n", c3, c4, c5, c6);
#endif
    }
  };
  if (use_big) {   // no big bucket in this set (uniform-random genomes): none of the ~100 passes of a step launches k_eval_big
    fetch(false);
    use_big = hc->nbig_total != 0;
  }
  const uint32_t nblk = (uint32_t)((nb + CB - 1) / CB);
  DevBuf<uint32_t> cblk((size_t)nblk * 3);
  auto launch_count = [&](uint32_t lo = 0, uint32_t hi = 0xFFFFFFFFu) {  // count + list of the dirty buckets of [lo, hi)
    hi = std::min<uint32_t>(hi, (uint32_t)nb);
    const uint32_t nbl = std::max<uint32_t>(1, (hi - lo + CB - 1) / CB);
    std::optional<KernelTimer> tmc;
    if (timed_misc) tmc.emplace("replay_misc", 0);
    hipLaunchKernelGGL(k_count_a, dim3(nbl), dim3(256), 0, s, r, lo, hi, cblk.p);
    hipLaunchKernelGGL(k_count_b, dim3(nbl), dim3(256), 0, s, r, lo, hi, cblk.p, nbl);
  };
  auto read_counters = [&](bool count_dirty) {
    if (count_dirty) {  // reset the three dirty statistics, keep the rest
      launch_count();
    }
    fetch(true);
    return hc->overflow == 0;
  };
  const size_t window = getenv("PGX_REPLAY_WIN") ? (size_t)atoll(getenv("PGX_REPLAY_WIN")) : (size_t)262144;
  // The schedule's constants.  Every one of them was an environment knob through round 3; tools/knob_sweep.sh (profiles/r04f_knob_sweep_c4s.txt)
  // moved each over its plausible range on the repeat-rich 9-Gbase set: 391-405 ms per step whatever the setting (only k_eval_big behind
  // instead of beside the narrow kernel is worse, 417 ms), so they are constants now; PGX_REPLAY_WIN / _K stay because the parity
  // tests randomise the schedule with them.
  const size_t win0 = 16384, win1 = 131072;   // the first pass ramps its window from win0 up to win1
  // dense rounds only while more than 1 / dense_den of the buckets is dirty.  (Through round 4 also from 65,536 dirty buckets on: the second sweep of a
  // full-size c4 chunk -- 275 k wrong guesses among 2.7 M buckets -- then went window by window, 41 passes each as long as its longest big bucket, 52 ms;
  // as sparse passes from the list: 7.06 -> 6.97 s per step, + 0.3 % evaluations.)
  const size_t dense_min = (size_t)1 << 40;
  const size_t tail_max = 4000;               // tail mode (file_for_reader, look-ahead) once a sweep asks for at most this many alignments, or 1/256 of the first sweep's
  const uint32_t ahead = 24u;                 // tail mode: partners of a row filed ahead
  // (round 6, tried and removed: k_settle filing, right where it finds a REJECTED guess, the alignment every other reader of that pair would ask
  //  for and the row's next 0 / 2 / 6 partners, aligned before the next sweep -- bit-exact, and no sweep fewer: 18-19 sweeps and 6.21-6.33 s per
  //  full-size c4 step against 18 and 6.13 s without, profiles/r06b_settle_fan_c4.txt.  The sweeps behind the second are not hand-overs of one
  //  pair between its readers -- tail mode's file_for_reader already covers those -- but a dependency chain through DIFFERENT pairs.)
  const bool use_win_list = true;
  // evaluate / update iterations per window of a dense round: 2 (through round 4: 3; at full-size c4 6.89 -> 6.81 s per step with 2 and 6.90 with 1,
  // c3 / c4s / c5s unchanged; the third iteration of a window mostly re-runs k_eval_big's longest bucket for a handful of dirty buckets that the
  // sparse passes behind the round pick up anyway)
  const int inner = getenv("PGX_REPLAY_K") ? atoi(getenv("PGX_REPLAY_K")) : 2;
  const bool wide = true;          // sparse passes: a wavefront per bucket, four rows per step
  const size_t dense_den = 3;      // dense rounds while more than 1/dense_den of the buckets is dirty
  const bool wide_dense = false;   // (the dense rounds keep 16 lanes per bucket: measured)
  // sparse passes per host round trip.  Round 2 measured 2 .. 8 within 1 % of each other and kept 4; with the GPU no longer waiting for the
  // host elsewhere (round 3) the round trips and the k_file pass that precedes each one show: 8 instead of 4 = replay kernels 38.0 -> 35.5 ms
  // and the step 113.4 -> 111.2 ms at c3, c4s 437 -> 429 ms, c5s 651 -> 639 (6), the E. coli-size set unchanged
  const bool big_side = true;   // k_eval_big beside the narrow kernel of a pass, on a second stream
  static hipStream_t side_stream = nullptr;
  static hipEvent_t side_ev[2] = {nullptr, nullptr};
  if (!side_stream) {
    PGX_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    for (auto &e : side_ev) PGX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    static ShutdownHook h_([] {   // (a later pgx_init may choose another device)
      if (side_stream) (void)hipStreamDestroy(side_stream), side_stream = nullptr;
      for (auto &e : side_ev)
        if (e) (void)hipEventDestroy(e), e = nullptr;
    });
  }
  const int big_every = 1;   // sparse passes per k_eval_big launch
  const int chain = 8;       // sparse passes per host round trip
  static const bool deep = getenv("PGX_TRACE") && atoi(getenv("PGX_TRACE")) >= 2;  // per-kernel wall times (synchronises after every launch)
  const bool timed = getenv("PGX_REPLAY_TIMING") && atoi(getenv("PGX_REPLAY_TIMING")) != 0;  // "replay" in pgx_timing_get
  double td = 0, t_eval = 0, t_upd = 0;
  size_t first_req = 0;  // requests [first_req, ...) belong to the running sweep
  unsigned sweeps = 0, rounds_total = 0;
  uint32_t d_lo = 0, d_hi = (uint32_t)nb;  // range holding the dirty buckets
  size_t n_dirty = nb;                     // as of the last fetch
  bool have_list = false;                  // the device holds a dirty list (a count has run since the last evaluations)
  bool known = true;                       // n_dirty / d_lo / d_hi are current
  double align_ms = 0;
  size_t first_batch = 0;   // alignments the first sweep asked for
  auto count_dirty = [&](bool reset = false) {  // no host round trip: ndirty, the range and the list stay on the device
    (void)reset;
    launch_count();
    have_list = true;
  };
  for (;;) {
    ++sweeps;
    const double p0 = now_ms();
    unsigned rounds = 0;
    for (;;) {  // until no bucket is dirty
      ++rounds;
      if (known && (n_dirty > dense_min || n_dirty * dense_den > nb)) {  // many of the buckets: a group of 16 lanes each, in bucket order
        // dense: window by window, in order (a window's buckets mostly depend on earlier windows)
        const size_t a_lo = d_lo & ~(size_t)63;  // (aligned: a bucket always belongs to the same wavefront slot)
        const size_t win = n_dirty > window / 4 ? window : (size_t)(d_hi - a_lo);
        // the very first pass ramps the window up (win0, 2 win0, ... window): with an empty pair table every bucket of a window
        // believes it owns all its pairs, and the early windows -- where nothing is "seen" yet -- re-evaluate nearly all of their
        // buckets; smaller windows there mean fewer wasted evaluations (2.20 M -> 1.7 M at 4.5 Gbases), later ones are launch-bound
        const bool first_pass = sweeps == 1 && rounds == 1;
        const size_t cap = first_pass ? std::min(win1, win) : win;
        size_t step = first_pass ? std::min(win0, cap) : win;
        for (size_t lo = a_lo, nxt; lo < d_hi; lo = nxt, step = std::min(step * 2, cap)) {
          nxt = lo + step;
          const uint32_t hi = (uint32_t)std::min<size_t>(d_hi, nxt);
          for (int k = 0; k < inner; ++k) {
            std::optional<KernelTimer> tm;  // PGX_REPLAY_TIMING=1: "replay_dense" = k_eval, "replay_rows" = k_eval_rows, "replay_update" = k_update
            if (timed) tm.emplace(wide_dense ? "replay_rows" : "replay_dense", k == 0 ? hi - lo : 0);  // (units: buckets of the window, once)
            if (deep) sync(), td = now_ms();
            // all but the very first evaluation of a window find only a part of its buckets dirty: they run from the window's
            // list (wavefronts full of live buckets) instead of over every bucket of the window
            const bool from_list = use_win_list && !(first_pass && k == 0) && !wide_dense && hi - lo <= LIST_CAP;
            if (from_list) launch_count((uint32_t)lo, hi);
            const bool side = use_big && from_list && big_side && !timed && !deep;   // (the window's big buckets are in the count's list: beside k_eval)
            if (side) {
              PGX_HIP(hipEventRecord(side_ev[0], s));
              PGX_HIP(hipStreamWaitEvent(side_stream, side_ev[0], 0));
              hipLaunchKernelGGL(k_eval_big, dim3((unsigned)std::min<size_t>(BIG_WG, hi - lo)), dim3(64 * BIG_NW), 0, side_stream, r, (uint32_t)LIST_CAP,
                                 (uint32_t)nb, DEV_LIST_WIN);
              PGX_HIP(hipEventRecord(side_ev[1], side_stream));
            }
            if (wide_dense) hipLaunchKernelGGL((k_eval_rows<64, 16>), dim3(cdiv256((size_t)(hi - lo) * 64)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            else if (from_list) hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST_WIN);
            else hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            tm.reset();
            if (side) {
              PGX_HIP(hipStreamWaitEvent(s, side_ev[1], 0));
            } else if (use_big) {
              if (timed) tm.emplace("replay_big", 0);
              const unsigned wgs = (unsigned)std::min<size_t>(BIG_WG, hi - lo);
              if (from_list) hipLaunchKernelGGL(k_eval_big, dim3(wgs), dim3(64 * BIG_NW), 0, s, r, (uint32_t)LIST_CAP, (uint32_t)nb, DEV_LIST_WIN);
              else hipLaunchKernelGGL(k_eval_big, dim3(wgs), dim3(64 * BIG_NW), 0, s, r, (uint32_t)lo, hi, 0u);
              tm.reset();
            }
            if (deep) sync(), t_eval += now_ms() - td, td = now_ms();
            if (timed) tm.emplace("replay_update", 0);
            if (from_list) hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST_WIN);
            else hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)(hi - lo) * GL)), dim3(256), 0, s, r, (uint32_t)lo, hi, 0u);
            tm.reset();
            if (deep) sync(), t_upd += now_ms() - td, fprintf(stderr, "[pgx]     iteration: eval %.3f ms, update %.3f ms\n", t_eval, t_upd), t_eval = t_upd = 0;
          }
        }
        count_dirty();
      }
      // sparse: `chain` passes straight from the list the last count left on the device -- the kernels read its length there,
      // so the host is not in the loop (a pass with nothing to do costs two empty launches); what a pass dirties is listed by
      // the count behind it
      if (!have_list) count_dirty(true);
      {
        const size_t est = known && n_dirty <= SPARSE_CAP / 4 ? std::max<size_t>(n_dirty * 4, 1024) : (size_t)SPARSE_CAP;
        const unsigned groups = (unsigned)std::min<size_t>(est, SPARSE_CAP);
        for (int c = 0; c < chain; ++c) {
          std::optional<KernelTimer> tm;
          const bool big_now = use_big && (c % big_every == big_every - 1 || c == chain - 1);
          // the big buckets of the pass (listed by the count, k_count_b) beside the narrow kernel, on a second stream: a launch of
          // k_eval_big lasts as long as its longest bucket (~0.4 ms of dependent probes), whatever else the GPU could be doing
          const bool side = big_now && big_side && !timed && !deep;
          if (side) {
            PGX_HIP(hipEventRecord(side_ev[0], s));
            PGX_HIP(hipStreamWaitEvent(side_stream, side_ev[0], 0));
            hipLaunchKernelGGL(k_eval_big, dim3(std::min<unsigned>(BIG_WG, groups)), dim3(64 * BIG_NW), 0, side_stream, r, groups, (uint32_t)nb, DEV_LIST);
            PGX_HIP(hipEventRecord(side_ev[1], side_stream));
          }
          if (timed) tm.emplace(wide ? "replay_rows" : "replay_dense", 0);
          if (wide) hipLaunchKernelGGL((k_eval_rows<64, 16>), dim3(cdiv256((size_t)groups * 64)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          else hipLaunchKernelGGL(k_eval, dim3(cdiv256((size_t)groups * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          tm.reset();
          if (side) {
            PGX_HIP(hipStreamWaitEvent(s, side_ev[1], 0));
          } else if (big_now) {
            if (timed) tm.emplace("replay_big", 0);
            hipLaunchKernelGGL(k_eval_big, dim3(std::min<unsigned>(BIG_WG, groups)), dim3(64 * BIG_NW), 0, s, r, groups, (uint32_t)nb, DEV_LIST);
            tm.reset();
          }
          if (timed) tm.emplace("replay_update", 0);
          hipLaunchKernelGGL(k_update, dim3(cdiv256((size_t)groups * GL)), dim3(256), 0, s, r, 0u, (uint32_t)nb, DEV_LIST);
          tm.reset();
          count_dirty();
        }
      }
      // file what the clean buckets need (always safe), then one round trip for everything: dirty count, range, requests
      {
        std::optional<KernelTimer> tmf;
        if (timed_misc) tmf.emplace("replay_misc", 0);
        hipLaunchKernelGGL(k_file, dim3(cdiv256(nb)), dim3(256), 0, s, r, (uint32_t)nb);
      }
      fetch(trace);
      if (hc->overflow) goto overflowed;
      n_dirty = hc->ndirty, known = true;
      r.memo_used = hc->nreq != 0;
      d_lo = n_dirty ? hc->min_dirty : 0, d_hi = n_dirty ? hc->max_dirty + 1 : 0;
      if (trace)
        fprintf(stderr, "[pgx]   round %u: %zu dirty left in [%u, %u), %llu evaluations, t = +%.2f ms\n", rounds, n_dirty, d_lo, d_hi,
                (unsigned long long)hc->evals, now_ms() - t0);
      if (!n_dirty) break;
      if (rounds > 20000) {  // (cannot happen: the lowest unstable bucket rises every pass)
        hc->overflow |= OV_PASSES;
        goto overflowed;
      }
    }
    rounds_total += rounds;
    const size_t nreq = hc->nreq;
    if (trace)
      fprintf(stderr, "[pgx] device sweep %u: %u rounds, %llu evaluations so far, %.2f ms, %zu requests\n", sweeps, rounds,
              (unsigned long long)hc->evals, now_ms() - p0, nreq - first_req);
    if (nreq == first_req) break;
    const double a0 = now_ms();
    const size_t batch = nreq - first_req;
    if (sweeps == 1) first_batch = batch;
    r.tail = tail_max && batch <= std::max(tail_max, first_batch / 256) ? ahead : 0u;   // (the NEXT sweep's k_file)
    dev_align(db, r.rq_key + first_req, batch, band, r.rq_res + first_req, sweeps > 2 ? 2 : sweeps > 1 ? 1 : 0);
    r.settled = (uint32_t)nreq;
    first_req = nreq;
    {
      std::optional<KernelTimer> tms;
      if (timed_misc) tms.emplace("replay_misc", 0);
      hipLaunchKernelGGL(k_settle, dim3(cdiv256(nb)), dim3(256), 0, s, r);
    }
    count_dirty();
    if (batch > 100000) {  // a big batch: worth a round trip to know how many guesses were wrong (dense or sparse next)
      fetch(false);
      if (hc->overflow) goto overflowed;
      n_dirty = hc->ndirty, known = true;
      d_lo = n_dirty ? hc->min_dirty : 0, d_hi = n_dirty ? hc->max_dirty + 1 : 0;
      if (trace) fprintf(stderr, "[pgx]   alignments + settle %.2f ms, %zu buckets guessed wrong\n", now_ms() - a0, n_dirty);
      if (!n_dirty) break;
    } else {
      known = false;  // (a small batch: the wrong guesses are handled by the sparse passes of the next round)
      n_dirty = std::min<size_t>(batch, SPARSE_CAP / 8);
      if (trace) fprintf(stderr, "[pgx]   %zu alignments + settle enqueued in %.2f ms\n", batch, now_ms() - a0);
    }
    align_ms += now_ms() - a0;
  }
  {
    const double e0 = now_ms();
    DevBuf<uint32_t> off(nb + 1);
    size_t tb = 0;
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, r.inum, off.p, (int)nb, s));
    DevBuf<uint8_t> tmp(tb + 256);
    PGX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, r.inum, off.p, (int)nb, s));
    uint32_t last_off = 0, last_num = 0;
    PGX_HIP(hipMemcpyAsync(&last_off, off.p + nb - 1, 4, hipMemcpyDeviceToHost, s));
    PGX_HIP(hipMemcpyAsync(&last_num, r.inum + nb - 1, 4, hipMemcpyDeviceToHost, s));
    sync();
    const size_t nrec = (size_t)last_off + last_num;
    // (the table census for the next stage's sizes runs HERE, ahead of the record copy: a memory-bound kernel beside the copy's blit kernel
    //  crawls -- 40 ms instead of 0.5 for these 2 GB, profiles/r05e_chunk_timeline_c4.txt)
    DevBuf<unsigned long long> d_keys(1);
    PGX_HIP(hipMemsetAsync(d_keys.p, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_count_pairs, dim3((unsigned)std::min<size_t>(cdiv256(pcap), 4096)), dim3(256), 0, s, ph.p, pcap, d_keys.p);
    results_wait();   // (pgx_results_async: the previous stage's record copy -- long finished -- gives its device buffer back first)
    pgx_ovlp *host = alloc_out(nrec);
    DevBuf<pgx_ovlp> d_out;
    {
      MemTag t7("replay.records_out");
      d_out.alloc(std::max<size_t>(nrec, 1));
    }
    {
      std::optional<KernelTimer> tme;
      if (timed_misc) tme.emplace("replay_emit", nrec);   // the records written and brought to the host (pinned destination)
      hipLaunchKernelGGL(k_emit, dim3(cdiv256(nb)), dim3(256), 0, s, r, off.p, d_out.p);
      // pgx_results_async: the copy runs on its own stream behind k_emit and this call returns without it -- the 2.9 GB of a human-scale
      // chunk (55 ms over PCIe) overlap the NEXT chunk's join and first sweep; the caller waits (pgx_results_wait) before it reads
      if (nrec && record_sink()) record_sink()->take(std::move(d_out), nrec);   // (served commands: device -> output file, pgx_served.cpp)
      else if (nrec && results_async() && !timed_misc) results_copy_async(host, std::move(d_out), nrec);
      else if (nrec) PGX_HIP(hipMemcpyAsync(host, d_out.p, nrec * sizeof(pgx_ovlp), hipMemcpyDeviceToHost, s));
    }
    unsigned long long n_keys = 0;
    d_keys.download(&n_keys, 1);
    if (!read_counters(false)) goto overflowed;
    // what this stage used, per bucket entry: read pairs, requests (= memo entries), items, reader nodes
    usage[0] = (double)n_keys / ne, usage[1] = (double)hc->nreq / ne, usage[2] = (double)hc->item_top / ne, usage[3] = (double)hc->rnode_top / ne;
    if (trace)
      fprintf(stderr, "[pgx]   tables: %llu read pairs in %u slots (load %.2f), %u alignments in %u memo slots (load %.2f), items %u of %u, reader nodes %u of %u, requests %u of %u\n",
              n_keys, pcap, (double)n_keys / pcap, hc->nreq, mcap, (double)hc->nreq / mcap, hc->item_top, r.item_cap, hc->rnode_top, r.rn_cap, hc->nreq, r.req_cap);
    g_last_pcap = pcap, g_last_mcap = mcap, g_last_ccap = ccap;
    *n_out = nrec;
    if (st) {
      st->n_align_needed = hc->lookups, st->n_seen_skip = hc->skips, st->n_align_gpu = first_req;
      st->rounds = sweeps;
      st->n_evaluations = hc->evals;
#if !defined(PGX_BIG_STATS) && !defined(PGX_SETTLE_STATS)
      st->stream_checksum = hc->records;
#endif
    }
    if (trace)
      fprintf(stderr, "[pgx] device replay: %u sweeps, %u rounds, %llu evaluations, %zu records; emit %.2f ms; alignments %.2f ms; total %.2f ms\n",
              sweeps, rounds_total, (unsigned long long)hc->evals, nrec, now_ms() - e0, align_ms, now_ms() - t0);
  }
  return 0;
overflowed:
  if (trace || !(hc->overflow & (OV_ITEMS | OV_NODES | OV_REQS | OV_PAIRS | OV_MEMO)))
    fprintf(stderr, "[pgx] note: the device replay gave up (code %u: items %u of %u, reader nodes %u of %u, requests %u of %u)\n", hc->overflow,
            hc->item_top, r.item_cap, hc->rnode_top, r.rn_cap, hc->nreq, r.req_cap);
  return hc->overflow ? hc->overflow : OV_PASSES;
}
}  // namespace

// the visit list on the device from the per-group slices the host's table replay left (pgx_overlap.cpp::build_visit, ids-only form)
namespace {
__global__ void k_place_bids(const uint32_t *__restrict__ ids_all, const uint32_t *__restrict__ psrc, const uint32_t *__restrict__ pcnt,
                             const uint64_t *__restrict__ pdst, uint32_t n_groups, uint32_t *__restrict__ bid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_groups) return;
  const uint32_t src = psrc[i], n = pcnt[i];
  const uint64_t dst = pdst[i];
  for (uint32_t k = 0; k < n; ++k) bid[dst + k] = ids_all[src + k];
}
}  // namespace
void dev_place_bids(const uint32_t *ids_all, size_t n_ids, const uint32_t *psrc, const uint32_t *pcnt, const uint64_t *pdst,
                    size_t n_groups, size_t nb, DevBuf<uint32_t> &bid) {
  bid.alloc(std::max<size_t>(nb, 1));
  if (!n_groups || !nb) return;
  DevBuf<uint32_t> d_ids(n_ids), d_src(n_groups), d_cnt(n_groups);
  DevBuf<uint64_t> d_dst(n_groups);
  d_ids.upload(ids_all, n_ids), d_src.upload(psrc, n_groups), d_cnt.upload(pcnt, n_groups), d_dst.upload(pdst, n_groups);
  hipLaunchKernelGGL(k_place_bids, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, ctx().stream, d_ids.p, d_src.p, d_cnt.p, d_dst.p,
                     (uint32_t)n_groups, bid.p);
  sync();   // (the upload sources are the caller's host arrays; the temporaries go back to the block cache)
}

namespace {
double g_learned[4] = {0, 0, 0, 0};
ShutdownHook h_learn([] { replay_forget_sizes(); });
}  // namespace
void replay_forget_sizes() {
  for (double &m : g_learned) m = 0;
  g_pre.drop();
  g_last_pcap = g_last_mcap = 0, g_last_ccap = 0;
}

void replay_drop_precleared() { g_pre.drop(); }   // (a stage that ended without a device replay: nothing cleared ahead of time outlives it)

// called by the stage's front while the GPU would otherwise wait for the host's outer table: tables of the last stage's sizes, cleared, on ctx().stream
void replay_preclear() {
  if (!g_last_pcap || g_pre.valid || (getenv("PGX_REPLAY_PRECLEAR") && atoi(getenv("PGX_REPLAY_PRECLEAR")) == 0)) return;
  size_t free_b = 0, total_b = 0;
  const size_t need = (size_t)g_last_pcap * sizeof(PHot) + g_last_ccap * sizeof(PCold) + (size_t)g_last_mcap * sizeof(MSlot);
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b + dev_cache_free_bytes()) return;   // (no room ahead of time: the attempt decides)
  try {
    hipStream_t s = ctx().stream;
    {
      MemTag t1("replay.pair_table_hot");
      g_pre.ph.alloc(g_last_pcap);
    }
    {
      MemTag t2("replay.pair_table_readers");
      g_pre.pc.alloc(g_last_ccap);
    }
    {
      MemTag t3("replay.memo_table");
      g_pre.mt.alloc(g_last_mcap);
    }
    PGX_HIP(hipMemsetAsync(g_pre.ph.p, 0, (size_t)g_last_pcap * sizeof(PHot), s));
    PGX_HIP(hipMemsetAsync(g_pre.pc.p, 0, g_last_ccap * sizeof(PCold), s));
    PGX_HIP(hipMemsetAsync(g_pre.mt.p, 0, (size_t)g_last_mcap * sizeof(MSlot), s));
    g_pre.pcap = g_last_pcap, g_pre.mcap = g_last_mcap, g_pre.ccap = g_last_ccap, g_pre.valid = true;
  } catch (const Fail &) {
    (void)hipGetLastError();
    g_pre.drop();
  }
}

bool dev_replay(const pgx_seqdb *db, const DevicePairs &dp, const uint32_t *visit_bids, const uint32_t *d_bids, size_t nb, size_t n_entries,
                uint32_t bestn, int band, bool predict, uint32_t ovlp_upper, const std::function<pgx_ovlp *(size_t)> &alloc_out,
                size_t *n_out, pgx_overlap_stats *st, bool trace) {
  *n_out = 0;
  struct DropPre {   // (tables cleared ahead of time that no attempt took over do not outlive the stage)
    ~DropPre() { g_pre.drop(); }
  } drop_pre;
  // what the encodings hold (anything else goes to the host replay)
  if (ovlp_upper > 128 || nb >= (1u << 29) - 2 || n_entries >= (1ULL << 31) || !dp.valid) return false;
  if (nb == 0) {
    alloc_out(0);
    return true;
  }
  {  // the device tables must fit beside what is already resident (otherwise the host replay, which only needs host memory)
    const size_t ne = std::max<size_t>(n_entries, 1024);
    const size_t need = (size_t)pow2_at_least(ne) * (sizeof(PHot) + sizeof(PCold) + sizeof(MSlot)) + ne * (6 * sizeof(Item) + 8 * sizeof(RNode) + sizeof(pgx_align_key) + sizeof(pgx_match)) +
                        nb * (size_t)(64 * sizeof(Item) + 64) + (256u << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b) {
      g_pre.drop();      // (tables cleared ahead of time are part of what `need` prices)
      dev_cache_trim();  // (blocks the cache holds for re-use count as used)
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || need > free_b) {
        fprintf(stderr, "[pgx] note: the device replay's tables (%.1f GB) do not fit the free device memory (%.1f GB); the host replay takes over\n",
                need / 1e9, free_b / 1e9);
        return false;
      }
    }
  }
  // Table sizes are multiples of the defaults (items 6 x, reader nodes 8 x, requests 1 x the bucket entries; pair and memo table: the power of
  // two from the entries up).  Two things were found at full-size configs[3] (round 5, profiles/r05a_chunk_timeline_c4.txt):
  //  * every one of the 8 chunks of a step ran two attempts in vain -- requests 1.72 x the entries, then the memo table -- and an
  //    overflowing attempt costs a whole first sweep (the request / memo overflow is only seen when k_file runs): 205 of 1,103 ms;
  //  * the open-addressing tables want a LOW load: with the pair table at 0.75 (64 M slots for 50 M read pairs) and the memo table at 0.71
  //    the evaluations of a chunk took 73 ms longer than at 0.37 / 0.36 (linear probing: ~8 probes per miss instead of ~2).
  // So a stage measures what it used per bucket entry -- read pairs, requests, items, reader nodes -- and the NEXT stage of the process (the
  // chunks of a job are alike) sizes its tables by that: hash tables for a load of at most 0.4, arenas with 20 % to spare.  The first stage of
  // a process starts from the defaults and, where they overflow, repeats with x 4 hash tables / x 2 arenas.
  double *learned = g_learned;   // read pairs, requests, items, reader nodes per bucket entry (0: not known)
  double mult[5] = {1, 1, 1, 1, 1};
  if (learned[0] > 0) {
    // (the arenas may also SHRINK to what the last stage used + 25-100 %: 8 reader nodes per entry are reserved by default and a c4 chunk links
    //  25 thousand of its 213 million; a stage that needs more than that repeats once and the next one knows)
    const double items_default = 6.0 + 64.0 * (double)nb / std::max<size_t>(n_entries, 1024);
    mult[0] = std::max(0.2, learned[2] * 1.25 / items_default), mult[1] = std::max(0.02, learned[3] * 2 / 8), mult[2] = std::max(1.0, learned[1] * 1.2);
    mult[3] = std::max(1.0, learned[0] / 0.4), mult[4] = std::max(1.0, learned[1] / 0.4);
  }
  if (getenv("PGX_REPLAY_PAIRS_X")) mult[3] = atof(getenv("PGX_REPLAY_PAIRS_X"));
  if (getenv("PGX_REPLAY_MEMO_X")) mult[4] = atof(getenv("PGX_REPLAY_MEMO_X"));
  for (int attempt = 0; attempt < 4; ++attempt) {
    double usage[4] = {0, 0, 0, 0};
    uint32_t ov;
    try {
      ov = replay_attempt(db, dp, visit_bids, d_bids, nb, n_entries, bestn, band, predict, alloc_out, n_out, st, trace, mult, usage);
    } catch (const Fail &f) {
      //  tables sized by what an EARLIER stage used -- possibly another job's in a long-lived server -- may not fit where the
      // defaults would: once more with the defaults and the memory forgotten, else the host replay (which only needs host memory)
      if (f.code != PGX_ENOMEM && !(f.code == PGX_EHIP && strstr(pgx_last_error(), "hipMalloc"))) throw;   // (only a failed allocation)
      (void)hipGetLastError();
      const bool had_learned = learned[0] > 0;
      replay_forget_sizes();
      dev_cache_trim();
      if (!had_learned || attempt > 0) {
        fprintf(stderr, "[pgx] note: the device replay's tables could not be allocated (%s); the host replay takes over\n", pgx_last_error());
        return false;
      }
      for (double &m : mult) m = 1;
      if (trace) fprintf(stderr, "[pgx]   the tables sized by the last stage's use do not fit (%s): once more with the default sizes\n", pgx_last_error());
      continue;
    }
    if (!ov) {
      for (int k = 0; k < 4; ++k) learned[k] = std::max(learned[k], usage[k]);
      if (st) st->replay_attempts = (uint32_t)attempt + 1;
      return true;
    }
    if (ov & (OV_QOFF | OV_PASSES)) break;  // not a matter of table sizes
    // unusual data (repeat-rich sets): the same walk again with larger tables.  Requests and memo entries are the same alignments: when the
    // request array was too small the memo table of the same size class is too, and k_file stopped before it could say so -- grow both
    for (int k = 0; k < 3; ++k)
      if (ov & (1u << k)) mult[k] = mult[k] < 1 ? 1.0 : mult[k] * 2;   // (an arena that was shrunk to the last stage's use goes back to the default first)
    if (ov & OV_PAIRS) mult[3] *= 4;
    if (ov & (OV_MEMO | OV_REQS)) mult[4] *= 4;
    if (trace) fprintf(stderr, "[pgx]   next attempt with items x %g, reader nodes x %g, requests x %g, pair table x %g, memo table x %g\n", mult[0], mult[1], mult[2], mult[3], mult[4]);
  }
  fprintf(stderr, "[pgx] note: the device replay's tables overflowed; the host replay takes over\n");
  return false;
}

}  // namespace pgx
