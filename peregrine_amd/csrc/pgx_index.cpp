// pgx_index.cpp -- the index stage: what main() of /root/reference/src/shmr_index.c:37-245 does for one chunk,
// with sketch / reduce / count on the GPU.  Host work here is read selection, file naming and file IO only.
#include <chrono>

#include "pgx_internal.h"

using namespace pgx;

namespace {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

void check_params(const pgx_index_params *p) {
  PGX_REQUIRE(p, PGX_EARG, "null params");
  // the reference asserts these (shmr_index.c:111-114)
  PGX_REQUIRE(p->total_chunk > 0, PGX_EARG, "total_chunk must be > 0");
  PGX_REQUIRE(p->mychunk > 0 && p->mychunk <= p->total_chunk, PGX_EARG, "mychunk must be in 1..total_chunk");
  PGX_REQUIRE(p->reduction > 0 && p->reduction < 256, PGX_EARG, "reduction factor must be 1..255");
  PGX_REQUIRE(p->window >= 24 && p->kmer >= 12 && p->window > p->kmer, PGX_EARG, "need w >= 24, k >= 12, w > k");
  PGX_REQUIRE(p->window < 256 && p->kmer <= 28, PGX_EARG, "need w < 256, k <= 28 (src/mm_sketch.c:77-78)");
}

template <typename T>
T *download_list(const DevBuf<T> &d, size_t n) {
  T *h = (T *)out_alloc(n ? n * sizeof(T) : 1);
  d.download(h, n);
  return h;
}

void run_index(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *out, DeviceIndex *keep = nullptr, bool host_arrays = true) {
  ++index_generation();   // every index-stage call rewrites the "ix.*" workspaces: views of them handed out earlier are stale now
  memset(out, 0, sizeof(*out));
  if (keep) keep->valid = false;
  const double t0 = now_ms();
  // read selection: rid % total == mychunk % total, in idx-file order (shmr_index.c:155-157); kept with the seqdb for the next call
  static uint64_t next_serial = 0, use_clock = 0;
  pgx_seqdb::IndexPlan *found = nullptr;
  for (auto &pl : db->plans)
    if (pl.total == p->total_chunk && pl.chunk == p->mychunk) found = &pl;
  if (!found) {
    if (db->plans.size() < 32) {
      db->plans.emplace_back();
      found = &db->plans.back();
    } else {
      found = &db->plans[0];
      for (auto &pl : db->plans)
        if (pl.last_use < found->last_use) found = &pl;
    }
    pgx_seqdb::IndexPlan &plan = *found;
    plan.reads.clear();
    plan.bases = 0;
    const uint32_t T = (uint32_t)p->total_chunk, c = (uint32_t)p->mychunk % T;
    for (size_t i = 0; i < db->rid.size(); ++i) {
      if (db->rid[i] % T != c) continue;
      PGX_REQUIRE(db->rlen[i] > 0, PGX_EARG, "read %u is empty (mm_sketch asserts len > 0)", db->rid[i]);
      plan.reads.push_back(ReadDesc{db->roff[i], db->rlen[i], db->rid[i]});
      plan.bases += db->rlen[i];
    }
    plan.total = p->total_chunk, plan.chunk = p->mychunk, plan.serial = ++next_serial;
  }
  pgx_seqdb::IndexPlan &plan = *found;
  plan.last_use = ++use_clock;
  const std::vector<ReadDesc> &reads = plan.reads;
  out->bases = plan.bases;
  out->reads = (uint32_t)reads.size();
  const int kbits = 2 * p->kmer;
  const bool trace = getenv("PGX_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[pgx] index: read selection %.2f ms\n", now_ms() - t0);

  // fast path: everything of the chunk goes through the wave kernel and the in-LDS reduce (pg_run.py's defaults)
  if (!p->want_l0) {
    const pgx_mm128 *d_top = nullptr;
    size_t ntop = 0;
    if (dev_index_fused(db, reads, p->window, p->kmer, p->reduction, p->levels, &d_top, &ntop, plan.serial, &out->reads_literal)) {
      PGX_REQUIRE(ntop < (1ULL << 31), PGX_EARG, "chunk too large (use more index chunks)");
      DevBuf<pgx_mm_count> mc;
      size_t nmc = 0;
      dev_count(d_top, ntop, kbits, mc, nmc);
      out->n_top = ntop, out->n_top_mc = nmc;
      if (host_arrays) {
        out->top = (pgx_mm128 *)out_alloc(ntop ? ntop * sizeof(pgx_mm128) : 1);
        if (ntop) PGX_HIP(hipMemcpyAsync(out->top, d_top, ntop * sizeof(pgx_mm128), hipMemcpyDeviceToHost, ctx().stream));
        out->top_mc = download_list(mc, nmc);
      }
      sync();
      if (trace) fprintf(stderr, "[pgx] index: counts (+ downloads) done at +%.2f ms\n", now_ms() - t0);
      if (keep) keep->d_top = d_top, keep->n_top = ntop, keep->mc = std::move(mc), keep->n_mc = nmc, keep->valid = true;
      timing_flush();
      out->gpu_ms = now_ms() - t0;
      if (trace) fprintf(stderr, "[pgx] index: stage total %.2f ms\n", out->gpu_ms);
      return;
    }
  }
  DevBuf<pgx_mm128> l0, l1, l2;
  size_t n0 = 0, n1 = 0, n2 = 0;
  dev_sketch(db, reads, p->window, p->kmer, l0, n0, &out->reads_literal);
  PGX_REQUIRE(n0 < (1ULL << 31), PGX_EARG, "chunk too large: %zu L0 minimizers (use more index chunks)", n0);
  if (p->want_l0) {
    DevBuf<pgx_mm_count> mc;
    size_t nmc = 0;
    dev_count(l0.p, n0, kbits, mc, nmc);
    out->l0 = download_list(l0, n0), out->n_l0 = n0;
    out->l0_mc = download_list(mc, nmc), out->n_l0_mc = nmc;
    sync();
  }
  dev_reduce(l0.p, n0, p->reduction, l1, n1);
  l0.release();
  const DevBuf<pgx_mm128> *top = &l1;
  size_t ntop = n1;
  if (p->levels > 1) {
    dev_reduce(l1.p, n1, p->reduction, l2, n2);
    l1.release();
    top = &l2, ntop = n2;
  }
  if (p->levels >= 1) {
    DevBuf<pgx_mm_count> mc;
    size_t nmc = 0;
    dev_count(top->p, ntop, kbits, mc, nmc);
    out->top = download_list(*top, ntop), out->n_top = ntop;
    out->top_mc = download_list(mc, nmc), out->n_top_mc = nmc;
    sync();
  }
  timing_flush();
  out->gpu_ms = now_ms() - t0;
}

void write_counted(const std::string &path, const void *data, size_t n, size_t elem) {
  FILE *f = fopen(path.c_str(), "wb");
  PGX_REQUIRE(f, PGX_EIO, "file '%s' open error", path.c_str());
  uint64_t n64 = n;
  bool ok = fwrite(&n64, 8, 1, f) == 1 && (n == 0 || fwrite(data, elem, n, f) == n);
  ok = (fclose(f) == 0) && ok;
  PGX_REQUIRE(ok, PGX_EIO, "short write to '%s'", path.c_str());
}

std::string level_path(const char *prefix, int level, bool mc, int chunk, int total) {
  char buf[8400];
  snprintf(buf, sizeof(buf), "%s-L%d-%s%02d-of-%02d.dat", prefix, level, mc ? "MC-" : "", chunk, total);
  return buf;
}

}  // namespace

namespace pgx {
void index_stage(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *out, DeviceIndex *keep, bool host_arrays) {
  MemTag mem_tag("index");
  run_index(db, p, out, keep, host_arrays);
}
}  // namespace pgx

// what pgx_index_resident_dev hands out as library-owned device memory; released by the next call and by pgx_shutdown()
namespace {
struct HeldIndex {
  pgx::DeviceIndex ix;
  pgx::DevBuf<pgx_mm128> top;
} g_held;
pgx::ShutdownHook g_held_reset([] { g_held.ix = pgx::DeviceIndex(); g_held.top.release(); });
}  // namespace
namespace pgx {
bool index_owns(const void *p) {
  const pgx_mm128 *q = (const pgx_mm128 *)p;
  return ws_contains(p) || (g_held.top.p && q >= g_held.top.p && q < g_held.top.p + g_held.top.n);
}
}  // namespace pgx

extern "C" {

void pgx_index_result_free(pgx_index_result *r) {
  if (!r) return;
  out_free(r->l0), out_free(r->l0_mc), out_free(r->top), out_free(r->top_mc);
  r->l0 = r->top = nullptr;
  r->l0_mc = r->top_mc = nullptr;
  r->n_l0 = r->n_l0_mc = r->n_top = r->n_top_mc = 0;
}

int pgx_index_resident(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *out) {
  try {
    require_ready();
    PGX_REQUIRE(db && out, PGX_EARG, "pgx_index_resident: null argument");
    check_params(p);
    run_index(db, p, out);
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

// index stage of a multi-GPU rank: the final-level list and its counts stay in HBM for the exchange that follows
int pgx_index_resident_dev(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *stats, const pgx_mm128 **d_top,
                           size_t *n_top, const pgx_mm_count **d_mc, size_t *n_mc) {
  DeviceIndex &held = g_held.ix;             // one context per process: the previous call's buffers are released here
  DevBuf<pgx_mm128> &held_top = g_held.top;  // (only when the general index path produced host arrays)
  try {
    require_ready();
    PGX_REQUIRE(db && stats && d_top && n_top && d_mc && n_mc, PGX_EARG, "pgx_index_resident_dev: null argument");
    check_params(p);
    PGX_REQUIRE(!p->want_l0, PGX_EARG, "pgx_index_resident_dev returns the final level only");
    held = DeviceIndex();
    held_top.release();
    run_index(db, p, stats, &held, false);
    if (!held.valid) {  // general path (other w / k, ambiguous bases ...): its result is in host arrays
      held_top.alloc(stats->n_top);
      held_top.upload(stats->top, stats->n_top);
      held.mc.alloc(stats->n_top_mc);
      held.mc.upload(stats->top_mc, stats->n_top_mc);
      sync();
      held.d_top = held_top.p, held.n_top = stats->n_top, held.n_mc = stats->n_top_mc, held.valid = true;
      out_free(stats->top), out_free(stats->top_mc);
      stats->top = nullptr, stats->top_mc = nullptr;
    }
    *d_top = held.d_top, *n_top = held.n_top, *d_mc = held.mc.p, *n_mc = held.n_mc;
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
  return PGX_OK;
}

int pgx_index_chunk_db(pgx_seqdb *db, const char *out_prefix, const pgx_index_params *p, pgx_index_result *stats) {
  pgx_index_result res;
  memset(&res, 0, sizeof(res));
  int rc = PGX_OK;
  try {
    require_ready();
    PGX_REQUIRE(db && out_prefix, PGX_EARG, "pgx_index_chunk_db: null argument");
    check_params(p);
    DeviceIndex dev;
    run_index(db, p, &res, &dev, true);
    // file names and order of writing as in shmr_index.c:165-233
    if (p->want_l0 == 1) {
      write_counted(level_path(out_prefix, 0, false, p->mychunk, p->total_chunk), res.l0, res.n_l0, sizeof(pgx_mm128));
      write_counted(level_path(out_prefix, 0, true, p->mychunk, p->total_chunk), res.l0_mc, res.n_l0_mc, sizeof(pgx_mm_count));
    }
    if (p->levels >= 1) {
      const int lv = p->levels == 1 ? 1 : 2;
      write_counted(level_path(out_prefix, lv, false, p->mychunk, p->total_chunk), res.top, res.n_top, sizeof(pgx_mm128));
      write_counted(level_path(out_prefix, lv, true, p->mychunk, p->total_chunk), res.top_mc, res.n_top_mc, sizeof(pgx_mm_count));
      // a resident database serves a JOB: the device copies stay for its overlap commands (pgx_served.cpp: list_stash)
      const bool d = dev.valid && dev.n_top == res.n_top && dev.n_mc == res.n_top_mc;
      list_stash_put(level_path(out_prefix, lv, false, p->mychunk, p->total_chunk), d ? dev.d_top : nullptr, res.top, res.n_top * sizeof(pgx_mm128));
      list_stash_put(level_path(out_prefix, lv, true, p->mychunk, p->total_chunk), d ? dev.mc.p : nullptr, res.top_mc, res.n_top_mc * sizeof(pgx_mm_count));
    }
  } catch (const Fail &f) {
    rc = f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    rc = PGX_ENOMEM;
  }
  if (stats) {
    *stats = res;
    stats->l0 = stats->top = nullptr;
    stats->l0_mc = stats->top_mc = nullptr;
  }
  pgx_index_result_free(&res);
  return rc;
}

int pgx_index_chunk(const char *seqdb_prefix, const char *out_prefix, const pgx_index_params *p,
                    pgx_index_result *stats) {
  pgx_seqdb *db = nullptr;
  try {
    require_ready();
    PGX_REQUIRE(seqdb_prefix && out_prefix, PGX_EARG, "pgx_index_chunk: null prefix");
    check_params(p);
  } catch (const Fail &f) {
    return f.code;
  }
  int rc = pgx_seqdb_load(seqdb_prefix, &db);
  if (rc) return rc;
  rc = pgx_index_chunk_db(db, out_prefix, p, stats);
  pgx_seqdb_free(db);
  return rc;
}

}  // extern "C"
