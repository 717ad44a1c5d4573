// pgx_served.cpp -- the FILE-level entry points of the overlap stage (pgx_overlap_chunk, pgx_overlap_chunk_db and its begin / finish
// form): what bin/native/shmr_overlap and `pgx_cli serve` call (/root/reference/src/shmr_overlap.c:233-419: read every index chunk's
// shimmer and count file, run the stage, write the headerless ovlp_t stream).  The stage itself is pgx_overlap.cpp.
//
// What a SERVED job (pg_run.py's 8 + 8 chunk commands against one resident database, py/scripts/pg_run.py:232-244,305-317) gets here:
//   * the lists stay on the device between the commands: an index command leaves a device copy of every list / count file it writes
//     (list_stash, keyed by the file's absolute path + size + mtime); an overlap command assembles its input from those copies and
//     only reads (and uploads) files it finds no current copy of; the assembled lists are kept for the job's next overlap command;
//   * the records go from the device straight to the output file: slices through a few pinned staging buffers into a shared mapping
//     of the file, several threads at once (FileSink) -- no 3 GB host array, no second pass over it;
//   * begin / finish: pgx_overlap_chunk_db_begin returns when the GPU stage is done and the file transfer is under way, so that the
//     server can start the next command's stage while the previous command's file is completed (pgx_output_finish), and answers each
//     client when ITS file is complete.
#include <fcntl.h>
#include <glob.h>
#include <limits.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pgx_internal.h"

namespace pgx {
namespace {

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

using FileId = std::pair<std::string, std::pair<long long, long long>>;   // path -> (size, mtime ns)

void glob_identity(const std::string &pat, std::vector<FileId> &out) {
  glob_t g;
  memset(&g, 0, sizeof(g));
  if (glob(pat.c_str(), 0, nullptr, &g) == 0) {   // name-sorted like wordexp in shmr_overlap.c:355-384
    for (size_t i = 0; i < g.gl_pathc; ++i) {
      struct stat sb;
      if (stat(g.gl_pathv[i], &sb) == 0)
        out.push_back({g.gl_pathv[i], {(long long)sb.st_size, (long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec}});
    }
  }
  globfree(&g);
}

std::string absolute(const std::string &path) {
  if (!path.empty() && path[0] == '/') return path;
  char cwd[PATH_MAX];
  return getcwd(cwd, sizeof(cwd)) ? std::string(cwd) + "/" + path : path;
}

// ---- device copies of the list / count files index commands of this process wrote ------------------------------------------------
struct Stashed {
  DevBuf<uint8_t> dev;    // the payload (entries only, without the 8-byte count header)
  size_t bytes = 0;
  long long size = 0, mtime_ns = 0;   // of the file as it was right after it was written
  uint64_t serial = 0;
};
std::mutex g_stash_mu;
std::map<std::string, Stashed> g_stash;
size_t g_stash_bytes = 0;
uint64_t g_stash_serial = 0;
constexpr size_t STASH_CAP = (size_t)24 << 30;   // (a full-size configs[4] job: 12.6 GB of L1 lists + counts)
ShutdownHook g_stash_reset([] { list_stash_clear(); });

// the payload of one counted file ("<uint64 n><n entries>", shmr_utils.c write_mmlist / shmr_index.c:165-233); a truncated file is an error
template <typename T>
size_t counted_entries(const std::string &path, long long size) {
  const int fd = open(path.c_str(), O_RDONLY);
  PGX_REQUIRE(fd >= 0, PGX_EIO, "file '%s' open error", path.c_str());
  uint64_t n = 0;
  const bool ok = pread(fd, &n, 8, 0) == 8;
  close(fd);
  PGX_REQUIRE(ok && size >= 8, PGX_EIO, "file '%s' open error", path.c_str());
  PGX_REQUIRE(n <= (uint64_t)(size - 8) / sizeof(T), PGX_EIO, "file '%s' is truncated: header says %llu entries, %lld bytes follow", path.c_str(),
              (unsigned long long)n, size - 8);
  return (size_t)n;
}

bool read_payload(const std::string &path, void *dst, size_t bytes) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  size_t got = 0;
  while (got < bytes) {
    const ssize_t r = pread(fd, (char *)dst + got, bytes - got, (off_t)(8 + got));
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    got += (size_t)r;
  }
  close(fd);
  return got == bytes;
}

// every file of a name-sorted group, one after the other, into ONE device array: from the stash where it holds a current copy
// (device to device), else from the file (read by a few threads side by side, then uploaded)
template <typename T>
void assemble(const std::vector<FileId> &files, DevBuf<T> &out, size_t *n_out, size_t *from_stash) {
  std::vector<size_t> cnt(files.size()), off(files.size());
  std::vector<const uint8_t *> src(files.size(), nullptr);
  size_t total = 0;
  {
    std::lock_guard<std::mutex> lk(g_stash_mu);
    for (size_t i = 0; i < files.size(); ++i) {
      auto it = g_stash.find(files[i].first);
      if (it != g_stash.end() && it->second.size == files[i].second.first && it->second.mtime_ns == files[i].second.second &&
          it->second.bytes % sizeof(T) == 0 && (long long)it->second.bytes + 8 <= files[i].second.first) {
        src[i] = it->second.dev.p, cnt[i] = it->second.bytes / sizeof(T);
        ++*from_stash;
      }
    }
  }
  for (size_t i = 0; i < files.size(); ++i) {
    if (!src[i]) cnt[i] = counted_entries<T>(files[i].first, files[i].second.first);
    off[i] = total, total += cnt[i];
  }
  out.alloc(std::max<size_t>(total, 1));
  *n_out = total;
  // the files without a device copy: host buffers filled by up to 8 threads, uploaded in order
  std::vector<size_t> todo;
  for (size_t i = 0; i < files.size(); ++i)
    if (!src[i] && cnt[i]) todo.push_back(i);
  std::vector<HostArray<uint8_t>> bufs(todo.size());
  std::vector<char> ok(todo.size(), 1);
  {
    std::atomic<size_t> next{0};
    std::vector<std::thread> ws;
    const size_t nt = std::min<size_t>(8, todo.size());
    for (size_t t = 0; t < nt; ++t)
      ws.emplace_back([&] {
        for (size_t k; (k = next.fetch_add(1)) < todo.size();) {
          try {
            bufs[k].alloc(cnt[todo[k]] * sizeof(T));
            ok[k] = read_payload(files[todo[k]].first, bufs[k].p, cnt[todo[k]] * sizeof(T)) ? 1 : 0;
          } catch (...) {
            ok[k] = 0;
          }
        }
      });
    for (auto &t : ws) t.join();
  }
  for (size_t k = 0; k < todo.size(); ++k) PGX_REQUIRE(ok[k], PGX_EIO, "file '%s' open error", files[todo[k]].first.c_str());
  hipStream_t s = ctx().stream;
  for (size_t k = 0; k < todo.size(); ++k)
    PGX_HIP(hipMemcpyAsync(out.p + off[todo[k]], bufs[k].p, cnt[todo[k]] * sizeof(T), hipMemcpyHostToDevice, s));
  for (size_t i = 0; i < files.size(); ++i)
    if (src[i] && cnt[i]) PGX_HIP(hipMemcpyAsync(out.p + off[i], src[i], cnt[i] * sizeof(T), hipMemcpyDeviceToDevice, s));
  sync();   // (the host buffers go out of scope; a stash entry may be replaced by the next index command)
}

// the lists of the last shimmer prefix, on the device, as long as the files behind it have not changed (names, sizes, mtimes): every overlap
// chunk of a job globs every index chunk (shmr_overlap.c:359-384), so the job's commands all read the same lists
struct DevListCache {
  std::string prefix;
  std::vector<FileId> mm_files, mc_files;
  DevBuf<pgx_mm128> mm;
  DevBuf<pgx_mm_count> mc;
  size_t n_mm = 0, n_mc = 0;
  void clear() {
    prefix.clear(), mm_files.clear(), mc_files.clear();
    mm.release(), mc.release();
    n_mm = n_mc = 0;
  }
};
DevListCache g_lists;
ShutdownHook g_lists_reset([] { g_lists.clear(); });

// ---- the records to out_path --------------------------------------------------------------------------------------------------------
// From a host array (stand-alone commands, the host replay).  A regular file: several threads pwrite slices into the page cache.
// Anything that cannot seek (-o /dev/stdout, a FIFO, a process substitution -- the reference's fwrite stream handles those,
// shmr_overlap.c:388-390): one sequential write loop.  EINTR is retried.
void write_records(const char *out_path, const pgx_ovlp *rec, size_t n) {
  const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  PGX_REQUIRE(fd >= 0, PGX_EIO, "file '%s' open error", out_path);
  const size_t total = n * sizeof(pgx_ovlp);
  struct stat sb;
  const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
  bool ok = true;
  if (!regular) {
    for (size_t off = 0; off < total;) {
      const ssize_t w = write(fd, (const char *)rec + off, total - off);
      if (w < 0 && errno == EINTR) continue;
      if (w <= 0) {
        ok = false;
        break;
      }
      off += (size_t)w;
    }
  } else {
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(8, total >> 24));
    std::vector<char> okv(nt, 1);
    std::vector<std::thread> ws;
    for (int t = 0; t < nt; ++t)
      ws.emplace_back([&, t] {
        const size_t lo = total * t / nt, hi = total * (t + 1) / nt;
        for (size_t off = lo; off < hi;) {
          const ssize_t w = pwrite(fd, (const char *)rec + off, hi - off, (off_t)off);
          if (w < 0 && errno == EINTR) continue;
          if (w <= 0) {
            okv[t] = 0;
            return;
          }
          off += (size_t)w;
        }
      });
    for (auto &t : ws) t.join();
    for (char c : okv) ok = ok && c;
  }
  ok = (close(fd) == 0) && ok;
  PGX_REQUIRE(ok, PGX_EIO, "short write to '%s'", out_path);
}

// From the device (served commands).  Buffered writes to one file are serialised by the inode lock (8 pwrite threads: 6-7 GB/s into
// /dev/shm, 0.45 s for the 2.9 GB of a human-scale chunk, after a 2.9 GB copy to a host array); page faults of a shared mapping are
// not: NT threads each bring slices of SLICE bytes down on their own stream into their own pinned buffer and copy them into the mapping
// of the (pre-sized) file.  Where the file cannot be mapped the slices are pwritten.
constexpr int SINK_NT = 12;
constexpr size_t SINK_SLICE = (size_t)8 << 20;
struct SinkLane {
  hipStream_t stream = nullptr;
  void *pin = nullptr;
};
SinkLane g_lane[SINK_NT];
hipEvent_t g_sink_ready = nullptr;
std::mutex g_sink_mu;   // one transfer at a time (the lanes are shared): g_sink_busy, taken by take() and given back by the transfer's thread
std::condition_variable g_sink_cv;
bool g_sink_busy = false;
void sink_acquire() {
  std::unique_lock<std::mutex> lk(g_sink_mu);
  g_sink_cv.wait(lk, [] { return !g_sink_busy; });
  g_sink_busy = true;
}
void sink_release() {
  {
    std::lock_guard<std::mutex> lk(g_sink_mu);
    g_sink_busy = false;
  }
  g_sink_cv.notify_all();
}
ShutdownHook g_sink_reset([] {
  sink_acquire();
  for (auto &l : g_lane) {
    if (l.stream) (void)hipStreamDestroy(l.stream), l.stream = nullptr;
    if (l.pin) (void)hipHostFree(l.pin), l.pin = nullptr;
  }
  if (g_sink_ready) (void)hipEventDestroy(g_sink_ready), g_sink_ready = nullptr;
  sink_release();
});

struct FileSink : RecordSink {
  std::string path;
  int fd = -1;
  bool taken = false;
  DevBuf<pgx_ovlp> dev;
  size_t n = 0;
  std::thread runner;
  std::atomic<int> failed{0};
  double t_take = 0, t_done = 0;

  explicit FileSink(const char *out_path) : path(out_path) {
    fd = open(out_path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    PGX_REQUIRE(fd >= 0, PGX_EIO, "file '%s' open error", out_path);
  }
  static bool usable(const char *out_path) {   // a regular file (or a name that does not exist yet): anything else takes the host-array path
    struct stat sb;
    return stat(out_path, &sb) != 0 || S_ISREG(sb.st_mode);
  }
  void take(DevBuf<pgx_ovlp> &&d, size_t count) override {
    dev = std::move(d), n = count, taken = true;
    t_take = now_ms();
    sink_acquire();   // (given back by the runner)
    try {
      if (!g_sink_ready) PGX_HIP(hipEventCreateWithFlags(&g_sink_ready, hipEventDisableTiming));
      for (auto &l : g_lane) {
        if (!l.stream) PGX_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        if (!l.pin) PGX_HIP(hipHostMalloc(&l.pin, SINK_SLICE, hipHostMallocDefault));
      }
      PGX_HIP(hipEventRecord(g_sink_ready, ctx().stream));
      for (auto &l : g_lane) PGX_HIP(hipStreamWaitEvent(l.stream, g_sink_ready, 0));
    } catch (...) {
      sink_release();
      throw;
    }
    const int device = ctx().device;
    runner = std::thread([this, device] {
      const size_t total = n * sizeof(pgx_ovlp);
      char *map = nullptr;
      if (ftruncate(fd, (off_t)total) != 0) failed = 1;
      if (!failed && total) {
        void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) map = (char *)m;
      }
      const size_t n_slices = (total + SINK_SLICE - 1) / SINK_SLICE;
      std::atomic<size_t> next{0};
      std::vector<std::thread> ws;
      for (int t = 0; t < SINK_NT && (size_t)t < n_slices && !failed; ++t)
        ws.emplace_back([&, t] {
          if (hipSetDevice(device) != hipSuccess) {
            failed = 1;
            return;
          }
          for (size_t i; (i = next.fetch_add(1)) < n_slices && !failed;) {
            const size_t off = i * SINK_SLICE, len = std::min(SINK_SLICE, total - off);
            if (hipMemcpyAsync(g_lane[t].pin, (const char *)dev.p + off, len, hipMemcpyDeviceToHost, g_lane[t].stream) != hipSuccess ||
                hipStreamSynchronize(g_lane[t].stream) != hipSuccess) {
              failed = 1;
              return;
            }
            if (map) {
              memcpy(map + off, g_lane[t].pin, len);
            } else {
              for (size_t w0 = 0; w0 < len;) {
                const ssize_t w = pwrite(fd, (const char *)g_lane[t].pin + w0, len - w0, (off_t)(off + w0));
                if (w < 0 && errno == EINTR) continue;
                if (w <= 0) {
                  failed = 1;
                  return;
                }
                w0 += (size_t)w;
              }
            }
          }
        });
      for (auto &t : ws) t.join();
      if (map && munmap(map, total) != 0) failed = 1;
      dev.release();
      t_done = now_ms();
      sink_release();
    });
  }
  // the file is complete and closed when this returns; throws on an I/O error
  void finish() {
    if (runner.joinable()) runner.join();
    bool ok = !failed;
    if (fd >= 0) ok = (close(fd) == 0) && ok, fd = -1;
    PGX_REQUIRE(ok, PGX_EIO, "short write to '%s'", path.c_str());
  }
  ~FileSink() override {
    if (runner.joinable()) runner.join();
    if (fd >= 0) close(fd);
  }
};

// the shimmer / count files of every index chunk, name-sorted as the reference's wordexp globs them (shmr_overlap.c:359-384), into host arrays
template <typename T>
void read_counted_files(const std::string &pattern, std::vector<T> &out) {
  std::vector<FileId> files;
  glob_identity(pattern, files);
  for (const FileId &f : files) {
    std::vector<uint8_t> buf;
    PGX_REQUIRE(read_file(f.first, buf) && buf.size() >= 8, PGX_EIO, "file '%s' open error", f.first.c_str());
    uint64_t n;
    memcpy(&n, buf.data(), 8);
    // a truncated index chunk must not yield a quietly smaller overlap set
    PGX_REQUIRE(n <= (buf.size() - 8) / sizeof(T), PGX_EIO, "file '%s' is truncated: header says %llu entries, %zu bytes follow", f.first.c_str(),
                (unsigned long long)n, buf.size() - 8);
    const size_t o = out.size();
    out.resize(o + n);
    if (n) memcpy(out.data() + o, buf.data() + 8, n * sizeof(T));
  }
}
void read_index_files(const char *shimmer_prefix, std::vector<pgx_mm128> &mm, std::vector<pgx_mm_count> &mc) {
  read_counted_files(std::string(shimmer_prefix) + "-[0-9]*-of-[0-9]*.dat", mm);
  read_counted_files(std::string(shimmer_prefix) + "-MC-[0-9]*-of-[0-9]*.dat", mc);
}

}  // namespace

void list_stash_put(const std::string &path, const void *d_payload, const void *h_payload, size_t bytes) {
  const std::string abs = absolute(path);
  struct stat sb;
  if (stat(abs.c_str(), &sb) != 0 || (long long)sb.st_size != (long long)bytes + 8) return;
  Stashed e;
  MemTag tag("served.list_stash");
  e.dev.alloc(std::max<size_t>(bytes, 1));
  if (bytes) {
    if (d_payload) PGX_HIP(hipMemcpyAsync(e.dev.p, d_payload, bytes, hipMemcpyDeviceToDevice, ctx().stream));
    else PGX_HIP(hipMemcpyAsync(e.dev.p, h_payload, bytes, hipMemcpyHostToDevice, ctx().stream));
    sync();
  }
  e.bytes = bytes, e.size = (long long)sb.st_size, e.mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec;
  std::lock_guard<std::mutex> lk(g_stash_mu);
  e.serial = ++g_stash_serial;
  auto it = g_stash.find(abs);
  if (it != g_stash.end()) g_stash_bytes -= it->second.bytes, g_stash.erase(it);
  g_stash_bytes += bytes;
  g_stash.emplace(abs, std::move(e));
  while (g_stash_bytes > STASH_CAP && g_stash.size() > 1) {   // the oldest copies go first
    auto old = g_stash.begin();
    for (auto jt = g_stash.begin(); jt != g_stash.end(); ++jt)
      if (jt->second.serial < old->second.serial) old = jt;
    g_stash_bytes -= old->second.bytes;
    g_stash.erase(old);
  }
}
void list_stash_clear() {   // (also the assembled lists of the last job: both belong to the database that is going away)
  g_lists.clear();
  std::lock_guard<std::mutex> lk(g_stash_mu);
  g_stash.clear();
  g_stash_bytes = 0;
}

}  // namespace pgx

using namespace pgx;

struct pgx_output {   // a served command whose GPU stage is done and whose output file is being completed
  FileSink *sink = nullptr;
  bool trace = false;
  double t_stage_done = 0;
  size_t n = 0;
};

extern "C" {

int pgx_output_finish(pgx_output *o) {
  if (!o) return PGX_OK;
  int rc = PGX_OK;
  try {
    if (o->sink) {
      o->sink->finish();
      if (o->trace)
        fprintf(stderr, "[pgx] overlap chunk (resident database): %zu records in their file %.1f ms after the stage (transfer %.1f ms)\n", o->n,
                now_ms() - o->t_stage_done, o->sink->t_done - o->sink->t_take);
    }
  } catch (const Fail &f) {
    rc = f.code;
  }
  delete o->sink;
  delete o;
  return rc;
}

int pgx_overlap_chunk_db_begin(pgx_seqdb *db, const char *shimmer_prefix, const char *out_path, const pgx_overlap_params *p,
                               pgx_overlap_stats *stats, pgx_output **pending) {
  int rc = PGX_OK;
  FileSink *sink = nullptr;
  if (pending) *pending = nullptr;
  try {
    require_ready();
    PGX_REQUIRE(db && shimmer_prefix && out_path && pending, PGX_EARG, "pgx_overlap_chunk_db_begin: null argument");
    overlap_check_params(p);
    const bool trace = getenv("PGX_TRACE") != nullptr;
    const double t0 = now_ms();
    // identity = the ABSOLUTE prefix (the server enters each client's directory: two jobs with the same relative prefix are different
    // files) + every file's name, size and mtime, taken before AND after the lists are assembled (a file rewritten in between
    // is not cached)
    const std::string abs_prefix = absolute(shimmer_prefix);
    auto identity = [&](std::vector<FileId> &mm, std::vector<FileId> &mc) {
      glob_identity(abs_prefix + "-[0-9]*-of-[0-9]*.dat", mm);
      glob_identity(abs_prefix + "-MC-[0-9]*-of-[0-9]*.dat", mc);
    };
    std::vector<FileId> mm_files, mc_files;
    identity(mm_files, mc_files);
    size_t from_stash = 0;
    DevListCache &c = g_lists;
    if (c.prefix != abs_prefix || c.mm_files != mm_files || c.mc_files != mc_files || mm_files.empty()) {
      c.clear();   // (another prefix: the old lists' memory goes back first)
      MemTag tag("served.lists");
      assemble(mm_files, c.mm, &c.n_mm, &from_stash);
      assemble(mc_files, c.mc, &c.n_mc, &from_stash);
      std::vector<FileId> mm2, mc2;
      identity(mm2, mc2);
      if (mm2 == mm_files && mc2 == mc_files) c.prefix = abs_prefix, c.mm_files = mm_files, c.mc_files = mc_files;
    }
    const double t1 = now_ms();
    OvOut v;
    if (FileSink::usable(out_path)) sink = new FileSink(out_path);
    const DeviceLists dl{c.mm.p, c.mc.p};
    record_sink() = sink;
    try {
      overlap_stage(db, nullptr, c.n_mm, nullptr, c.n_mc, p, v, stats, &dl);
    } catch (...) {
      record_sink() = nullptr;
      throw;
    }
    record_sink() = nullptr;
    if (c.prefix.empty()) c.clear();   // (lists that could not be tied to their files are not kept)
    const double t2 = now_ms();
    if (!sink || !sink->taken) {   // the host replay / an empty set / an output that cannot seek: the records are a host array
      results_wait();
      if (sink) delete sink, sink = nullptr;
      write_records(out_path, v.a, v.a ? v.n : 0);
    }
    if (trace)
      fprintf(stderr, "[pgx] overlap chunk (resident database): lists %.1f ms (%zu + %zu files, %zu from device copies), stage %.1f ms, %zu records%s\n", t1 - t0,
              mm_files.size(), mc_files.size(), from_stash, t2 - t1, v.n, sink ? " on their way to the file" : " written");
    pgx_output *o = new pgx_output;
    o->sink = sink, o->trace = trace, o->t_stage_done = t2, o->n = v.n;
    sink = nullptr;
    *pending = o;
  } catch (const Fail &f) {
    rc = f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    rc = PGX_ENOMEM;
  }
  delete sink;
  return rc;
}

int pgx_overlap_chunk_db(pgx_seqdb *db, const char *shimmer_prefix, const char *out_path, const pgx_overlap_params *p,
                         pgx_overlap_stats *stats) {
  pgx_output *o = nullptr;
  const int rc = pgx_overlap_chunk_db_begin(db, shimmer_prefix, out_path, p, stats, &o);
  if (rc) return rc;
  return pgx_output_finish(o);
}

int pgx_overlap_chunk(const char *seqdb_prefix, const char *shimmer_prefix, const char *out_path,
                      const pgx_overlap_params *p, pgx_overlap_stats *stats) {
  pgx_seqdb *db = nullptr;
  int rc = PGX_OK;
  try {
    require_ready();
    PGX_REQUIRE(seqdb_prefix && shimmer_prefix && out_path, PGX_EARG, "pgx_overlap_chunk: null argument");
    overlap_check_params(p);
    // the shimmer / count files are read by a second thread WHILE the seqdb goes to HBM (they are independent inputs)
    std::vector<pgx_mm128> mm;
    std::vector<pgx_mm_count> mc;
    int rd_code = PGX_OK;
    std::string rd_err;
    std::thread reader([&] {
      try {
        read_index_files(shimmer_prefix, mm, mc);
      } catch (const Fail &f) {
        rd_code = f.code, rd_err = pgx_last_error();
      } catch (const std::bad_alloc &) {
        rd_code = PGX_ENOMEM, rd_err = "out of host memory";
      } catch (...) {
        rd_code = PGX_EIO, rd_err = "reading the shimmer files failed";
      }
    });
    rc = pgx_seqdb_load(seqdb_prefix, &db);
    const std::string load_err = rc ? pgx_last_error() : "";
    reader.join();
    if (rc) {
      set_error("%s", load_err.c_str());
      return rc;
    }
    PGX_REQUIRE(rd_code == PGX_OK, rd_code, "%s", rd_err.c_str());
    OvOut v;
    overlap_stage(db, mm.data(), mm.size(), mc.data(), mc.size(), p, v, stats);
    results_wait();
    write_records(out_path, v.a, v.n);
  } catch (const Fail &f) {
    rc = f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    rc = PGX_ENOMEM;
  }
  pgx_seqdb_free(db);
  return rc;
}

}  // extern "C"
