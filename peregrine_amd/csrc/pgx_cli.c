/*
 * pgx_cli.c -- native drop-ins for the reference's stage executables, one multi-call binary (the tool is chosen by the
 * name it is invoked under): shmr_mkseqdb, shmr_index, shmr_overlap, shmr_dedup, shmr_map.  Same getopt strings, defaults
 * and output files / streams as /root/reference/src/shmr_mkseqdb.c:14-128, shmr_index.c:37-245, shmr_overlap.c:233-419,
 * shmr_dedup.c:19-104, shmr_map.c:163-373; everything else happens behind the C-ABI of include/pgx.h on the GPU.
 * Plain C against libpgx.so: this is the cgo / FFI-free form of the boundary (the Python shims in bin/ do the same).
 *
 * Resident mode (round 4): `pgx_cli serve -p <seqdb_prefix>` loads the read database into HBM ONCE and then serves shmr_index /
 * shmr_overlap commands for that prefix over a UNIX socket (<seqdb_prefix>.pgx.sock, mode 0600).  The drop-ins look for that socket
 * first: if a server for their -p prefix answers -- and the .seqdb file is still the one it loaded (size + mtime) -- the command runs
 * there (pgx_index_chunk_db / pgx_overlap_chunk_db: same files written, same exit status) and the process start, HIP context and
 * 93 GB upload that each of pg_run.py's 8 + 8 chunk commands would pay are paid once per job.  No socket, a stale one, a different
 * file: the drop-in runs stand-alone as before.  PGX_NO_SERVER=1 never attaches.
 */
#include <errno.h>
#include <libgen.h>
#include <pthread.h>
#include <limits.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include "../../include/pgx.h"

static int fail(const char *tool, const char *what) {
  fprintf(stderr, "%s: %s failed: %s\n", tool, what, pgx_last_error());
  return 1;
}

static int device_of_env(void) {
  const char *d = getenv("PGX_DEVICE");
  if (d) return atoi(d);
  const char *lr = getenv("LOCAL_RANK");
  return lr ? atoi(lr) : 0;
}

/* ---- resident mode: wire format = one request per connection: "<tool>\0<cwd>\0<size>\0<mtime_ns>\0<argc>\0<argv0>\0...";
 * reply = "<exit status>\0<text for stderr>" ------------------------------------------------------------------------------------ */
static void sock_path_of(const char *prefix, char *out, size_t cap) { snprintf(out, cap, "%s.pgx.sock", prefix); }
static int seqdb_identity(const char *prefix, long long *size, long long *mtime_ns) {
  char path[PATH_MAX];
  struct stat sb;
  snprintf(path, sizeof(path), "%s.seqdb", prefix);
  if (stat(path, &sb)) return -1;
  *size = (long long)sb.st_size;
  *mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec;
  return 0;
}
static int write_all(int fd, const char *b, size_t n) {
  while (n) {
    const ssize_t w = write(fd, b, n);
    if (w < 0 && errno == EINTR) continue;
    if (w <= 0) return -1;
    b += w, n -= (size_t)w;
  }
  return 0;
}
/* client side: 0..255 = the served command's exit status; -1 = no server took it (run stand-alone) */
static int try_server(const char *tool, const char *prefix, int argc, char **argv) {
  if (getenv("PGX_NO_SERVER")) return -1;
  struct sockaddr_un sa;
  char sp[PATH_MAX];
  sock_path_of(prefix, sp, sizeof(sp));
  if (strlen(sp) >= sizeof(sa.sun_path) || access(sp, F_OK)) return -1;
  long long size = 0, mt = 0;
  if (seqdb_identity(prefix, &size, &mt)) return -1;
  const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
  if (fd < 0) return -1;
  memset(&sa, 0, sizeof(sa));
  sa.sun_family = AF_UNIX;
  strcpy(sa.sun_path, sp);
  if (connect(fd, (struct sockaddr *)&sa, sizeof(sa))) {   /* (blocks while the server's backlog is full: a queued command WAITS, it does not
                                                             * fall back to a stand-alone run beside the server) */
    close(fd);
    return -1;
  }
  {
    struct timeval io = {10, 0};   /* (the request must go out promptly; the reply takes as long as the stage does) */
    (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &io, sizeof(io));
  }
  char cwd[PATH_MAX], num[64];
  if (!getcwd(cwd, sizeof(cwd))) cwd[0] = 0;
  int ok = !write_all(fd, tool, strlen(tool) + 1) && !write_all(fd, cwd, strlen(cwd) + 1);
  snprintf(num, sizeof(num), "%lld", size), ok = ok && !write_all(fd, num, strlen(num) + 1);
  snprintf(num, sizeof(num), "%lld", mt), ok = ok && !write_all(fd, num, strlen(num) + 1);
  snprintf(num, sizeof(num), "%d", argc), ok = ok && !write_all(fd, num, strlen(num) + 1);
  for (int i = 0; ok && i < argc; ++i) ok = !write_all(fd, argv[i], strlen(argv[i]) + 1);
  shutdown(fd, SHUT_WR);
  char reply[8192];
  size_t got = 0;
  for (;;) {
    const ssize_t r = read(fd, reply + got, sizeof(reply) - 1 - got);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    got += (size_t)r;
    if (got == sizeof(reply) - 1) break;
  }
  close(fd);
  reply[got] = 0;
  if (!ok || got < 2) return -1;
  const int status = atoi(reply);
  if (status == 250) return -1; /* the server declined (another file behind that prefix now): stand-alone */
  const char *text = reply + strlen(reply) + 1;
  if (text < reply + got && *text) fputs(text, stderr);
  return status;
}

/* An output the SERVER cannot open on the client's behalf: /dev/stdout, /dev/fd/N, /proc/self/fd/N (process substitution) name
 * the calling process's descriptors, and anything that exists and is not a regular file (a FIFO, a socket, a tty) may mean something else over
 * there too.  Such a command runs stand-alone, where fopen(path) does what the reference's does (src/shmr_overlap.c:341-352). */
static int output_needs_this_process(const char *path) {
  if (!path) return 0;
  if (!strncmp(path, "/dev/std", 8) || !strncmp(path, "/dev/fd/", 8) || !strncmp(path, "/dev/tty", 8) || !strncmp(path, "/proc/", 6)) return 1;   /* (not /dev/shm/...: plain files) */
  struct stat sb;
  if (stat(path, &sb) == 0 && !S_ISREG(sb.st_mode)) return 1;
  return 0;
}

static int run_index_args(int argc, char **argv, pgx_seqdb *db, const char *served_prefix, char *msg, size_t cap);
static int run_overlap_args(int argc, char **argv, pgx_seqdb *db, const char *served_prefix, char *msg, size_t cap, pgx_output **pending);

static void send_reply(int fd, int status, const char *msg) {
  char head[16];
  snprintf(head, sizeof(head), "%d", status);
  if (!write_all(fd, head, strlen(head) + 1)) (void)write_all(fd, msg, strlen(msg) + 1);
  close(fd);
}
/* A served overlap command whose GPU stage is done: its output file is completed (pgx_output_finish) and its client answered on a thread of
 * its own, while the server's loop already runs the next command's stage (round 6). */
struct finisher {
  int fd, status;
  pgx_output *pending;
  char msg[4096];
};
static void *finish_and_reply(void *arg) {
  struct finisher *f = (struct finisher *)arg;
  if (pgx_output_finish(f->pending)) {
    const size_t n = strlen(f->msg);
    snprintf(f->msg + n, sizeof(f->msg) - n, "shmr_overlap: pgx_output_finish failed: %s\n", pgx_last_error());
    f->status = 1;
  }
  send_reply(f->fd, f->status, f->msg);
  free(f);
  return NULL;
}

static volatile sig_atomic_t g_stop = 0;
static char g_sock[PATH_MAX];
static void on_term(int sig) {
  (void)sig;
  g_stop = 1;
  if (g_sock[0]) unlink(g_sock);
  _exit(0);
}
static int main_serve(int argc, char **argv) {
  const char *prefix = "seq_dataset";
  int c, idle_s = 0;
  while ((c = getopt(argc, argv, "p:i:")) != -1) {
    if (c == 'p') prefix = optarg;
    else if (c == 'i') idle_s = atoi(optarg); /* leave after that many seconds without a request (0: never) */
    else {
      fprintf(stderr, "Usage: pgx_cli serve -p seqdb_prefix [-i idle_seconds]\n");
      return 1;
    }
  }
  long long size = 0, mt = 0;
  if (seqdb_identity(prefix, &size, &mt)) {
    fprintf(stderr, "pgx_cli serve: %s.seqdb: %s\n", prefix, strerror(errno));
    return 1;
  }
  if (pgx_init(device_of_env())) return fail("pgx_cli serve", "pgx_init");
  pgx_seqdb *db = NULL;
  if (pgx_seqdb_load(prefix, &db)) return fail("pgx_cli serve", "pgx_seqdb_load");
  struct sockaddr_un sa;
  sock_path_of(prefix, g_sock, sizeof(g_sock));
  if (strlen(g_sock) >= sizeof(sa.sun_path)) {
    fprintf(stderr, "pgx_cli serve: socket path too long: %s\n", g_sock);
    return 1;
  }
  unlink(g_sock);
  const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
  memset(&sa, 0, sizeof(sa));
  sa.sun_family = AF_UNIX;
  strcpy(sa.sun_path, g_sock);
  const mode_t old = umask(0177);
  if (ls < 0 || bind(ls, (struct sockaddr *)&sa, sizeof(sa)) || listen(ls, 64)) {
    fprintf(stderr, "pgx_cli serve: cannot listen on %s: %s\n", g_sock, strerror(errno));
    return 1;
  }
  umask(old);
  signal(SIGTERM, on_term), signal(SIGINT, on_term), signal(SIGPIPE, SIG_IGN);
  fprintf(stderr, "pgx_cli serve: %s.seqdb (%llu reads, %llu bases) resident on GPU %d; serving %s\n", prefix, (unsigned long long)pgx_seqdb_reads(db),
          (unsigned long long)pgx_seqdb_bases(db), device_of_env(), g_sock);
  while (!g_stop) {
    if (idle_s > 0) {
      fd_set rf;
      struct timeval tv = {idle_s, 0};
      FD_ZERO(&rf);
      FD_SET(ls, &rf);
      if (select(ls + 1, &rf, NULL, NULL, &tv) == 0) break;
    }
    const int fd = accept(ls, NULL, NULL);
    if (fd < 0) {
      if (errno == EINTR) continue;
      break;
    }
    {  /* a stalled or half-open client must not hold every later command of the job: the request is a few hundred bytes */
      struct timeval io = {10, 0};
      (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &io, sizeof(io));
      (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &io, sizeof(io));
    }
    static char req[1 << 16];
    size_t got = 0;
    for (;;) {
      const ssize_t r = read(fd, req + got, sizeof(req) - 1 - got);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      got += (size_t)r;
      if (got == sizeof(req) - 1) break;
    }
    req[got] = 0;
    /* split at the NULs */
    char *fld[520];
    int nf = 0;
    for (size_t o = 0; o < got && nf < 520; o += strlen(req + o) + 1) fld[nf++] = req + o;
    char msg[4096];
    msg[0] = 0;
    int status = 250;
    pgx_output *pending = NULL;
    if (nf >= 5) {
      const char *tool = fld[0], *cwd = fld[1];
      const long long csize = atoll(fld[2]), cmt = atoll(fld[3]);
      const int ac = atoi(fld[4]);
      long long nsize = 0, nmt = 0;
      if (ac >= 1 && 5 + ac <= nf && csize == size && cmt == mt && !seqdb_identity(prefix, &nsize, &nmt) && nsize == size && nmt == mt) {
        char here[PATH_MAX];
        if (!getcwd(here, sizeof(here))) here[0] = 0;
        if (cwd[0] && chdir(cwd)) snprintf(msg, sizeof(msg), "pgx_cli serve: cannot enter %s\n", cwd), status = 1;
        else {
          optind = 0; /* (glibc: 0 re-initialises getopt completely, also after a scan that stopped inside a clustered option) */
          if (!strcmp(tool, "shmr_index")) status = run_index_args(ac, fld + 5, db, prefix, msg, sizeof(msg));
          else if (!strcmp(tool, "shmr_overlap")) status = run_overlap_args(ac, fld + 5, db, prefix, msg, sizeof(msg), &pending);
          if (here[0] && chdir(here)) status = status ? status : 1;
        }
      }
    }
    if (pending) {   /* the records are still on their way to the file: finish + reply on a thread, the loop takes the next command */
      struct finisher *f = (struct finisher *)malloc(sizeof(*f));
      pthread_t th;
      pthread_attr_t at;
      pthread_attr_init(&at);
      pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
      if (f) f->fd = fd, f->status = status, f->pending = pending, snprintf(f->msg, sizeof(f->msg), "%s", msg);
      if (!f || pthread_create(&th, &at, finish_and_reply, f)) {   /* no thread: the same inline */
        if (pgx_output_finish(pending)) status = 1;
        free(f);
        send_reply(fd, status, msg);
      }
      pthread_attr_destroy(&at);
    } else {
      send_reply(fd, status, msg);
    }
  }
  unlink(g_sock);
  pgx_seqdb_free(db);
  return 0;
}

static int main_mkseqdb(int argc, char **argv) {
  const char *list = "seq_dataset.lst", *prefix = "seq_dataset";
  int c;
  while ((c = getopt(argc, argv, "d:p:")) != -1) {
    if (c == 'd') list = optarg;
    else if (c == 'p') prefix = optarg;
    else {
      fprintf(stderr, "Usage: shmr_mkseqdb -d seq_dataset.lst -p seq_dataset_prefix\n");
      return 1;
    }
  }
  printf("input sequence dataset file list: '%s'\noutput index file: %s.idx\noutput seqdb file: %s.idx\n", list, prefix, prefix);
  if (pgx_init(device_of_env())) return fail("shmr_mkseqdb", "pgx_init");
  uint64_t nr = 0, nb = 0;
  if (pgx_mkseqdb(list, prefix, &nr, &nb)) return fail("shmr_mkseqdb", "pgx_mkseqdb");
  return 0;
}

/* db != NULL: the served form (the database of served_prefix is resident); messages go to msg instead of stderr */
static int run_index_args(int argc, char **argv, pgx_seqdb *db, const char *served_prefix, char *msg, size_t cap) {
  const char *p = "seq_dataset", *o = "shimmer";
  pgx_index_params ip = {1, 1, 2, 6, 80, 16, 1}; /* -t -c -l -r -w -k -m : shmr_index.c:21-23,49-55 */
  int c;
  while ((c = getopt(argc, argv, "p:o:t:c:l:r:m:w:k:")) != -1) {
    switch (c) {
      case 'p': p = optarg; break;
      case 'o': o = optarg; break;
      case 't': ip.total_chunk = atoi(optarg); break;
      case 'c': ip.mychunk = atoi(optarg); break;
      case 'l': ip.levels = atoi(optarg); break;
      case 'r': ip.reduction = atoi(optarg); break;
      case 'm': ip.want_l0 = atoi(optarg); break;
      case 'w': ip.window = atoi(optarg); break;
      case 'k': ip.kmer = atoi(optarg); break;
      default: return 1; /* the reference returns 1 on a missing option argument (:93-107) */
    }
  }
  if (!db) {
    const int served = try_server("shmr_index", p, argc, argv);
    if (served >= 0) return served;
    fprintf(stderr, "reduction factor= %d\nusing index file: %s.idx\nusing seqdb file: %s.seqdb\n", ip.reduction, p, p);
    if (pgx_init(device_of_env())) return fail("shmr_index", "pgx_init");
    if (pgx_index_chunk(p, o, &ip, NULL)) return fail("shmr_index", "pgx_index_chunk");
    return 0;
  }
  if (strcmp(p, served_prefix)) return 250; /* not the database this server holds */
  int n = snprintf(msg, cap, "reduction factor= %d\nusing index file: %s.idx\nusing seqdb file: %s.seqdb (resident: pgx_cli serve)\n", ip.reduction, p, p);
  if (pgx_index_chunk_db(db, o, &ip, NULL)) {
    snprintf(msg + n, cap - (size_t)n, "shmr_index: pgx_index_chunk_db failed: %s\n", pgx_last_error());
    return 1;
  }
  return 0;
}
static int main_index(int argc, char **argv) { return run_index_args(argc, argv, NULL, NULL, NULL, 0); }

static int run_overlap_args(int argc, char **argv, pgx_seqdb *db, const char *served_prefix, char *msg, size_t cap, pgx_output **pending) {
  const char *p = "seq_dataset", *l = "shimmer-L2", *o = NULL;
  pgx_overlap_params op = {1, 1, 4, 2, 240, 100, 120}; /* -t -c -b -m -M -w -n : shmr_overlap.c:28-42,245-251 */
  char dflt[64];
  int c;
  while ((c = getopt(argc, argv, "p:l:t:c:b:o:m:M:w:n:")) != -1) {
    switch (c) {
      case 'p': p = optarg; break;
      case 'l': l = optarg; break;
      case 'o': o = optarg; break;
      case 't': op.total_chunk = atoi(optarg); break;
      case 'c': op.mychunk = atoi(optarg); break;
      case 'b': op.bestn = atoi(optarg); break;
      case 'm': op.mc_lower = atoi(optarg); break;
      case 'M': op.mc_upper = atoi(optarg); break;
      case 'w': op.align_bandwidth = atoi(optarg); break;
      case 'n': op.ovlp_upper = atoi(optarg); break;
      default: return 1;
    }
  }
  if (!o) { /* :341-344 */
    snprintf(dflt, sizeof(dflt), "ovlp.%02d", op.mychunk);
    o = dflt;
  }
  if (!db) {
    const int served = output_needs_this_process(o) ? -1 : try_server("shmr_overlap", p, argc, argv);
    if (served >= 0) return served;
    if (pgx_init(device_of_env())) return fail("shmr_overlap", "pgx_init");
    if (pgx_overlap_chunk(p, l, o, &op, NULL)) return fail("shmr_overlap", "pgx_overlap_chunk");
    return 0;
  }
  if (strcmp(p, served_prefix)) return 250;
  if (pgx_overlap_chunk_db_begin(db, l, o, &op, NULL, pending)) {   /* (the caller finishes the output: pgx_output_finish) */
    snprintf(msg, cap, "shmr_overlap: pgx_overlap_chunk_db failed: %s\n", pgx_last_error());
    return 1;
  }
  return 0;
}
static int main_overlap(int argc, char **argv) { return run_overlap_args(argc, argv, NULL, NULL, NULL, 0, NULL); }

static int main_dedup(int argc, char **argv) {
  (void)argc, (void)argv;
  size_t cap = 1 << 20, n = 0;
  char *buf = (char *)malloc(cap);
  if (!buf) return 1;
  for (;;) { /* the whole ovlp_t stream from stdin (cat ovlp*.dat | shmr_dedup, pg_run.py:351-352) */
    if (n == cap) {
      char *nb = (char *)realloc(buf, cap *= 2);
      if (!nb) {
        free(buf);
        return 1;
      }
      buf = nb;
    }
    const size_t got = fread(buf + n, 1, cap - n, stdin);
    if (got == 0) break;
    n += got;
  }
  if (pgx_init(device_of_env())) return fail("shmr_dedup", "pgx_init");
  char *text = NULL;
  size_t len = 0;
  uint64_t nu = 0;
  if (pgx_dedup((const pgx_ovlp *)buf, n / sizeof(pgx_ovlp), &text, &len, &nu)) return fail("shmr_dedup", "pgx_dedup");
  fwrite(text, 1, len, stdout);
  pgx_free(text);
  free(buf);
  return 0;
}

static int main_map(int argc, char **argv) {
  const char *r = "ref", *m = "ref-L2", *p = "seq_dataset", *l = "shimmer-L2";
  pgx_map_params mp = {1, 1, 1, 240}; /* -t -c -n -M : shmr_map.c:28-29,176-177 */
  int c;
  while ((c = getopt(argc, argv, "r:m:p:l:M:n:t:c:b:")) != -1) {
    switch (c) {
      case 'r': r = optarg; break;
      case 'm': m = optarg; break;
      case 'p': p = optarg; break;
      case 'l': l = optarg; break;
      case 'M': mp.mc_upper = atoi(optarg); break;
      case 'n': mp.mc_lower = atoi(optarg); break;
      case 't': mp.total_chunk = atoi(optarg); break;
      case 'c': mp.mychunk = atoi(optarg); break;
      case 'b': break; /* accepted by the reference's getopt string, never used */
      default: return 1;
    }
  }
  if (pgx_init(device_of_env())) return fail("shmr_map", "pgx_init");
  char *text = NULL;
  size_t len = 0;
  uint64_t nl = 0;
  if (pgx_map_chunk(r, m, p, l, &mp, &text, &len, &nl)) return fail("shmr_map", "pgx_map_chunk");
  fwrite(text, 1, len, stdout);
  pgx_free(text);
  return 0;
}

int main(int argc, char **argv) {
  char *self = strdup(argv[0]);
  const char *tool = basename(self);
  if (argc > 1 && strcmp(tool, "pgx_cli") == 0) { /* pgx_cli <tool> args... */
    tool = argv[1];
    ++argv, --argc;
  }
  int rc;
  if (strcmp(tool, "shmr_mkseqdb") == 0) rc = main_mkseqdb(argc, argv);
  else if (strcmp(tool, "shmr_index") == 0) rc = main_index(argc, argv);
  else if (strcmp(tool, "shmr_overlap") == 0) rc = main_overlap(argc, argv);
  else if (strcmp(tool, "shmr_dedup") == 0) rc = main_dedup(argc, argv);
  else if (strcmp(tool, "shmr_map") == 0) rc = main_map(argc, argv);
  else if (strcmp(tool, "serve") == 0) rc = main_serve(argc, argv);
  else {
    fprintf(stderr, "usage: pgx_cli {shmr_mkseqdb|shmr_index|shmr_overlap|shmr_dedup|shmr_map} [flags]   (or invoke through a link of that name)\n"
                    "       pgx_cli serve -p seqdb_prefix [-i idle_seconds]   (keeps the read database in HBM; the drop-ins attach to it)\n");
    rc = 2;
  }
  fflush(stdout);
  fflush(stderr);
  /* The results are in their files (closed) or on stdout (flushed): leave without tearing the HIP runtime down -- unmapping the
   * device allocations, the pinned pools and the runtime's own threads costs 50-150 ms that no caller waits for anything in.
   * PGX_CLI_TEARDOWN=1 keeps the orderly exit (leak checkers). */
  if (getenv("PGX_CLI_TEARDOWN")) {
    pgx_shutdown();
    free(self);
    return rc;
  }
  _exit(rc);
}
