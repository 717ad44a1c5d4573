/*
 * pgx_cli.c -- native drop-ins for the reference's stage executables, one multi-call binary (the tool is chosen by the
 * name it is invoked under): shmr_mkseqdb, shmr_index, shmr_overlap, shmr_dedup, shmr_map.  Same getopt strings, defaults
 * and output files / streams as /root/reference/src/shmr_mkseqdb.c:14-128, shmr_index.c:37-245, shmr_overlap.c:233-419,
 * shmr_dedup.c:19-104, shmr_map.c:163-373; everything else happens behind the C-ABI of include/pgx.h on the GPU.
 * Plain C against libpgx.so: this is the cgo / FFI-free form of the boundary (the Python shims in bin/ do the same).
 */
#include <errno.h>
#include <libgen.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

static int main_index(int argc, char **argv) {
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
  fprintf(stderr, "reduction factor= %d\nusing index file: %s.idx\nusing seqdb file: %s.seqdb\n", ip.reduction, p, p);
  if (pgx_init(device_of_env())) return fail("shmr_index", "pgx_init");
  if (pgx_index_chunk(p, o, &ip, NULL)) return fail("shmr_index", "pgx_index_chunk");
  return 0;
}

static int main_overlap(int argc, char **argv) {
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
  if (pgx_init(device_of_env())) return fail("shmr_overlap", "pgx_init");
  if (pgx_overlap_chunk(p, l, o, &op, NULL)) return fail("shmr_overlap", "pgx_overlap_chunk");
  return 0;
}

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
  else {
    fprintf(stderr, "usage: pgx_cli {shmr_mkseqdb|shmr_index|shmr_overlap|shmr_dedup|shmr_map} [flags]   (or invoke through a link of that name)\n");
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
