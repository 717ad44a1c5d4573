/*
 * pgx.h -- C-ABI of libpgx.so: the MI355X (gfx950) implementation of Peregrine's SHIMMER index + overlap
 * hot path.  Plain pointers and sizes only; no torch / C++ types.  Every entry point returns 0 on success
 * or a negative PGX_E* code (the reference exit(1)s / asserts instead: shmr_index.c:15-19,111-114,
 * shmr_utils.c:101-104); pgx_last_error() gives the message.  One context per process, one process per GPU.
 *
 * What each group replaces in the reference (/root/reference):
 *   stage level    : main() of src/shmr_index.c:37-245 and src/shmr_overlap.c:233-419 (the two executables that
 *                    py/scripts/pg_run.py:232-244,305-317 runs per chunk)
 *   resident level : the same stages with the seqdb already in HBM (what bench.py times)
 *   batch level    : mm_sketch (src/mm_sketch.c:70-151), mm_reduce (src/shmr_reduce.c:53-90), mm_count
 *                    (src/shmr_utils.c:131-160), ovlp_match (src/DWmatch.c:66-204) over many reads / pairs
 *   shimmer4py     : the cdef surface of py/peregrine/build_shimmer4py.py:8-84 (same symbol names and C
 *                    signatures) so a ctypes / cffi-ABI loader can stand in for peregrine._shimmer4py
 */
#ifndef PGX_H
#define PGX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGX_OK 0
#define PGX_EARG -1    /* bad argument (the reference would assert) */
#define PGX_EIO -2     /* file open/read/write failure (the reference would exit(1)) */
#define PGX_EHIP -3    /* HIP runtime error / no device */
#define PGX_ENOMEM -4
#define PGX_ESTATE -5  /* pgx_init not called; or the call needs something the state no longer has (the seqdb bytes after pgx_seqdb_release_bytes) */

/* ---- types shared with the on-disk formats (src/shimmer.h:24-30,61-64,97-110) ---- */
typedef struct { uint64_t x, y; } pgx_mm128;                        /* x = hash<<8|span ; y = rid<<32|lastPos<<1|strand */
typedef struct { uint64_t mer; uint32_t count; uint32_t pad; } pgx_mm_count;
typedef struct { int32_t m_size, dist, q_bgn, q_end, t_bgn, t_end, t_m_end, q_m_end; } pgx_match;
typedef struct {
  uint64_t y0, y1;
  uint32_t rl0, rl1;
  uint8_t strand0, strand1, ovlp_type, pad0;
  pgx_match match;
  uint32_t pad1;
} pgx_ovlp; /* 64 bytes; pad0/pad1 are written as 0 */

/* one candidate alignment: query = read rid0 from byte q_off on strand dir0, target = whole read rid1 on dir1
 * (the arguments shimmer_to_overlap passes to ovlp_match, src/shmr_overlap.c:117-125) */
typedef struct { uint32_t rid0, rid1, q_off; uint8_t dir0, dir1, pad[2]; } pgx_align_key;

/* one shimmer-pair record of build_map (src/shmr_utils.c:337-400: mp128_t {y0, y1, direction} filed under [key0][key1]) in the
 * form the ranks of a multi-GPU job exchange: npos = ~((y0 & 0xFFFFFFFF) >> 1), the descending-position sort key */
typedef struct { uint64_t key0, key1, y0; uint32_t npos; uint8_t dir, pad[3]; } pgx_pair_rec; /* 32 bytes */

typedef struct pgx_seqdb pgx_seqdb; /* opaque: a read database resident in HBM */

/* ---- context ---- */
int pgx_init(int device);            /* select the GPU, create the stream; idempotent */
void pgx_shutdown(void);
const char *pgx_last_error(void);
int pgx_device_count(void);
const char *pgx_version(void);
void pgx_free(void *p);              /* releases any host array returned by this library */

/* per-kernel device time (HIP events on the library's stream), accumulated since the last reset.
 * names: "sketch", "sketch_general", "sketch_redo", "sketch_nreads", "sketch_gather", "pack", "reduce", "count", "pairs", "replay_dense" / "replay_rows" / "replay_update" (k_eval, k_eval_rows, k_update of the device replay; only with PGX_REPLAY_TIMING=1), "align" (k_align4), "align1" (k_align1: launches of
 * at most 13 k alignments), "encode", "dedup", "map". */
int pgx_timing_get(const char *kernel, double *total_ms, uint64_t *launches, uint64_t *units);
void pgx_timing_reset(void);
/* HBM ledger: JSON text of the library's device memory -- live bytes, bytes held in the block cache, and the live bytes BY OWNER (seqdb,
 * packs, index workspaces, join tables, replay tables ...) at the moment the live total peaked; the named workspaces; what the driver
 * reports as used.  Returns the length of the text (buf may be NULL to ask for it).  reset_peak != 0: the peak starts again from now. */
int pgx_mem_ledger(char *buf, size_t cap, int reset_peak);

/* ---- resident read database ---- */
/* rid/rlen/roff: the idx file's columns in file order (src/shmr_mkseqdb.c:111-112); copies to HBM. */
int pgx_seqdb_upload(const uint8_t *seqdb, size_t nbytes, const uint32_t *rid, const uint32_t *rlen,
                     const uint64_t *roff, uint32_t nreads, pgx_seqdb **out);
int pgx_seqdb_load(const char *seqdb_prefix, pgx_seqdb **out); /* reads <prefix>.idx + <prefix>.seqdb */
/* the same with the seqdb bytes already in HBM (e.g. all-gathered over xGMI by the ranks of a multi-GPU job): d_seqdb is a
 * device pointer, copied device-to-device; rid/rlen/roff are host arrays */
int pgx_seqdb_upload_dev(const uint8_t *d_seqdb, size_t nbytes, const uint32_t *rid, const uint32_t *rlen,
                         const uint64_t *roff, uint32_t nreads, pgx_seqdb **out);
/* the same WITHOUT a copy: the library reads the bytes where they are (d_seqdb stays owned by the caller and must outlive the
 * pgx_seqdb).  capacity >= nbytes + 1024: the library zeroes [nbytes, nbytes + 1024) once (its kernels' wide loads may run
 * that far past the last read).  This is what lets the ranks of a multi-GPU job all-gather the job's read set straight into
 * its final place -- one copy of the seqdb per GPU instead of three (93 GB at 30x human). */
int pgx_seqdb_adopt_dev(uint8_t *d_seqdb, size_t nbytes, size_t capacity, const uint32_t *rid, const uint32_t *rlen,
                        const uint64_t *roff, uint32_t nreads, pgx_seqdb **out);
void pgx_seqdb_free(pgx_seqdb *db);
/* The seqdb's BYTES out of HBM once the 2-bit packs exist (a quarter of the size, the same information; built by the first overlap stage on
 * the database, or here): the index stage's closed-form kernels and every alignment kernel of the default path read the packs.  Refused --
 * PGX_ESTATE, bytes kept -- for a database with a read that holds an ambiguous base (sketched run by run / aligned nibble by nibble from the
 * bytes: src/mm_sketch.c:112-113, src/DWmatch.c:136-137) or a read longer than 65,535 bases.  Afterwards the entry points that need bytes (w / k
 * other than 80 / 16, want_l0, pgx_sketch_batch) return PGX_ESTATE; a buffer handed over with pgx_seqdb_adopt_dev is no longer referenced. */
int pgx_seqdb_release_bytes(pgx_seqdb *db);
int pgx_seqdb_has_bytes(const pgx_seqdb *db);
uint64_t pgx_seqdb_bases(const pgx_seqdb *db);
uint32_t pgx_seqdb_reads(const pgx_seqdb *db);

/* ---- index stage (replaces shmr_index) ---- */
typedef struct {
  int total_chunk;   /* -t */
  int mychunk;       /* -c, 1-based; selects reads with rid % t == c % t (shmr_index.c:157) */
  int levels;        /* -l 1|2 */
  int reduction;     /* -r (<256) */
  int window;        /* -w (24..255) */
  int kmer;          /* -k (12..28) */
  int want_l0;       /* -m : also return / write the L0 list and its counts */
} pgx_index_params;

typedef struct {
  pgx_mm128 *l0; size_t n_l0;           /* only when want_l0 */
  pgx_mm_count *l0_mc; size_t n_l0_mc;
  pgx_mm128 *top; size_t n_top;         /* L1 (levels==1) or L2 */
  pgx_mm_count *top_mc; size_t n_top_mc;/* sorted by mer (the reference writes khash slot order; consumers only aggregate) */
  uint64_t bases; uint32_t reads;       /* what this chunk sketched */
  uint32_t reads_literal;               /* reads sketched run by run (ambiguous bases, or flagged by a closed-form kernel: pgx_sketch_n.hip); the name is round 1's */
  double gpu_ms;                        /* device time of this call */
} pgx_index_result;

int pgx_index_resident(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *out);
void pgx_index_result_free(pgx_index_result *r);
/* file level: same flags, same output file names/bytes as shmr_index (MC files: same (mer,count) multiset) */
int pgx_index_chunk(const char *seqdb_prefix, const char *out_prefix, const pgx_index_params *p,
                    pgx_index_result *stats /* nullable; arrays are not returned */);

/* ---- sequence database (replaces shmr_mkseqdb, src/shmr_mkseqdb.c:99-121) ----
 * seq_dataset_path: text file with one FASTA/FASTQ(.gz) path per whitespace-separated token; writes <prefix>.seqdb and
 * <prefix>.idx byte-identical to the reference's.  The two-strand encoding runs on the GPU. */
int pgx_mkseqdb(const char *seq_dataset_path, const char *seqdb_prefix, uint64_t *n_reads, uint64_t *n_bases);

/* ---- overlap stage (replaces shmr_overlap) ---- */
typedef struct {
  int total_chunk, mychunk;  /* -t -c : bucket ownership (x>>8) % t == c % t (shmr_utils.c:337,362) */
  int bestn;                 /* -b (uint8 in the reference) */
  int mc_lower, mc_upper;    /* -m -M */
  int align_bandwidth;       /* -w */
  int ovlp_upper;            /* -n */
} pgx_overlap_params;

typedef struct {
  uint64_t n_records;        /* ovlp_t records produced */
  uint64_t n_pair_records;   /* shimmer-pair records built (build_map) */
  uint64_t n_buckets;        /* buckets processed (2 < n <= ovlp_upper) */
  uint64_t n_align_needed;   /* alignments the reference would have computed */
  uint64_t n_align_gpu;      /* alignments computed on the GPU (>= needed: speculative replay) */
  uint64_t n_seen_skip;
  uint32_t rounds;           /* replay rounds until the fixed point */
  double gpu_ms;             /* device time */
  double host_ms;            /* host orchestration time (order emulation + replay) */
  uint64_t n_evaluations;    /* bucket evaluations of the replay (>= n_buckets: the fixed point re-evaluates) */
  uint32_t device_replay;    /* 1: the greedy walk ran on the GPU (pgx_replay.hip); 0: on the host threads */
  uint32_t device_visit;     /* 0: bucket visit order built by host threads; else 1 + the number of first-key groups a whole
                                wavefront replayed (pgx_visit.hip; the others take a lane each) */
  uint32_t replay_attempts;  /* device replay: attempts until the tables were large enough (1: the first sizes held; the sizes a
                                stage needed are where the next stage of the process starts) */
  uint32_t reserved0;
  uint64_t stream_checksum;  /* order-sensitive 64-bit checksum of the records' fields (padding bytes excluded), computed where the
                                records are written: two stages with equal checksums produced the same stream (0: not computed) */
} pgx_overlap_stats;

/* mmers: concatenation of all index chunks' final-level lists in chunk order; counts: all MC entries */
int pgx_overlap_resident(pgx_seqdb *db, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
                         size_t n_counts, const pgx_overlap_params *p, pgx_ovlp **out, size_t *n_out,
                         pgx_overlap_stats *stats);
/* index + overlap of a single-chunk job in one call: the final-level list and its counts stay in HBM between the two
 * stages (the reference hands them over through files).  ip must name chunk 1 of 1.  want_index_arrays == 0: index_out
 * carries the sizes / statistics only (top, top_mc stay NULL).  Results are those of pgx_index_resident followed by
 * pgx_overlap_resident. */
int pgx_index_overlap_resident(pgx_seqdb *db, const pgx_index_params *ip, const pgx_overlap_params *op,
                               int want_index_arrays, pgx_index_result *index_out, pgx_ovlp **out, size_t *n_out,
                               pgx_overlap_stats *stats);
/* ---- multi-GPU hand-over, everything device-resident (one process per GPU, chunk c on rank c-1) ----
 * The reference couples the chunks through files: every overlap chunk reads EVERY index chunk's list and counts
 * (src/shmr_overlap.c:359-384) and keeps the records whose first key it owns (src/shmr_utils.c:337,362).  Here each rank
 * builds the pair records of its OWN index chunk's reads and routes them to the owner chunk: the caller moves (1) the count
 * tables with an all-gather and (2) the records with an all-to-all(v) (RCCL), both on device pointers, between these calls:
 *   pgx_index_resident_dev   : index stage; the final-level list and its counts stay in HBM (*d_top, *d_mc: library-owned, valid
 *                              until the next index call of this process)
 *   pgx_pairs_prepare_dev    : d_counts_all = all chunks' count tables concatenated; flags the kept shimmers of d_top;
 *                              *first_strict = list index of the first shimmer with lower <= count < upper, -1 if none (the global
 *                              scan starts there, src/shmr_utils.c:311-320: the caller passes `start` = that index on the first rank
 *                              that has one, 0 on later ranks, -1 on earlier ranks)
 *   pgx_pairs_scatter_dev    : *d_send = the records grouped by destination chunk 1..total_chunk, scan order inside a group
 *                              (library-owned, valid until the next prepare); send_counts[c-1] = records for chunk c
 *   pgx_overlap_records_dev  : overlap stage of chunk p->mychunk over the records received, concatenated in SOURCE chunk order
 *                              (then scan order) = the insertion order of build_map over the concatenated lists; results are
 *                              those of pgx_overlap_resident on the concatenated lists */
int pgx_index_resident_dev(pgx_seqdb *db, const pgx_index_params *p, pgx_index_result *stats, const pgx_mm128 **d_top,
                           size_t *n_top, const pgx_mm_count **d_mc, size_t *n_mc);
int pgx_pairs_prepare_dev(pgx_seqdb *db, const pgx_mm128 *d_top, size_t n_top, const pgx_mm_count *d_counts_all,
                          size_t n_counts_all, int mc_lower, int mc_upper, int64_t *first_strict);
int pgx_pairs_scatter_dev(pgx_seqdb *db, int total_chunk, int64_t start, const pgx_pair_rec **d_send, uint64_t *send_counts);
int pgx_overlap_records_dev(pgx_seqdb *db, const pgx_pair_rec *d_records, size_t n_records, const pgx_overlap_params *p,
                            pgx_ovlp **out, size_t *n_out, pgx_overlap_stats *stats);
/* overlap stage over DEVICE lists (the concatenation of all index chunks' lists / counts, e.g. after an all-gather) */
int pgx_overlap_resident_dev(pgx_seqdb *db, const pgx_mm128 *d_mmers, size_t n_mm, const pgx_mm_count *d_counts,
                             size_t n_counts, const pgx_overlap_params *p, pgx_ovlp **out, size_t *n_out,
                             pgx_overlap_stats *stats);
/* Ordering against another runtime's stream (e.g. the stream torch / RCCL enqueued a collective on) without stopping the host:
 *   pgx_stream_wait(s)   : work the library enqueues from now on starts after everything enqueued on s so far
 *   pgx_stream_signal(s) : work enqueued on s from now on starts after everything the library has enqueued so far
 * s is a hipStream_t (NULL = the default stream).  The replacement for a full stream synchronisation around each hand-over. */
int pgx_stream_wait(void *other_stream);
int pgx_stream_signal(void *other_stream);
/* plain device-to-device copy on the library's stream, synchronous (for callers that hold device memory of another runtime) */
int pgx_copy_dev(void *d_dst, const void *d_src, size_t nbytes);

/* file level: globs <shimmer_prefix>-[0-9]*-of-[0-9]*.dat and -MC- twins like shmr_overlap.c:355-384 */
int pgx_overlap_chunk(const char *seqdb_prefix, const char *shimmer_prefix, const char *out_path,
                      const pgx_overlap_params *p, pgx_overlap_stats *stats);
/* Asynchronous delivery of the overlap records (resident pipelines that run several chunks in one process).  pgx_results_async(1): the
 * pgx_overlap_*resident* entry points return once the copy of their records to the host array is ENQUEUED (on a stream of its own,
 * behind the kernel that writes them) instead of finished; *n_records, the statistics and the array's address are valid at once, the
 * array's CONTENT only after pgx_results_wait() -- which the next overlap stage, pgx_free of the array and pgx_results_async(0) also
 * imply.  Returns the previous setting.  Off by default: every call returns finished arrays. */
int pgx_results_async(int on);
int pgx_results_wait(void);

/* The same two stages on a read database that is ALREADY resident (pgx_seqdb_load / _upload): what a long-lived process serves
 * several chunk commands from -- `pgx_cli serve`, which the shmr_index / shmr_overlap drop-ins attach to, INTEGRATION.md section 5 --
 * so that a job's 8 + 8 chunk commands upload the seqdb once instead of sixteen times. */
int pgx_index_chunk_db(pgx_seqdb *db, const char *out_prefix, const pgx_index_params *p, pgx_index_result *stats);
int pgx_overlap_chunk_db(pgx_seqdb *db, const char *shimmer_prefix, const char *out_path, const pgx_overlap_params *p,
                         pgx_overlap_stats *stats);
/* The overlap command of a served JOB in two steps (round 6; caller contract py/scripts/pg_run.py:305-317 -- the job's commands follow
 * each other against one resident database).  pgx_overlap_chunk_db_begin returns when the GPU stage is done and the records are on
 * their way from the device to out_path (the statistics are final); pgx_output_finish blocks until that file is complete and closed,
 * reports an I/O error and releases *pending -- it may be called from another thread while the next command's _begin already runs,
 * which is how `pgx_cli serve` overlaps one command's file with the next command's kernels.  pgx_overlap_chunk_db = _begin + _finish.
 * An index command on a resident database leaves device copies of the list / count files it wrote; _begin assembles its lists from
 * those copies (checked against the files' size and mtime) and reads only files it holds no current copy of. */
typedef struct pgx_output pgx_output;
int pgx_overlap_chunk_db_begin(pgx_seqdb *db, const char *shimmer_prefix, const char *out_path, const pgx_overlap_params *p,
                               pgx_overlap_stats *stats, pgx_output **pending);
int pgx_output_finish(pgx_output *pending);

/* ---- dedup (replaces shmr_dedup, src/shmr_dedup.c:32-101) ----
 * recs: the concatenated ovlp_t streams (cat ovlp*.dat).  The first record of every read pair wins; *text receives the
 * FALCON-style overlap lines (malloc'd, release with pgx_free), byte-identical to the reference's stdout. */
int pgx_dedup(const pgx_ovlp *recs, size_t n, char **text, size_t *text_len, uint64_t *n_unique);

/* ---- reads -> contigs mapping (replaces shmr_map, src/shmr_map.c:48-161,163-373) ----
 * The reads' shimmer-pair map is built exactly as in the overlap stage (build_map with -t/-c/-n/-M); the reference
 * shimmers are chained the same way and every hit bucket is reported, one line per record in insertion order:
 * "ref_id ref_bgn ref_end read_id read_bgn read_end direction mcount0 mcount1\n" (shmr_map.c:152-153).  *text is malloc'd
 * (pgx_free) and byte-identical to the reference's stdout. */
typedef struct {
  int total_chunk, mychunk;  /* -t -c */
  int mc_lower, mc_upper;    /* -n -M (defaults 1, 240: shmr_map.c:28-29) */
} pgx_map_params;
int pgx_map(const pgx_mm128 *ref_mmers, size_t n_ref, const pgx_mm128 *mmers, size_t n_mm, const pgx_mm_count *counts,
            size_t n_counts, const uint32_t *rlen_by_rid, uint32_t n_rid, const pgx_map_params *p, char **text,
            size_t *text_len, uint64_t *n_lines);
/* file level: -r refdb_prefix (unused, as in the reference) -m ref_shimmer_prefix -p seqdb_prefix -l shimmer_prefix */
int pgx_map_chunk(const char *refdb_prefix, const char *ref_shimmer_prefix, const char *seqdb_prefix,
                  const char *shimmer_prefix, const pgx_map_params *p, char **text, size_t *text_len, uint64_t *n_lines);

/* ---- batch level ---- */
int pgx_sketch_batch(pgx_seqdb *db, const uint32_t *read_slots, uint32_t n, int w, int k, pgx_mm128 **out,
                     size_t *n_out);  /* read_slots index the idx-file order; output in that order */
int pgx_reduce_batch(const pgx_mm128 *in, size_t n, int rs, pgx_mm128 **out, size_t *n_out);
int pgx_count_batch(const pgx_mm128 *in, size_t n, pgx_mm_count **out, size_t *n_out);
int pgx_align_batch(pgx_seqdb *db, const pgx_align_key *keys, size_t n, int band, pgx_match *out);

/* host utility: the order in which klib khash (src/khash.h:232-336, integer hash :373) iterates n DISTINCT 64-bit keys
 * inserted in the given order -- the order shmr_overlap visits its tables in (src/shmr_overlap.c:206-215).  out receives
 * the n keys in ascending slot order.  Runs on the host (it is the routine the overlap stage uses for its outer table). */
int pgx_khash_slot_order(const uint64_t *keys, size_t n, uint64_t *out);
/* the same with a trailing put of a present key (touch != 0: it only runs the load check, src/khash.h:298-306, and may resize once
 * more).  (Round 3 also offered a device form of this routine -- priority insertion + a fixed point over the eviction order of a
 * resize; exact, and 90x slower than one host thread on the reference's keys, whose constant span byte makes 1 / 256 of the slots
 * the home of every key: removed in round 4) */
int pgx_khash_slot_order_ex(const uint64_t *keys, size_t n, int touch, uint64_t *out);

/* ---- shimmer4py surface (py/peregrine/build_shimmer4py.py:8-84), GPU-backed single-call forms ----
 * NOT exported: shmr_aln / free_shmr_alns (build_shimmer4py.py:64-77; src/shmr_align.c is the consensus stage's aligner, out of
 * scope per SURVEY.md section 2 #10) -- a caller that needs them keeps the reference's own shimmer4py for those two symbols. */
typedef struct { size_t n, m; pgx_mm128 *a; } mm128_v; /* kvec layout, src/shimmer.h:27-30; .a is malloc'd, caller frees */
typedef pgx_match ovlp_match_t;
void decode_biseq(uint8_t *src, char *seq, size_t len, uint8_t strand);
void encode_biseq(uint8_t *target, char *seq, size_t len);
void mm_sketch(void *km, const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p);
void mm_reduce(mm128_v *in, mm128_v *out, uint8_t rs);
ovlp_match_t *ovlp_match(uint8_t *query_seq, int32_t q_len, uint8_t q_strand, uint8_t *target_seq,
                         int32_t t_len, uint8_t t_strand, int32_t band_tolerance);
void free_ovlp_match(ovlp_match_t *m);
mm128_v read_mmlist(char *fn);
void write_mmlist(char *fn, mm128_v *p);

/* query helpers (src/shimmer4py.c:44-196, cdef py/peregrine/build_shimmer4py.py:44-62).  The map is
 * built on the GPU by build_shimmer_map4py; the three getters are host lookups over its sorted tables.  On failure
 * build_shimmer_map4py leaves every field of *py_mmer NULL (the reference exit(1)s / asserts) and the getters then return
 * nothing.  get_shimmers_for_read returns a VIEW into py_mmer->mmers (do not free); get_shimmer_hits appends to a
 * caller-owned kvec whose .a is realloc'd (free it with free / pgx_free). */
typedef struct { uint64_t x0, x1, y0, y1; uint8_t direction; } mp256_t;   /* src/shimmer.h:122-126 */
typedef struct { size_t n, m; mp256_t *a; } mp256_v;
typedef struct { mm128_v *mmers; void *mmer0_map; void *rlmap; void *mcmap; void *ridmm; } py_mmer_t; /* shimmer.h:132-138 */
void build_shimmer_map4py(py_mmer_t *py_mmer, char *seqdb_prefix, char *shimmer_prefix, uint32_t mychunk,
                          uint32_t total_chunk, uint32_t lowerbound, uint32_t upperbound);
void get_shimmers_for_read(mm128_v *out, py_mmer_t *py_mmer, uint32_t rid);
uint32_t get_mmer_count(py_mmer_t *py_mmer, uint64_t mhash);
void get_shimmer_hits(mp256_v *out, py_mmer_t *py_mmer, uint64_t mhash0, uint32_t span);
void pgx_shimmer_map_free(py_mmer_t *py_mmer); /* extension: the reference never releases the map */

#ifdef __cplusplus
}
#endif
#endif
