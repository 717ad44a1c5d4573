/*
 * shimmer_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement, in our own words, of the SHIMMER index + overlap hot path of
 * cschin/Peregrine.  Every function cites the reference file:line it restates.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so; the product
 * (libpgx.so, peregrine_amd/) never includes, links or calls anything in this directory.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function here against the real
 * reference compiled in place from /root/reference/src into oracle/_ref/ (see oracle/Makefile), and
 * tests/golden/ holds outputs of those reference binaries for small datasets.
 */
#ifndef SHIMMER_ORACLE_H
#define SHIMMER_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/shimmer.h:24-30 (mm128_t / mm128_v) */
typedef struct { uint64_t x, y; } orc_mm_t;
typedef struct { size_t n, cap; orc_mm_t *a; } orc_mmv_t;

/* reference: src/shimmer.h:61-64 (mm_count_t, sizeof 16, 4 padding bytes) */
typedef struct { uint64_t mer; uint32_t count; uint32_t pad; } orc_mc_t;
typedef struct { size_t n, cap; orc_mc_t *a; } orc_mcv_t;

/* reference: src/shimmer.h:97-102 (ovlp_match_t) */
typedef struct {
  int32_t m_size, dist, q_bgn, q_end, t_bgn, t_end, t_m_end, q_m_end;
} orc_match_t;

/* reference: src/shimmer.h:104-110 (ovlp_t, 64 bytes; pad0 @27, pad1 @60..63) */
typedef struct {
  uint64_t y0, y1;
  uint32_t rl0, rl1;
  uint8_t strand0, strand1, ovlp_type, pad0;
  orc_match_t match;
  uint32_t pad1;
} orc_ovlp_t;
typedef struct { size_t n, cap; orc_ovlp_t *a; } orc_ovlpv_t;

typedef struct {
  uint64_t n_align;      /* ovlp_match calls */
  uint64_t n_seen_skip;  /* partner skipped because the read pair was already recorded */
  uint64_t n_buckets;    /* buckets with 2 < n <= ovlp_upper that were processed */
  uint64_t n_records;    /* pair records produced by the map build */
  uint64_t bases_cmp;    /* bases compared inside ovlp_match (for roofline accounting) */
} orc_stats_t;

/* ---- sequence codec: src/shmr_utils.c:18-62 ---- */
void orc_encode_biseq(uint8_t *dst, const char *seq, size_t len);
void orc_decode_biseq(const uint8_t *src, char *seq, size_t len, uint8_t strand);

/* ---- minimizer sketch: src/mm_sketch.c:23-32,70-151 (is_hpc==0 branch only) ---- */
uint64_t orc_hash64(uint64_t key, uint64_t mask);
void orc_sketch_ascii(const char *seq, int len, int w, int k, uint32_t rid, orc_mmv_t *out);
void orc_sketch_seqdb(const uint8_t *bytes, int len, int w, int k, uint32_t rid, orc_mmv_t *out);

/* ---- hierarchical reduction: src/shmr_reduce.c:27-90 ---- */
void orc_reduce(const orc_mmv_t *in, orc_mmv_t *out, uint8_t rs);

/* ---- multiplicity counting, khash slot order: src/shmr_utils.c:131-160 ---- */
void orc_count(const orc_mmv_t *in, orc_mcv_t *out);

/* ---- banded O(ND) confirmation: src/DWmatch.c:66-204 ---- */
void orc_ovlp_match(const uint8_t *q, int32_t q_len, uint8_t q_strand, const uint8_t *t, int32_t t_len,
                    uint8_t t_strand, int32_t band, orc_match_t *out, uint64_t *bases_cmp);

/* ---- khash slot-order emulation (keys only): src/khash.h:232-336,373 ---- */
/* inserts keys[0..n) in order (repeats allowed) and writes the distinct keys in ascending slot order;
 * returns the number of distinct keys. out must have room for n keys. */
size_t orc_khash_order(const uint64_t *keys, size_t n, uint64_t *out);

/* ---- overlap stage in memory: src/shmr_utils.c:295-404 + src/shmr_overlap.c:46-231 ----
 * mmers      : concatenation of all index chunks' final-level lists (chunk 1..N order)
 * counts     : concatenation of all MC files' entries (aggregated here, shmr_utils.c:162-176)
 * rlen/roff  : read length / seqdb byte offset indexed by rid (nreads entries)
 */
int orc_overlap(const uint8_t *seqdb, const uint32_t *rlen, const uint64_t *roff, uint32_t nreads,
                const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts,
                uint32_t mychunk, uint32_t total_chunk, uint32_t mc_lower, uint32_t mc_upper,
                uint32_t bestn, uint32_t ovlp_upper, uint32_t band, orc_ovlpv_t *out, orc_stats_t *stats);

/* ---- the insertion sequence of build_map for one overlap chunk (checks the multi-GPU record exchange) ---- */
typedef struct { uint64_t key0, key1, y0; uint32_t npos; uint8_t dir, pad[3]; } orc_pair_rec_t; /* 32 bytes */
orc_pair_rec_t *orc_pair_records(const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts, const uint32_t *rlen,
                                 uint32_t mychunk, uint32_t total_chunk, uint32_t mc_lower, uint32_t mc_upper, size_t *n_out);

/* ---- file-level stages (the CPU baseline that travels to the GPU box) ----
 * same flags / file names as src/shmr_index.c:37-245 and src/shmr_overlap.c:233-419; return 0 or -1. */
int orc_index_chunk(const char *seqdb_prefix, const char *out_prefix, int total, int mychunk, int levels,
                    int reduction, int write_l0, int w, int k, uint64_t *bases_done);
int orc_overlap_chunk(const char *seqdb_prefix, const char *shimmer_prefix, const char *out_path, int total,
                      int mychunk, int bestn, int mc_lower, int mc_upper, int band, int ovlp_upper,
                      orc_stats_t *stats, uint64_t *n_out);

/* ---- row f2: first-wins dedup + FALCON text lines, src/shmr_dedup.c:32-101.  Returns malloc'd text (orc_free). ---- */
char *orc_dedup(const orc_ovlp_t *recs, size_t n, size_t *text_len, uint64_t *n_unique);

/* ---- row f4: query helpers of src/shimmer4py.c:44-196 over the pair map of build_map ---- */
typedef struct { uint64_t x0, x1, y0, y1; uint8_t direction; uint8_t pad[7]; } orc_mp256_t; /* mp256_t, shimmer.h:122-126 */
typedef struct orc_map orc_map_t;
orc_map_t *orc_map_build(const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts, size_t n_counts, const uint32_t *rlen,
                         uint32_t mychunk, uint32_t total_chunk, uint32_t lower, uint32_t upper);
void orc_map_free(orc_map_t *m);
uint32_t orc_map_count(const orc_map_t *m, uint64_t mhash);
size_t orc_map_hits(orc_map_t *m, uint64_t mhash0, uint32_t span, orc_mp256_t **out); /* *out malloc'd (orc_free) */
void orc_read_shimmers(const orc_mm_t *mmers, size_t n_mm, uint32_t rid, size_t *first, size_t *count);

/* ---- row f3: shmr_map, src/shmr_map.c:48-161.  Returns the stdout text (malloc'd, orc_free). ---- */
char *orc_map_reads_to_ref(const orc_mm_t *ref, size_t n_ref, const orc_mm_t *mmers, size_t n_mm, const orc_mc_t *counts,
                           size_t n_counts, const uint32_t *rlen, uint32_t mychunk, uint32_t total_chunk, uint32_t lower,
                           uint32_t upper, size_t *text_len, uint64_t *n_lines);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
