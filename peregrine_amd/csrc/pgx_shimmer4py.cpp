// pgx_shimmer4py.cpp -- the symbols of peregrine._shimmer4py (cdef: /root/reference/py/peregrine/
// build_shimmer4py.py:8-84) that lie on the index/overlap path, with the same names and C signatures, so a ctypes or
// cffi-ABI loader can stand in for the cffi module.  The compute (mm_sketch, mm_reduce, ovlp_match) runs on the GPU
// through the batch entry points; the codec and the list files are host-side format handling.
// Ownership follows the reference: output vectors are kvec structs whose .a is libc-malloc'd here and free()d by the
// caller (py/peregrine/utils.py:155,170,178); mm_sketch / mm_reduce APPEND to the vector they are given.
// Errors: the reference exit(1)s / asserts; these print the message to stderr and leave the output untouched.
#include "pgx_internal.h"

using namespace pgx;

namespace {
const uint8_t kFwd[4] = {1, 2, 4, 8};
inline int code_of_ascii(unsigned char c) {  // src/mm_sketch.c:10-21 (bytes 0..3 are bases too)
  if (c < 4) return c;
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}
inline uint8_t onehot_fwd(char c) {  // src/shmr_utils.c:18-30: only A C G T (either case), everything else 0
  switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 4;
    case 'T': case 't': return 8;
    default: return 0;
  }
}
inline uint8_t onehot_rev(char c) {  // src/shmr_utils.c:32-42
  switch (c) {
    case 'A': case 'a': return 8;
    case 'C': case 'c': return 4;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 1;
    default: return 0;
  }
}
void complain(const char *who, int rc) { fprintf(stderr, "%s: error %d: %s\n", who, rc, pgx_last_error()); }

void append(mm128_v *v, const pgx_mm128 *src, size_t n) {
  if (!n) return;
  if (v->n + n > v->m) {
    v->m = v->n + n;
    v->a = (pgx_mm128 *)realloc(v->a, v->m * sizeof(pgx_mm128));
  }
  memcpy(v->a + v->n, src, n * sizeof(pgx_mm128));
  v->n += n;
}
}  // namespace

extern "C" {

void encode_biseq(uint8_t *target, char *seq, size_t len) {
  for (size_t p = 0; p < len; ++p) target[p] = (uint8_t)(onehot_rev(seq[len - 1 - p]) << 4 | onehot_fwd(seq[p]));
}

void decode_biseq(uint8_t *src, char *seq, size_t len, uint8_t strand) {
  static const char nib[16] = {'N', 'A', 'C', 'N', 'G', 'N', 'N', 'N', 'T', 'N', 'N', 'N', 'N', 'N', 'N', 'N'};
  for (size_t p = 0; p < len; ++p) seq[p] = nib[strand == 0 ? (src[p] & 0x0F) : (src[p] >> 4)];
}

void mm_sketch(void *km, const char *str, int len, int w, int k, uint32_t rid, int is_hpc, mm128_v *p) {
  (void)km;
  if (is_hpc) {
    set_error("mm_sketch: is_hpc=1 is not on the index path and is not implemented");
    complain("mm_sketch", PGX_EARG);
    return;
  }
  if (len <= 0 || !p) return;
  if (!ctx().ready && pgx_init(0)) return complain("mm_sketch", PGX_EHIP);
  std::vector<uint8_t> enc((size_t)len);
  for (int i = 0; i < len; ++i) {
    const int c = code_of_ascii((unsigned char)str[i]);
    enc[i] = c < 4 ? kFwd[c] : 0;  // only the forward nibble matters to the sketch
  }
  pgx_seqdb *db = nullptr;
  const uint32_t l = (uint32_t)len, slot = 0;
  const uint64_t off = 0;
  int rc = pgx_seqdb_upload(enc.data(), enc.size(), &rid, &l, &off, 1, &db);
  pgx_mm128 *out = nullptr;
  size_t n = 0;
  if (!rc) rc = pgx_sketch_batch(db, &slot, 1, w, k, &out, &n);
  if (rc) complain("mm_sketch", rc);
  else append(p, out, n);
  free(out);
  pgx_seqdb_free(db);
}

void mm_reduce(mm128_v *in, mm128_v *out, uint8_t rs) {
  if (!in || !out) return;
  if (!ctx().ready && pgx_init(0)) return complain("mm_reduce", PGX_EHIP);
  pgx_mm128 *res = nullptr;
  size_t n = 0;
  int rc = pgx_reduce_batch(in->a, in->n, rs, &res, &n);
  if (rc) complain("mm_reduce", rc);
  else append(out, res, n);
  free(res);
}

ovlp_match_t *ovlp_match(uint8_t *query_seq, int32_t q_len, uint8_t q_strand, uint8_t *target_seq, int32_t t_len,
                         uint8_t t_strand, int32_t band_tolerance) {
  ovlp_match_t *r = (ovlp_match_t *)calloc(1, sizeof(ovlp_match_t));
  if (q_len < 0 || t_len < 0) return r;
  if (!ctx().ready && pgx_init(0)) {
    complain("ovlp_match", PGX_EHIP);
    return r;
  }
  std::vector<uint8_t> buf((size_t)q_len + (size_t)t_len);
  if (q_len) memcpy(buf.data(), query_seq, (size_t)q_len);
  if (t_len) memcpy(buf.data() + q_len, target_seq, (size_t)t_len);
  const uint32_t rid[2] = {0, 1}, len[2] = {(uint32_t)q_len, (uint32_t)t_len};
  const uint64_t off[2] = {0, (uint64_t)q_len};
  pgx_seqdb *db = nullptr;
  int rc = pgx_seqdb_upload(buf.data(), buf.size(), rid, len, off, 2, &db);
  pgx_align_key key{0, 1, 0, (uint8_t)(q_strand ? 1 : 0), (uint8_t)(t_strand ? 1 : 0), {0, 0}};
  if (!rc) rc = pgx_align_batch(db, &key, 1, band_tolerance, r);
  if (rc) complain("ovlp_match", rc);
  pgx_seqdb_free(db);
  return r;
}

void free_ovlp_match(ovlp_match_t *m) { free(m); }

mm128_v read_mmlist(char *fn) {  // src/shmr_utils.c:110-123
  mm128_v v = {0, 0, nullptr};
  std::vector<uint8_t> buf;
  if (!read_file(fn, buf) || buf.size() < 8) {
    fprintf(stderr, "file '%s' open error\n", fn);
    return v;
  }
  uint64_t n;
  memcpy(&n, buf.data(), 8);
  if (8 + n * 16 > buf.size()) n = (buf.size() - 8) / 16;
  v.n = v.m = (size_t)n;
  v.a = (pgx_mm128 *)malloc(n ? n * 16 : 1);
  if (n) memcpy(v.a, buf.data() + 8, n * 16);
  return v;
}

void write_mmlist(char *fn, mm128_v *p) {  // src/shmr_utils.c:98-108
  FILE *f = fopen(fn, "wb");
  if (!f) {
    fprintf(stderr, "file '%s' open error\n", fn);
    return;
  }
  uint64_t n = p->n;
  fwrite(&n, 8, 1, f);
  fwrite(p->a, 16, p->n, f);
  fclose(f);
}

}  // extern "C"
