// pgx_mkseqdb.hip -- SURVEY.md 8(f) row f1: what /root/reference/src/shmr_mkseqdb.c:99-121 does (FASTA/FASTQ(.gz) -> seqdb +
// idx), with the two-strand encoding (encode_biseq, src/shmr_utils.c:44-51) on the GPU.
//
// Host: the file list is whitespace separated (fscanf "%s", shmr_mkseqdb.c:99); every file is inflated with zlib and parsed
// with the record grammar of the reference's kseq reader (restated, not copied): a record starts at '>' or '@'; the name is
// the header up to the first whitespace; sequence lines are concatenated until a line starts with '>', '@' or '+'; after
// '+' as many quality characters as sequence characters follow.  rid counts records across files; idx lines are
// "%09d %s %u %lu\n" (rid, name, length, byte offset).
// GPU: k_encode_biseq, one workgroup per read: byte p = onehot(base p) | onehot(complement(base len-1-p)) << 4 -- a pure
// streaming kernel (1 B/base read + mirrored re-read that hits L2, 1 B/base written).
#include <zlib.h>

#include "pgx_internal.h"

namespace pgx {
namespace {

__device__ __forceinline__ uint32_t onehot_fwd(uint32_t c) {
  c |= 0x20;  // A/a C/c G/g T/t only (src/shmr_utils.c:18-30); everything else encodes as 0
  return c == 'a' ? 1u : c == 'c' ? 2u : c == 'g' ? 4u : c == 't' ? 8u : 0u;
}
__device__ __forceinline__ uint32_t onehot_rev(uint32_t c) {
  c |= 0x20;
  return c == 'a' ? 8u : c == 'c' ? 4u : c == 'g' ? 2u : c == 't' ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_encode_biseq(const uint8_t *__restrict__ ascii, const uint64_t *__restrict__ off,
                                                      uint32_t n, uint8_t *__restrict__ out) {
  const uint32_t r = blockIdx.x;
  if (r >= n) return;
  const uint64_t o = off[r];
  const uint32_t len = (uint32_t)(off[r + 1] - o);
  const uint8_t *s = ascii + o;
  uint8_t *d = out + o;
  for (uint32_t p = threadIdx.x; p < len; p += blockDim.x)
    d[p] = (uint8_t)(onehot_rev(s[len - 1 - p]) << 4 | onehot_fwd(s[p]));
}

inline bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

struct Records {
  std::vector<std::string> names;
  std::vector<uint64_t> off;  // n+1 offsets into seq
  std::vector<uint8_t> seq;
  void clear() {
    names.clear(), seq.clear();
    off.assign(1, 0);
  }
};
struct ParseState {
  int last = 0;          // the header marker ('>' / '@') the reader has already consumed, or 0
  bool stopped = false;  // a malformed FASTQ record ended the file's records (as the reference's loop ends)
};

// The kseq record grammar (see the header comment) over the bytes b[0, n) of a file that is read PIECE BY PIECE: complete
// records are appended to `out`; the return value is the number of bytes consumed, i.e. where parsing resumes once more
// data has been appended behind b[n).  A record whose parse reached the end of the piece is not committed unless the piece
// ends the file (`at_eof`): it is parsed again, from its start, with more data.  At the end of the file the reader's
// end-of-file behaviour applies (a header marker at the very end, a FASTQ record without its quality string, ... end the
// records like the reference's reader does).
size_t parse_some(const uint8_t *b, size_t n, bool at_eof, ParseState &st, Records &out) {
  size_t i = 0;
  int last = st.last;
  for (;;) {
    const size_t rec_i = i;      // where this record's parse starts, and in which reader state
    const int rec_last = last;
    const size_t s0 = out.seq.size();
    auto more_needed = [&]() {   // ran into the end of the piece: come back with more data
      out.seq.resize(s0);
      st.last = rec_last;
      return rec_i;
    };
    if (last == 0) {
      while (i < n && b[i] != '>' && b[i] != '@') ++i;
      if (i >= n) {
        st.last = 0;
        return n;  // (nothing but skipped bytes)
      }
      last = b[i++];
    }
    // name: up to the first whitespace; the rest of the header line is the comment
    if (i >= n) {
      if (!at_eof) return more_needed();
      st.last = last;
      return n;  // header marker at the very end: the reference's reader reports EOF
    }
    std::string name;
    while (i < n && !is_space(b[i])) name.push_back((char)b[i++]);
    int c = i < n ? b[i++] : -1;
    if (c != '\n' && c != -1)
      while (i < n && b[i++] != '\n') {
      }
    c = -1;
    while (i < n) {
      c = b[i++];
      if (c == '>' || c == '+' || c == '@') break;
      if (c == '\n') {
        c = -1;
        continue;
      }
      out.seq.push_back((uint8_t)c);
      while (i < n && b[i] != '\n') out.seq.push_back(b[i++]);
      if (i < n) ++i;  // the newline
      if (out.seq.size() - s0 > 1 && out.seq.back() == '\r') out.seq.pop_back();
      c = -1;
    }
    const size_t slen = out.seq.size() - s0;
    last = (c == '>' || c == '@') ? c : 0;
    bool bad = false;
    if (c == '+') {  // FASTQ: skip the '+' line, then read at least slen quality characters
      while (i < n && b[i] != '\n') ++i;
      if (i >= n) {  // no quality string: error, the record is dropped and reading stops
        bad = true;
      } else {
        ++i;
        size_t ql = 0;
        while (i < n) {
          const size_t ls = i;
          while (i < n && b[i] != '\n') ++i;
          const size_t ll = i - ls;
          if (i < n) ++i;
          ql += ll;
          if (ql > 1 && ll > 0 && b[ls + ll - 1] == '\r') --ql;
          if (ql >= slen) break;
        }
        if (ql != slen) bad = true;
      }
    }
    if (i >= n && !at_eof) return more_needed();  // (the record may continue in the next piece)
    if (bad) {
      out.seq.resize(s0);
      st.stopped = true;
      st.last = 0;
      return n;
    }
    out.names.push_back(name);
    out.off.push_back(out.seq.size());
    if (i >= n) {  // (at_eof)
      st.last = last;
      return n;
    }
  }
}

}  // namespace
}  // namespace pgx

using namespace pgx;

extern "C" int pgx_mkseqdb(const char *seq_dataset_path, const char *seqdb_prefix, uint64_t *n_reads, uint64_t *n_bases) {
  try {
    require_ready();
    PGX_REQUIRE(seq_dataset_path && seqdb_prefix, PGX_EARG, "pgx_mkseqdb: null argument");
    FILE *lst = fopen(seq_dataset_path, "r");
    PGX_REQUIRE(lst, PGX_EIO, "file '%s' open error", seq_dataset_path);
    std::string pre(seqdb_prefix);
    FILE *fidx = fopen((pre + ".idx").c_str(), "w");
    FILE *fdb = fopen((pre + ".seqdb").c_str(), "wb");
    if (!fidx || !fdb) {
      if (fidx) fclose(fidx);
      if (fdb) fclose(fdb);
      fclose(lst);
      PGX_REQUIRE(false, PGX_EIO, "cannot create %s.idx / %s.seqdb", seqdb_prefix, seqdb_prefix);
    }
    uint64_t rid = 0, offset = 0;
    char fn[8192];
    int rc = PGX_OK;
    // Bounded host memory (the reference streams record by record): the file is inflated PIECE bytes at a time, complete
    // records are collected until BATCH bases are pending, encoded by one kernel launch and appended to the output files.
    const size_t PIECE = getenv("PGX_MKSEQDB_PIECE") ? (size_t)atoll(getenv("PGX_MKSEQDB_PIECE")) : ((size_t)64 << 20);
    const size_t BATCH = getenv("PGX_MKSEQDB_BATCH") ? (size_t)atoll(getenv("PGX_MKSEQDB_BATCH")) : ((size_t)256 << 20);
    Records rec;
    rec.clear();
    std::vector<uint8_t> enc;
    auto flush = [&]() -> bool {  // encode and write what is pending
      const uint32_t n = (uint32_t)rec.names.size();
      const size_t nb = rec.seq.size();
      if (!n) return true;
      enc.resize(nb);
      if (nb) {
        KernelTimer tm("encode", nb);
        uint8_t *d_in = ws<uint8_t>("mk.in", nb), *d_out = ws<uint8_t>("mk.out", nb);
        uint64_t *d_off = ws<uint64_t>("mk.off", (size_t)n + 1);
        PGX_HIP(hipMemcpyAsync(d_in, rec.seq.data(), nb, hipMemcpyHostToDevice, ctx().stream));
        PGX_HIP(hipMemcpyAsync(d_off, rec.off.data(), ((size_t)n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx().stream));
        hipLaunchKernelGGL(k_encode_biseq, dim3(n), dim3(256), 0, ctx().stream, d_in, d_off, n, d_out);
        PGX_HIP(hipMemcpyAsync(enc.data(), d_out, nb, hipMemcpyDeviceToHost, ctx().stream));
        pgx::sync();
      }
      for (uint32_t i = 0; i < n; ++i) {
        const uint64_t len = rec.off[i + 1] - rec.off[i];
        fprintf(fidx, "%09d %s %u %lu\n", (int)rid, rec.names[i].c_str(), (unsigned)len, (unsigned long)offset);
        ++rid, offset += len;
      }
      rec.clear();
      return !nb || fwrite(enc.data(), 1, nb, fdb) == nb;
    };
    while (rc == PGX_OK && fscanf(lst, "%8191s", fn) == 1) {
      gzFile gz = gzopen(fn, "r");  // gzopen reads plain and gzip files alike, as the reference's gzread does
      if (!gz) {
        set_error("file '%s' open error", fn);
        rc = PGX_EIO;
        break;
      }
      std::vector<uint8_t> buf(std::max<size_t>(PIECE, 16));
      size_t have = 0;
      bool eof = false;
      ParseState st;
      while (!st.stopped) {
        while (!eof && have < buf.size()) {  // fill the piece
          const int got = gzread(gz, buf.data() + have, (unsigned)std::min<size_t>(buf.size() - have, (size_t)1 << 30));
          if (got <= 0) eof = true;
          else have += (size_t)got;
        }
        const size_t used = parse_some(buf.data(), have, eof, st, rec);
        if (rec.seq.size() >= BATCH && !flush()) {
          set_error("short write to %s.seqdb", seqdb_prefix);
          rc = PGX_EIO;
          break;
        }
        if (eof) break;
        if (used == 0 && have == buf.size()) buf.resize(buf.size() * 2);  // a record longer than the piece (a contig): a larger piece
        memmove(buf.data(), buf.data() + used, have - used);
        have -= used;
      }
      gzclose(gz);
    }
    if (rc == PGX_OK && !flush()) {
      set_error("short write to %s.seqdb", seqdb_prefix);
      rc = PGX_EIO;
    }
    fclose(lst), fclose(fidx), fclose(fdb);
    timing_flush();
    if (n_reads) *n_reads = rid;
    if (n_bases) *n_bases = offset;
    return rc;
  } catch (const Fail &f) {
    return f.code;
  } catch (const std::bad_alloc &) {
    set_error("out of host memory");
    return PGX_ENOMEM;
  }
}
