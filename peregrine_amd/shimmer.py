"""Host-side mirror of the reference interface for the SHIMMER index + overlap path.

`shmr_index(...)` / `shmr_overlap(...)` take the same options as the reference executables
(/root/reference/src/shmr_index.c:64-114, src/shmr_overlap.c:271-326) and produce the same files; the
`ResidentDB` class is the HBM-resident form used by bench.py and the multi-GPU driver.  All compute happens in
libpgx.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .formats import MC_DTYPE, MM_DTYPE, OVLP_DTYPE, SeqDB


@dataclass
class IndexOut:
    top: np.ndarray            # L1 or L2 minimizers (mm128)
    top_mc: np.ndarray         # (mer, count), sorted by mer
    l0: np.ndarray | None
    l0_mc: np.ndarray | None
    bases: int
    reads: int
    reads_literal: int
    ms: float


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class ResidentDB:
    """A read database uploaded once to HBM (pgx_seqdb)."""

    def __init__(self, db: SeqDB, device: int | None = None):
        _lib.init(device)
        self._lib = _lib.load()
        self.h = C.c_void_p()
        seq = np.ascontiguousarray(db.seqdb, np.uint8)
        rid = np.ascontiguousarray(db.rid, np.uint32)
        rlen = np.ascontiguousarray(db.rlen, np.uint32)
        roff = np.ascontiguousarray(db.roff, np.uint64)
        _lib.check(self._lib.pgx_seqdb_upload(_ptr(seq), seq.size, _ptr(rid), _ptr(rlen), _ptr(roff), len(rid),
                                              C.byref(self.h)), "pgx_seqdb_upload")
        self.n_reads, self.n_bases = len(rid), int(rlen.sum(dtype=np.uint64))

    @classmethod
    def from_device(cls, d_seqdb: int, nbytes: int, rid, rlen, roff, device: int | None = None):
        """the seqdb bytes are already in HBM at device pointer d_seqdb (e.g. all-gathered by the ranks of a multi-GPU job)"""
        self = cls.__new__(cls)
        _lib.init(device)
        self._lib = _lib.load()
        self.h = C.c_void_p()
        rid = np.ascontiguousarray(rid, np.uint32)
        rlen = np.ascontiguousarray(rlen, np.uint32)
        roff = np.ascontiguousarray(roff, np.uint64)
        _lib.check(self._lib.pgx_seqdb_upload_dev(C.c_void_p(d_seqdb), nbytes, _ptr(rid), _ptr(rlen), _ptr(roff), len(rid),
                                                  C.byref(self.h)), "pgx_seqdb_upload_dev")
        self.n_reads, self.n_bases = len(rid), int(rlen.sum(dtype=np.uint64))
        return self

    @classmethod
    def adopt_device(cls, seq_tensor, nbytes: int, rid, rlen, roff, device: int | None = None):
        """NO copy: the library reads the seqdb where it is -- `seq_tensor` is a torch uint8 device tensor of at least
        nbytes + 1024 elements (e.g. the buffer the ranks of a multi-GPU job all-gathered their read sets into); the object
        keeps it alive.  One copy of the job's seqdb per GPU (pgx_seqdb_adopt_dev)."""
        self = cls.__new__(cls)
        _lib.init(device)
        self._lib = _lib.load()
        self.h = C.c_void_p()
        rid = np.ascontiguousarray(rid, np.uint32)
        rlen = np.ascontiguousarray(rlen, np.uint32)
        roff = np.ascontiguousarray(roff, np.uint64)
        assert seq_tensor.is_contiguous() and seq_tensor.element_size() == 1
        _lib.stream_wait()   # whatever filled the buffer on torch's stream comes first
        _lib.check(self._lib.pgx_seqdb_adopt_dev(C.c_void_p(seq_tensor.data_ptr()), int(nbytes), int(seq_tensor.numel()), _ptr(rid),
                                                 _ptr(rlen), _ptr(roff), len(rid), C.byref(self.h)), "pgx_seqdb_adopt_dev")
        self._adopted = seq_tensor
        self.n_reads, self.n_bases = len(rid), int(rlen.sum(dtype=np.uint64))
        return self

    # ---- multi-GPU hand-over on device pointers (include/pgx.h) ------------------------------------------
    def index_dev(self, total_chunk=1, mychunk=1, levels=2, reduction=6, window=80, kmer=16):
        """index stage; returns (IndexOut without arrays, d_top, n_top, d_mc, n_mc): list and counts stay in HBM (library-owned)"""
        p = _lib.IndexParams(total_chunk, mychunk, levels, reduction, window, kmer, 0)
        r = _lib.IndexResult()
        d_top, n_top, d_mc, n_mc = C.c_void_p(), C.c_size_t(0), C.c_void_p(), C.c_size_t(0)
        _lib.check(self._lib.pgx_index_resident_dev(self.h, C.byref(p), C.byref(r), C.byref(d_top), C.byref(n_top), C.byref(d_mc),
                                                    C.byref(n_mc)), "pgx_index_resident_dev")
        ix = IndexOut(top=None, top_mc=None, l0=None, l0_mc=None, bases=int(r.bases), reads=int(r.reads),
                      reads_literal=int(r.reads_literal), ms=float(r.gpu_ms))
        return ix, int(d_top.value or 0), int(n_top.value), int(d_mc.value or 0), int(n_mc.value)

    def pairs_prepare_dev(self, d_top: int, n_top: int, d_counts_all: int, n_counts_all: int, mc_lower=2, mc_upper=240) -> int:
        first = C.c_int64(-1)
        _lib.check(self._lib.pgx_pairs_prepare_dev(self.h, C.c_void_p(d_top), n_top, C.c_void_p(d_counts_all), n_counts_all,
                                                   mc_lower, mc_upper, C.byref(first)), "pgx_pairs_prepare_dev")
        return int(first.value)

    def pairs_scatter_dev(self, total_chunk: int, start: int):
        """returns (device pointer of the send buffer, records per destination chunk 1..total_chunk)"""
        d_send = C.c_void_p()
        counts = np.zeros(total_chunk, np.uint64)
        _lib.check(self._lib.pgx_pairs_scatter_dev(self.h, total_chunk, start, C.byref(d_send), _ptr(counts)), "pgx_pairs_scatter_dev")
        return int(d_send.value or 0), counts

    def overlap_records_dev(self, d_records: int, n_records: int, total_chunk=1, mychunk=1, bestn=4, mc_lower=2, mc_upper=240,
                            align_bandwidth=100, ovlp_upper=120):
        p = _lib.OverlapParams(total_chunk, mychunk, bestn, mc_lower, mc_upper, align_bandwidth, ovlp_upper)
        out, n, st = C.c_void_p(), C.c_size_t(0), _lib.OverlapStats()
        _lib.check(self._lib.pgx_overlap_records_dev(self.h, C.c_void_p(d_records), n_records, C.byref(p), C.byref(out), C.byref(n),
                                                     C.byref(st)), "pgx_overlap_records_dev")
        return _lib.take(out.value, n.value, OVLP_DTYPE), st.asdict()

    def overlap_dev(self, d_mmers: int, n_mm: int, d_counts: int, n_counts: int, total_chunk=1, mychunk=1, bestn=4, mc_lower=2,
                    mc_upper=240, align_bandwidth=100, ovlp_upper=120):
        p = _lib.OverlapParams(total_chunk, mychunk, bestn, mc_lower, mc_upper, align_bandwidth, ovlp_upper)
        out, n, st = C.c_void_p(), C.c_size_t(0), _lib.OverlapStats()
        _lib.check(self._lib.pgx_overlap_resident_dev(self.h, C.c_void_p(d_mmers), n_mm, C.c_void_p(d_counts), n_counts, C.byref(p),
                                                      C.byref(out), C.byref(n), C.byref(st)), "pgx_overlap_resident_dev")
        return _lib.take(out.value, n.value, OVLP_DTYPE), st.asdict()

    def release_bytes(self) -> bool:
        """The seqdb's BYTES out of HBM (pgx_seqdb_release_bytes): the 2-bit packs carry the same information at a quarter of the size and
        the default path's kernels read them.  Returns False -- bytes kept, nothing changed -- for a database with an ambiguous base or a
        read beyond 65,535 bases; True: an adopted device buffer is no longer referenced (this object drops its hold on it)."""
        rc = self._lib.pgx_seqdb_release_bytes(self.h)
        if rc == _lib.PGX_ESTATE:
            return False
        _lib.check(rc, "pgx_seqdb_release_bytes")
        self._adopted = None
        return True

    @property
    def has_bytes(self) -> bool:
        return bool(self._lib.pgx_seqdb_has_bytes(self.h))

    def close(self):
        if self.h:
            self._lib.pgx_seqdb_free(self.h)
            self.h = C.c_void_p()
        self._adopted = None   # (the caller's device buffer is the caller's again)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stages ------------------------------------------------------------------------------------------
    def index(self, total_chunk=1, mychunk=1, levels=2, reduction=6, window=80, kmer=16, want_l0=False) -> IndexOut:
        p = _lib.IndexParams(total_chunk, mychunk, levels, reduction, window, kmer, 1 if want_l0 else 0)
        r = _lib.IndexResult()
        _lib.check(self._lib.pgx_index_resident(self.h, C.byref(p), C.byref(r)), "pgx_index_resident")
        return IndexOut(
            top=_lib.take(r.top, r.n_top, MM_DTYPE), top_mc=_lib.take(r.top_mc, r.n_top_mc, MC_DTYPE),
            l0=_lib.take(r.l0, r.n_l0, MM_DTYPE) if want_l0 else None,
            l0_mc=_lib.take(r.l0_mc, r.n_l0_mc, MC_DTYPE) if want_l0 else None,
            bases=int(r.bases), reads=int(r.reads), reads_literal=int(r.reads_literal), ms=float(r.gpu_ms))

    def overlap(self, mmers: np.ndarray, counts: np.ndarray, total_chunk=1, mychunk=1, bestn=4, mc_lower=2,
                mc_upper=240, align_bandwidth=100, ovlp_upper=120):
        mm = np.ascontiguousarray(mmers, MM_DTYPE)
        mc = np.ascontiguousarray(counts, MC_DTYPE)
        p = _lib.OverlapParams(total_chunk, mychunk, bestn, mc_lower, mc_upper, align_bandwidth, ovlp_upper)
        out, n, st = C.c_void_p(), C.c_size_t(0), _lib.OverlapStats()
        _lib.check(self._lib.pgx_overlap_resident(self.h, _ptr(mm), len(mm), _ptr(mc), len(mc), C.byref(p),
                                                  C.byref(out), C.byref(n), C.byref(st)), "pgx_overlap_resident")
        return _lib.take(out.value, n.value, OVLP_DTYPE), st.asdict()

    def index_overlap(self, want_index_arrays=False, levels=2, reduction=6, window=80, kmer=16, bestn=4, mc_lower=2,
                      mc_upper=240, align_bandwidth=100, ovlp_upper=120):
        """single-chunk index + overlap in one call; the shimmer list and its counts never leave HBM between the stages.
        Returns (IndexOut — arrays None unless want_index_arrays —, ovlp records, stats)."""
        ip = _lib.IndexParams(1, 1, levels, reduction, window, kmer, 0)
        op = _lib.OverlapParams(1, 1, bestn, mc_lower, mc_upper, align_bandwidth, ovlp_upper)
        r, out, n, st = _lib.IndexResult(), C.c_void_p(), C.c_size_t(0), _lib.OverlapStats()
        _lib.check(self._lib.pgx_index_overlap_resident(self.h, C.byref(ip), C.byref(op), 1 if want_index_arrays else 0,
                                                        C.byref(r), C.byref(out), C.byref(n), C.byref(st)),
                   "pgx_index_overlap_resident")
        ix = IndexOut(top=_lib.take(r.top, r.n_top, MM_DTYPE) if r.top else None,
                      top_mc=_lib.take(r.top_mc, r.n_top_mc, MC_DTYPE) if r.top_mc else None, l0=None, l0_mc=None,
                      bases=int(r.bases), reads=int(r.reads), reads_literal=int(r.reads_literal), ms=float(r.gpu_ms))
        return ix, _lib.take(out.value, n.value, OVLP_DTYPE), st.asdict()

    # ---- batch level -------------------------------------------------------------------------------------
    def sketch(self, read_slots, w=80, k=16) -> np.ndarray:
        slots = np.ascontiguousarray(read_slots, np.uint32)
        out, n = C.c_void_p(), C.c_size_t(0)
        _lib.check(self._lib.pgx_sketch_batch(self.h, _ptr(slots), len(slots), w, k, C.byref(out), C.byref(n)),
                   "pgx_sketch_batch")
        return _lib.take(out.value, n.value, MM_DTYPE)

    def align(self, keys: np.ndarray, band=100) -> np.ndarray:
        keys = np.ascontiguousarray(keys, _lib.ALIGN_KEY_DTYPE)
        out = np.zeros(len(keys), _lib.MATCH_DTYPE)
        _lib.check(self._lib.pgx_align_batch(self.h, _ptr(keys), len(keys), band, _ptr(out)), "pgx_align_batch")
        return out


def mm_reduce(mm: np.ndarray, rs: int) -> np.ndarray:
    """GPU mm_reduce over an arbitrary multi-read list (src/shmr_reduce.c:53-90)."""
    _lib.init()
    mm = np.ascontiguousarray(mm, MM_DTYPE)
    out, n = C.c_void_p(), C.c_size_t(0)
    _lib.check(_lib.load().pgx_reduce_batch(_ptr(mm), len(mm), rs, C.byref(out), C.byref(n)), "pgx_reduce_batch")
    return _lib.take(out.value, n.value, MM_DTYPE)


def mm_count(mm: np.ndarray) -> np.ndarray:
    """GPU mm_count (src/shmr_utils.c:131-160); sorted by mer."""
    _lib.init()
    mm = np.ascontiguousarray(mm, MM_DTYPE)
    out, n = C.c_void_p(), C.c_size_t(0)
    _lib.check(_lib.load().pgx_count_batch(_ptr(mm), len(mm), C.byref(out), C.byref(n)), "pgx_count_batch")
    return _lib.take(out.value, n.value, MC_DTYPE)


# ---- file-level stages: drop-ins for the two executables -------------------------------------------------------
def shmr_index(seqdb_prefix: str, out_prefix: str = "shimmer", total_chunk=1, mychunk=1, levels=2, reduction=6,
               write_l0=1, window=80, kmer=16, device=None) -> dict:
    """shmr_index -p -o -t -c -l -r -m -w -k   (defaults of src/shmr_index.c:21-23,49-55)."""
    _lib.init(device)
    p = _lib.IndexParams(total_chunk, mychunk, levels, reduction, window, kmer, write_l0)
    r = _lib.IndexResult()
    _lib.check(_lib.load().pgx_index_chunk(seqdb_prefix.encode(), out_prefix.encode(), C.byref(p), C.byref(r)),
               "pgx_index_chunk")
    return dict(bases=int(r.bases), reads=int(r.reads), reads_literal=int(r.reads_literal), ms=float(r.gpu_ms))


def shmr_overlap(seqdb_prefix: str, shimmer_prefix: str, out_path: str | None = None, total_chunk=1, mychunk=1,
                 bestn=4, mc_lower=2, mc_upper=240, align_bandwidth=100, ovlp_upper=120, device=None) -> dict:
    """shmr_overlap -p -l -t -c -b -m -M -w -n -o   (defaults of src/shmr_overlap.c:28-42,245-251,341-344)."""
    _lib.init(device)
    if out_path is None:
        out_path = "ovlp.%02d" % mychunk
    p = _lib.OverlapParams(total_chunk, mychunk, bestn, mc_lower, mc_upper, align_bandwidth, ovlp_upper)
    st = _lib.OverlapStats()
    _lib.check(_lib.load().pgx_overlap_chunk(seqdb_prefix.encode(), shimmer_prefix.encode(), out_path.encode(),
                                             C.byref(p), C.byref(st)), "pgx_overlap_chunk")
    return st.asdict()


def shmr_mkseqdb(seq_dataset_path: str = "seq_dataset.lst", seqdb_prefix: str = "seq_dataset", device=None) -> dict:
    """shmr_mkseqdb -d -p   (defaults of src/shmr_mkseqdb.c:61-69): FASTA/FASTQ(.gz) list -> <prefix>.seqdb + <prefix>.idx."""
    _lib.init(device)
    nr, nb = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.load().pgx_mkseqdb(seq_dataset_path.encode(), seqdb_prefix.encode(), C.byref(nr), C.byref(nb)),
               "pgx_mkseqdb")
    return dict(reads=int(nr.value), bases=int(nb.value))


def shmr_dedup(ovlp_paths, out_path: str | None = None, device=None):
    """cat ovlp*.dat | shmr_dedup > preads.ovl (pg_run.py:351-352): returns the text (bytes) and the number of unique pairs."""
    _lib.init(device)
    if isinstance(ovlp_paths, (str, bytes)):
        ovlp_paths = [ovlp_paths]
    recs = np.concatenate([np.fromfile(p, dtype=OVLP_DTYPE) for p in ovlp_paths]) if ovlp_paths else np.zeros(0, OVLP_DTYPE)
    recs = np.ascontiguousarray(recs)
    text, tl, nu = C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
    _lib.check(_lib.load().pgx_dedup(_ptr(recs), len(recs), C.byref(text), C.byref(tl), C.byref(nu)), "pgx_dedup")
    data = C.string_at(text.value, tl.value)
    _lib.load().pgx_free(text)
    if out_path:
        with open(out_path, "wb") as f:
            f.write(data)
    return data, int(nu.value)


def shmr_map(ref_shimmer_prefix: str = "ref-L2", seqdb_prefix: str = "seq_dataset", shimmer_prefix: str = "shimmer-L2",
             refdb_prefix: str = "ref", total_chunk=1, mychunk=1, mc_lower=1, mc_upper=240, out_path: str | None = None, device=None):
    """shmr_map -r -m -p -l -t -c -n -M (src/shmr_map.c:163-373): the reads' shimmer pairs located on the contigs.  Returns the
    reference's stdout text (bytes) and the number of lines."""
    _lib.init(device)
    p = _lib.MapParams(total_chunk, mychunk, mc_lower, mc_upper)
    text, tl, nl = C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
    _lib.check(_lib.load().pgx_map_chunk(refdb_prefix.encode(), ref_shimmer_prefix.encode(), seqdb_prefix.encode(),
                                         shimmer_prefix.encode(), C.byref(p), C.byref(text), C.byref(tl), C.byref(nl)), "pgx_map_chunk")
    data = C.string_at(text.value, tl.value)
    _lib.load().pgx_free(text)
    if out_path:
        with open(out_path, "wb") as f:
            f.write(data)
    return data, int(nl.value)


def map_reads_to_ref(ref_mmers, mmers, counts, rlen_by_rid, total_chunk=1, mychunk=1, mc_lower=1, mc_upper=240, device=None):
    """in-memory form of shmr_map (pgx_map): arrays in, text out"""
    _lib.init(device)
    rf = np.ascontiguousarray(ref_mmers, MM_DTYPE)
    mm = np.ascontiguousarray(mmers, MM_DTYPE)
    mc = np.ascontiguousarray(counts, MC_DTYPE)
    rl = np.ascontiguousarray(rlen_by_rid, np.uint32)
    p = _lib.MapParams(total_chunk, mychunk, mc_lower, mc_upper)
    text, tl, nl = C.c_void_p(), C.c_size_t(0), C.c_uint64(0)
    _lib.check(_lib.load().pgx_map(_ptr(rf), len(rf), _ptr(mm), len(mm), _ptr(mc), len(mc), _ptr(rl), len(rl), C.byref(p),
                                   C.byref(text), C.byref(tl), C.byref(nl)), "pgx_map")
    data = C.string_at(text.value, tl.value)
    _lib.load().pgx_free(text)
    return data, int(nl.value)


class ShimmerMap:
    """The shimmer4py query object (py_mmer_t + build_shimmer_map4py / get_* of src/shimmer4py.c:44-196), as the reference's
    notebooks and py/peregrine/utils.py use it: built once on the GPU, queried from the host."""

    def __init__(self, seqdb_prefix: str, shimmer_prefix: str, mychunk=1, total_chunk=1, lowerbound=2, upperbound=240, device=None):
        _lib.init(device)
        self._lib = _lib.load()
        self.h = _lib.PyMmer()
        self._lib.build_shimmer_map4py(C.byref(self.h), seqdb_prefix.encode(), shimmer_prefix.encode(), mychunk, total_chunk,
                                       lowerbound, upperbound)
        if not self.h.mmer0_map:
            raise _lib.PgxError("build_shimmer_map4py failed: " + self._lib.pgx_last_error().decode(errors="replace"))

    @property
    def mmers(self) -> np.ndarray:
        v = self.h.mmers.contents
        return np.frombuffer((C.c_uint8 * (v.n * 16)).from_address(v.a), MM_DTYPE) if v.n else np.zeros(0, MM_DTYPE)

    def shimmers_for_read(self, rid: int) -> np.ndarray:
        v = _lib.KVec()
        self._lib.get_shimmers_for_read(C.byref(v), C.byref(self.h), int(rid))
        return np.frombuffer((C.c_uint8 * (v.n * 16)).from_address(v.a), MM_DTYPE).copy() if v.n else np.zeros(0, MM_DTYPE)

    def read_range(self, rid: int):
        """(first index, count) of the read's run inside .mmers"""
        v = _lib.KVec()
        self._lib.get_shimmers_for_read(C.byref(v), C.byref(self.h), int(rid))
        return ((v.a - self.h.mmers.contents.a) // 16 if v.n else 0), int(v.n)

    def mmer_count(self, mhash: int) -> int:
        return int(self._lib.get_mmer_count(C.byref(self.h), int(mhash)))

    def hits(self, mhash0: int, span: int) -> np.ndarray:
        v = _lib.KVec()
        self._lib.get_shimmer_hits(C.byref(v), C.byref(self.h), int(mhash0), int(span))
        out = np.frombuffer(C.string_at(v.a, v.n * _lib.MP256_DTYPE.itemsize), _lib.MP256_DTYPE).copy() if v.n else np.zeros(0, _lib.MP256_DTYPE)
        if v.a:
            self._lib.pgx_free(C.c_void_p(v.a))
        return out

    def close(self):
        if self.h.mmer0_map:
            self._lib.pgx_shimmer_map_free(C.byref(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
