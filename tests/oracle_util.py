"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when present, to the real reference compiled
in place into oracle/_ref/.  TEST INFRASTRUCTURE: imported only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke()."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from peregrine_amd.formats import MM_DTYPE, MC_DTYPE, OVLP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


class MMV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("cap", C.c_size_t), ("a", C.c_void_p)]


class Match(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("m_size", "dist", "q_bgn", "q_end", "t_bgn", "t_end", "t_m_end", "q_m_end")]

    def astuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class Stats(C.Structure):
    _fields_ = [(f, C.c_uint64) for f in ("n_align", "n_seen_skip", "n_buckets", "n_records", "bases_cmp")]


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"])
        lib = C.CDLL(so)
        lib.orc_hash64.restype = C.c_uint64
        lib.orc_hash64.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_khash_order.restype = C.c_size_t
        lib.orc_free.argtypes = [C.c_void_p]
        _oracle = lib
    return _oracle


def _take(v: MMV, dtype) -> np.ndarray:
    n = int(v.n)
    if n == 0:
        out = np.zeros(0, dtype)
    else:
        buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(v.a)
        out = np.frombuffer(buf, dtype=dtype).copy()
    if v.a:
        oracle().orc_free(C.c_void_p(v.a))
    return out


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def orc_sketch_seqdb(read_bytes, w, k, rid) -> np.ndarray:
    b = _u8(read_bytes)
    v = MMV()
    oracle().orc_sketch_seqdb(b.ctypes.data_as(C.c_void_p), C.c_int(len(b)), C.c_int(w), C.c_int(k), C.c_uint32(rid), C.byref(v))
    return _take(v, MM_DTYPE)


def orc_sketch_ascii(seq: bytes, w, k, rid) -> np.ndarray:
    v = MMV()
    oracle().orc_sketch_ascii(C.c_char_p(seq), C.c_int(len(seq)), C.c_int(w), C.c_int(k), C.c_uint32(rid), C.byref(v))
    return _take(v, MM_DTYPE)


def _as_mmv(arr):
    arr = np.ascontiguousarray(arr, MM_DTYPE)
    v = MMV(len(arr), len(arr), arr.ctypes.data)
    return v, arr


def orc_reduce(mm, rs) -> np.ndarray:
    vin, keep = _as_mmv(mm)
    v = MMV()
    oracle().orc_reduce(C.byref(vin), C.byref(v), C.c_uint8(rs))
    return _take(v, MM_DTYPE)


def orc_count(mm) -> np.ndarray:
    vin, keep = _as_mmv(mm)
    v = MMV()
    oracle().orc_count(C.byref(vin), C.byref(v))
    return _take(v, MC_DTYPE)


def orc_ovlp_match(q, q_strand, t, t_strand, band):
    q = _u8(q); t = _u8(t)
    m = Match()
    oracle().orc_ovlp_match(q.ctypes.data_as(C.c_void_p), C.c_int32(len(q)), C.c_uint8(q_strand),
                            t.ctypes.data_as(C.c_void_p), C.c_int32(len(t)), C.c_uint8(t_strand), C.c_int32(band),
                            C.byref(m), None)
    return m.astuple()


def orc_khash_order(keys) -> np.ndarray:
    keys = np.ascontiguousarray(keys, np.uint64)
    out = np.zeros(len(keys), np.uint64)
    n = oracle().orc_khash_order(keys.ctypes.data_as(C.c_void_p), C.c_size_t(len(keys)), out.ctypes.data_as(C.c_void_p))
    return out[:n]


def orc_overlap(db, mmers, counts, mychunk=1, total=1, mc_lower=2, mc_upper=240, bestn=4, ovlp_upper=120, band=100):
    rl, ro = db.by_rid()
    seq = _u8(db.seqdb)
    mm = np.ascontiguousarray(mmers, MM_DTYPE)
    mc = np.ascontiguousarray(counts, MC_DTYPE)
    v = MMV()
    st = Stats()
    oracle().orc_overlap(seq.ctypes.data_as(C.c_void_p), rl.ctypes.data_as(C.c_void_p), ro.ctypes.data_as(C.c_void_p),
                         C.c_uint32(len(rl)), mm.ctypes.data_as(C.c_void_p), C.c_size_t(len(mm)),
                         mc.ctypes.data_as(C.c_void_p), C.c_size_t(len(mc)), C.c_uint32(mychunk), C.c_uint32(total),
                         C.c_uint32(mc_lower), C.c_uint32(mc_upper), C.c_uint32(bestn), C.c_uint32(ovlp_upper),
                         C.c_uint32(band), C.byref(v), C.byref(st))
    return _take(v, OVLP_DTYPE), {f: getattr(st, f) for f, _ in Stats._fields_}


def orc_index_chunk(seqdb_prefix, out_prefix, total=1, mychunk=1, levels=2, reduction=6, write_l0=1, w=80, k=16):
    bases = C.c_uint64(0)
    rc = oracle().orc_index_chunk(seqdb_prefix.encode(), out_prefix.encode(), total, mychunk, levels, reduction,
                                  write_l0, w, k, C.byref(bases))
    assert rc == 0
    return int(bases.value)


def orc_overlap_chunk(seqdb_prefix, shimmer_prefix, out_path, total=1, mychunk=1, bestn=4, mc_lower=2, mc_upper=240,
                      band=100, ovlp_upper=120):
    st = Stats()
    n = C.c_uint64(0)
    rc = oracle().orc_overlap_chunk(seqdb_prefix.encode(), shimmer_prefix.encode(), out_path.encode(), total, mychunk,
                                    bestn, mc_lower, mc_upper, band, ovlp_upper, C.byref(st), C.byref(n))
    assert rc == 0
    return int(n.value), {f: getattr(st, f) for f, _ in Stats._fields_}


def orc_dedup(recs):
    recs = np.ascontiguousarray(recs, OVLP_DTYPE)
    tl, nu = C.c_size_t(0), C.c_uint64(0)
    fn = oracle().orc_dedup
    fn.restype = C.c_void_p
    p = fn(recs.ctypes.data_as(C.c_void_p), C.c_size_t(len(recs)), C.byref(tl), C.byref(nu))
    data = C.string_at(p, tl.value)
    oracle().orc_free(C.c_void_p(p))
    return data, int(nu.value)


MP256_DTYPE = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("y0", "<u8"), ("y1", "<u8"), ("direction", "u1"), ("pad", "u1", 7)])


class OrcMap:
    """rows f4: the oracle's pair map + query helpers (orc_map_*)"""

    def __init__(self, mmers, counts, rlen_by_rid, mychunk=1, total=1, lower=2, upper=240):
        self.mm = np.ascontiguousarray(mmers, MM_DTYPE)
        mc = np.ascontiguousarray(counts, MC_DTYPE)
        rl = np.ascontiguousarray(rlen_by_rid, np.uint32)
        fn = oracle().orc_map_build
        fn.restype = C.c_void_p
        self.h = fn(self.mm.ctypes.data_as(C.c_void_p), C.c_size_t(len(self.mm)), mc.ctypes.data_as(C.c_void_p), C.c_size_t(len(mc)),
                    rl.ctypes.data_as(C.c_void_p), C.c_uint32(mychunk), C.c_uint32(total), C.c_uint32(lower), C.c_uint32(upper))

    def count(self, mhash):
        fn = oracle().orc_map_count
        fn.restype = C.c_uint32
        return int(fn(C.c_void_p(self.h), C.c_uint64(int(mhash))))

    def hits(self, mhash0, span):
        fn = oracle().orc_map_hits
        fn.restype = C.c_size_t
        p = C.c_void_p()
        n = fn(C.c_void_p(self.h), C.c_uint64(int(mhash0)), C.c_uint32(int(span)), C.byref(p))
        out = np.frombuffer(C.string_at(p, n * MP256_DTYPE.itemsize), MP256_DTYPE).copy() if n else np.zeros(0, MP256_DTYPE)
        if p:
            oracle().orc_free(p)
        return out

    def read_shimmers(self, rid):
        f, c = C.c_size_t(0), C.c_size_t(0)
        oracle().orc_read_shimmers(self.mm.ctypes.data_as(C.c_void_p), C.c_size_t(len(self.mm)), C.c_uint32(int(rid)), C.byref(f), C.byref(c))
        return int(f.value), int(c.value)

    def close(self):
        if self.h:
            oracle().orc_map_free(C.c_void_p(self.h))
            self.h = None


def orc_map_reads_to_ref(ref_mmers, mmers, counts, rlen_by_rid, mychunk=1, total=1, lower=1, upper=240):
    """row f3: the oracle's shmr_map text"""
    rf = np.ascontiguousarray(ref_mmers, MM_DTYPE)
    mm = np.ascontiguousarray(mmers, MM_DTYPE)
    mc = np.ascontiguousarray(counts, MC_DTYPE)
    rl = np.ascontiguousarray(rlen_by_rid, np.uint32)
    tl, nl = C.c_size_t(0), C.c_uint64(0)
    fn = oracle().orc_map_reads_to_ref
    fn.restype = C.c_void_p
    p = fn(rf.ctypes.data_as(C.c_void_p), C.c_size_t(len(rf)), mm.ctypes.data_as(C.c_void_p), C.c_size_t(len(mm)),
           mc.ctypes.data_as(C.c_void_p), C.c_size_t(len(mc)), rl.ctypes.data_as(C.c_void_p), C.c_uint32(mychunk), C.c_uint32(total),
           C.c_uint32(lower), C.c_uint32(upper), C.byref(tl), C.byref(nl))
    data = C.string_at(p, tl.value)
    oracle().orc_free(C.c_void_p(p))
    return data, int(nl.value)


# ------------------------------------------------------------------------------------------------------------
# the real reference (oracle/_ref): present in the build container, prebuilt binaries on the GPU box
# ------------------------------------------------------------------------------------------------------------
def have_ref() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, f)) for f in ("libshimmer_ref.so", "shmr_index", "shmr_overlap"))


class RefV(C.Structure):  # kvec: {size_t n, m; T* a}
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


_ref = None


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(REF_DIR, "libshimmer_ref.so"))
        lib.ovlp_match.restype = C.POINTER(Match)
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        lib._libc = libc
        _ref = lib
    return _ref


def _take_ref(v: RefV, dtype):
    n = int(v.n)
    out = np.frombuffer((C.c_uint8 * (n * dtype.itemsize)).from_address(v.a), dtype=dtype).copy() if n else np.zeros(0, dtype)
    if v.a:
        ref()._libc.free(C.c_void_p(v.a))
    return out


def ref_sketch_ascii(seq: bytes, w, k, rid) -> np.ndarray:
    v = RefV()
    ref().mm_sketch(None, C.c_char_p(seq), C.c_int(len(seq)), C.c_int(w), C.c_int(k), C.c_uint32(rid), C.c_int(0), C.byref(v))
    return _take_ref(v, MM_DTYPE)


def ref_reduce(mm, rs) -> np.ndarray:
    arr = np.ascontiguousarray(mm, MM_DTYPE)
    vin = RefV(len(arr), len(arr), arr.ctypes.data)
    v = RefV()
    ref().mm_reduce(C.byref(vin), C.byref(v), C.c_uint8(rs))
    return _take_ref(v, MM_DTYPE)


def ref_ovlp_match(q, q_strand, t, t_strand, band):
    q = _u8(q); t = _u8(t)
    p = ref().ovlp_match(q.ctypes.data_as(C.c_void_p), C.c_int32(len(q)), C.c_uint8(q_strand),
                         t.ctypes.data_as(C.c_void_p), C.c_int32(len(t)), C.c_uint8(t_strand), C.c_int32(band))
    out = p.contents.astuple()
    ref().free_ovlp_match(p)
    return out


def ref_encode(seq: bytes) -> np.ndarray:
    out = np.zeros(len(seq), np.uint8)
    ref().encode_biseq(out.ctypes.data_as(C.c_void_p), C.c_char_p(seq), C.c_size_t(len(seq)))
    return out


def ref_decode(b, strand) -> bytes:
    b = _u8(b)
    out = C.create_string_buffer(len(b))
    ref().decode_biseq(b.ctypes.data_as(C.c_void_p), out, C.c_size_t(len(b)), C.c_uint8(strand))
    return out.raw


def ref_run(tool, *args, cwd=None):
    return subprocess.run([os.path.join(REF_DIR, tool), *map(str, args)], cwd=cwd, check=True,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


PAIR_REC_DTYPE = np.dtype([("key0", "<u8"), ("key1", "<u8"), ("y0", "<u8"), ("npos", "<u4"), ("dir", "u1"), ("pad", "u1", 3)])
assert PAIR_REC_DTYPE.itemsize == 32


def orc_pair_records(mmers, counts, rlen_by_rid, mychunk=1, total=1, mc_lower=2, mc_upper=240) -> np.ndarray:
    """the insertion sequence of build_map for overlap chunk `mychunk` of `total` (what the record exchange must deliver)"""
    mm = np.ascontiguousarray(mmers, MM_DTYPE)
    mc = np.ascontiguousarray(counts, MC_DTYPE)
    rl = np.ascontiguousarray(rlen_by_rid, np.uint32)
    n = C.c_size_t(0)
    fn = oracle().orc_pair_records
    fn.restype = C.c_void_p
    p = fn(mm.ctypes.data_as(C.c_void_p), C.c_size_t(len(mm)), mc.ctypes.data_as(C.c_void_p), C.c_size_t(len(mc)),
           rl.ctypes.data_as(C.c_void_p), C.c_uint32(mychunk), C.c_uint32(total), C.c_uint32(mc_lower), C.c_uint32(mc_upper), C.byref(n))
    out = np.frombuffer(C.string_at(p, n.value * 32), PAIR_REC_DTYPE).copy() if n.value else np.zeros(0, PAIR_REC_DTYPE)
    oracle().orc_free(C.c_void_p(p))
    return out
