#!/usr/bin/env python3
"""Golden vectors for rows f3 (shmr_map) and f4 (the shimmer4py query helpers), from the REAL reference compiled in place
(oracle/_ref): the tiny dataset of tiny_stage.npz is written back to files, the reference's own code answers, and inputs
+ answers are stored as data.  Run in the build container: python tests/golden/make_golden_query.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as U  # noqa: E402
from peregrine_amd import formats, simreads  # noqa: E402

MP256 = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("y0", "<u8"), ("y1", "<u8"), ("direction", "u1"), ("pad", "u1", 7)])


class PyMmer(C.Structure):  # py_mmer_t, shimmer.h:132-138
    _fields_ = [("mmers", C.c_void_p), ("mmer0_map", C.c_void_p), ("rlmap", C.c_void_p), ("mcmap", C.c_void_p), ("ridmm", C.c_void_p)]


def main():
    assert U.have_ref()
    tiny = np.load(os.path.join(HERE, "tiny_stage.npz"))
    tmp = tempfile.mkdtemp(prefix="goldenq_")
    pre = os.path.join(tmp, "sd")
    open(pre + ".seqdb", "wb").write(tiny["seqdb"].tobytes())
    open(pre + ".idx", "wb").write(tiny["idx_text"].tobytes())
    sp = os.path.join(tmp, "ix-L2")
    for c in (1, 2):   # the two-chunk level-2 index of the fixture
        formats.write_mmlist(f"{sp}-{c:02d}-of-02.dat", tiny[f"ix2l2_L2_{c}"])
        mc = np.zeros(len(tiny[f"ix2l2_L2MC_{c}"]), formats.MC_DTYPE)
        mc["mer"], mc["count"] = tiny[f"ix2l2_L2MC_{c}"][:, 0], tiny[f"ix2l2_L2MC_{c}"][:, 1]
        formats.write_mm_count(f"{sp}-MC-{c:02d}-of-02.dat", mc)
    mmers = np.concatenate([tiny["ix2l2_L2_1"], tiny["ix2l2_L2_2"]])
    store = {}
    lib = U.ref()
    lib.get_mmer_count.restype = C.c_uint32
    keys = np.unique(mmers["x"])
    rng = np.random.default_rng(4)
    absent_keys = (rng.integers(1, 1 << 40, 8).astype(np.uint64) << np.uint64(8)) | np.uint64(16)
    qkeys = np.concatenate([keys, absent_keys])
    store["qkeys"] = qkeys
    rids = np.concatenate([np.unique(mmers["y"] >> np.uint64(32)).astype(np.uint32), np.array([100000, 7777777], np.uint32)])
    store["qrids"] = rids
    for (c, T, lo, hi) in ((1, 1, 2, 240), (1, 2, 2, 240), (2, 2, 2, 240), (1, 1, 1, 3), (2, 3, 2, 30)):
        pm = PyMmer()
        lib.build_shimmer_map4py(C.byref(pm), C.c_char_p(pre.encode()), C.c_char_p(sp.encode()), C.c_uint32(c), C.c_uint32(T),
                                 C.c_uint32(lo), C.c_uint32(hi))
        tag = f"c{c}t{T}lo{lo}hi{hi}"
        mv = U.RefV.from_address(pm.mmers)
        got = np.frombuffer((C.c_uint8 * (mv.n * 16)).from_address(mv.a), formats.MM_DTYPE)
        assert np.array_equal(got, mmers)
        hits, hoff = [], [0]
        for k in qkeys:
            v = U.RefV()
            lib.get_shimmer_hits(C.byref(v), C.byref(pm), C.c_uint64(int(k) >> 8), C.c_uint32(int(k) & 0xFF))
            h = np.frombuffer((C.c_uint8 * (v.n * MP256.itemsize)).from_address(v.a), MP256).copy() if v.n else np.zeros(0, MP256)
            h["pad"] = 0
            hits.append(h)
            hoff.append(hoff[-1] + len(h))
        store[f"hits_{tag}"] = np.concatenate(hits)
        store[f"hoff_{tag}"] = np.array(hoff, np.int64)
        if tag == "c1t1lo2hi240":
            store["counts"] = np.array([lib.get_mmer_count(C.byref(pm), C.c_uint64(int(k) >> 8)) for k in qkeys], np.uint32)
            first, cnt = [], []
            for r in rids:
                v = U.RefV()
                lib.get_shimmers_for_read(C.byref(v), C.byref(pm), C.c_uint32(int(r)))
                cnt.append(v.n)
                first.append((v.a - mv.a) // 16 if v.n else 0)
                assert v.m == v.n
            store["read_first"], store["read_count"] = np.array(first, np.int64), np.array(cnt, np.int64)
        print(tag, "hit records", hoff[-1])

    # ---- f3: contigs cut from the genome the tiny reads were sampled from (one forward, one reverse-complemented) ----
    cfg = dict(simreads.WORKLOADS["tiny"])
    g = simreads.make_genome(cfg["genome_len"], cfg["genome_seed"])
    contigs = [g[:30000], (3 - g[24000:])[::-1], g[10000:10900]]

    def enc(codes):
        codes = np.asarray(codes, np.uint8)
        return ((np.uint8(1) << codes) | ((np.uint8(8) >> codes[::-1]) << np.uint8(4))).astype(np.uint8)

    rlen = np.array([len(c) for c in contigs], np.uint32)
    ref_db = formats.SeqDB(np.concatenate([enc(c) for c in contigs]), np.arange(len(contigs), dtype=np.uint32), rlen,
                           np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64),
                           [f"ctg{i}" for i in range(len(contigs))])
    rpre = os.path.join(tmp, "ref")
    formats.write_seqdb(rpre, ref_db)
    U.ref_run("shmr_index", "-p", rpre, "-t", 1, "-c", 1, "-l", 2, "-r", 6, "-o", os.path.join(tmp, "ref"))
    ref_l2 = formats.read_mmlist(os.path.join(tmp, "ref-L2-01-of-01.dat"))
    store["ref_rlen"], store["ref_l2"] = rlen, ref_l2
    for (c, T, lo, hi) in ((1, 1, 1, 240), (2, 2, 1, 240), (1, 1, 2, 4)):
        out = subprocess.run([os.path.join(U.REF_DIR, "shmr_map"), "-r", rpre, "-m", os.path.join(tmp, "ref-L2"), "-p", pre, "-l", sp,
                              "-t", str(T), "-c", str(c), "-n", str(lo), "-M", str(hi)], check=True, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL).stdout
        store[f"map_c{c}t{T}lo{lo}hi{hi}"] = np.frombuffer(out, np.uint8)
        print("shmr_map", c, T, lo, hi, out.count(b"\n"), "lines")
    np.savez_compressed(os.path.join(HERE, "query_cases.npz"), **store)
    print(os.path.getsize(os.path.join(HERE, "query_cases.npz")), "bytes")


if __name__ == "__main__":
    main()
