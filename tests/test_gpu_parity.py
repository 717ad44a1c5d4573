"""GPU parity tests proper: every call goes through the C-ABI of libpgx.so and is compared bit-for-bit with
(1) the committed outputs of the real reference (tests/golden) and (2) the CPU oracle on seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_util as G
import oracle_util as U
from peregrine_amd import _lib, formats, simreads
from peregrine_amd.shimmer import ResidentDB, mm_count, mm_reduce, shmr_index, shmr_overlap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    _lib.init(0)
    return _lib.load()


class KV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


def _take_kv(v, libc=C.CDLL(None)):
    n = int(v.n)
    out = np.frombuffer((C.c_uint8 * (n * 16)).from_address(v.a), dtype=formats.MM_DTYPE).copy() if n else np.zeros(0, formats.MM_DTYPE)
    if v.a:
        libc.free.argtypes = [C.c_void_p]
        libc.free(C.c_void_p(v.a))
    return out


def test_mm_sketch_symbol_on_reference_vectors(lib):
    """the shimmer4py mm_sketch symbol (ASCII in, kvec out) against the reference's outputs for adversarial strings"""
    n = 0
    for i, s, w, k, want in G.sketch_cases():
        v = KV()
        lib.mm_sketch(None, C.c_char_p(s), C.c_int(len(s)), C.c_int(w), C.c_int(k), C.c_uint32(7 + i), C.c_int(0), C.byref(v))
        got = _take_kv(v)
        assert np.array_equal(got, want), f"case {i} w={w} k={k} len={len(s)}"
        n += 1
    assert n > 200


def test_mm_reduce_on_reference_vectors(lib):
    for i, inp, rs, want in G.reduce_cases():
        assert np.array_equal(mm_reduce(inp, rs), want), f"case {i} rs={rs}"


def test_ovlp_match_symbol_on_reference_vectors(lib):
    lib.ovlp_match.restype = C.POINTER(U.Match)
    for i, q, qs, t, ts, band, want in G.match_cases():
        q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
        p = lib.ovlp_match(q.ctypes.data_as(C.c_void_p), C.c_int32(len(q)), C.c_uint8(qs), t.ctypes.data_as(C.c_void_p),
                           C.c_int32(len(t)), C.c_uint8(ts), C.c_int32(band))
        got = p.contents.astuple()
        lib.free_ovlp_match(p)
        assert got == want, f"case {i}: {got} != {want}"


@pytest.mark.parametrize("T", [1, 2])
def test_index_files_match_reference(tmp_path, T):
    z = G.load("tiny_stage.npz")
    db = G.tiny_db(z)
    pre = str(tmp_path / "sd")
    formats.write_seqdb(pre, db)
    for c in range(1, T + 1):
        for lv in (2, 1):
            o = str(tmp_path / f"o{lv}")
            shmr_index(pre, o, T, c, lv, 6, 1, 80, 16)
            tag = f"{c:02d}-of-{T:02d}"
            for L in (("L0", "L2") if lv == 2 else ("L1",)):
                assert np.array_equal(formats.read_mmlist(f"{o}-{L}-{tag}.dat"), z[f"ix{T}l{lv}_{L}_{c}"]), (T, c, L)
                mc = formats.mc_as_sorted_pairs(formats.read_mm_count(f"{o}-{L}-MC-{tag}.dat"))
                assert np.array_equal(mc, z[f"ix{T}l{lv}_{L}MC_{c}"]), (T, c, L)


@pytest.mark.parametrize("name", sorted(G.OVERLAP_RUNS))
def test_overlap_files_match_reference(tmp_path, name):
    z = G.load("tiny_stage.npz")
    db = G.tiny_db(z)
    IT, lv, OT, kw = G.OVERLAP_RUNS[name]
    kw = dict(kw)
    if "band" in kw:
        kw["align_bandwidth"] = kw.pop("band")
    pre = str(tmp_path / "sd")
    formats.write_seqdb(pre, db)
    for c in range(1, IT + 1):
        shmr_index(pre, str(tmp_path / "ix"), IT, c, lv, 6, 0, 80, 16)
    for c in range(1, OT + 1):
        out = str(tmp_path / f"ov.{c}")
        st = shmr_overlap(pre, str(tmp_path / f"ix-L{lv}"), out, OT, c, **kw)
        got = formats.read_ovlp(out)
        want = z[f"{name}_{c}"]
        assert len(got) == st["n_records"] == len(want), (name, c, len(got), len(want))
        assert formats.ovlp_fields_equal(got, want), (name, c)
        assert st["n_align_gpu"] >= st["n_align_needed"] > 0


@pytest.fixture(scope="module")
def small():
    db = simreads.make_workload("small")  # 1 Mb x 16x, ~1.1k reads, 16.7 Mbases
    rdb = ResidentDB(db, 0)
    yield db, rdb
    rdb.close()


def test_small_dataset_index_vs_oracle(small):
    db, rdb = small
    ix = rdb.index(want_l0=True)
    l0 = np.concatenate([U.orc_sketch_seqdb(db.seqdb[int(o):int(o) + int(n)], 80, 16, int(r))
                         for r, n, o in zip(db.rid, db.rlen, db.roff)])
    assert np.array_equal(ix.l0, l0)
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    assert np.array_equal(ix.top, l2)
    assert np.array_equal(formats.mc_as_sorted_pairs(ix.top_mc), formats.mc_as_sorted_pairs(U.orc_count(l2)))
    assert np.array_equal(formats.mc_as_sorted_pairs(ix.l0_mc), formats.mc_as_sorted_pairs(U.orc_count(l0)))
    assert np.array_equal(rdb.index(levels=1).top, l1)
    # the fused path (no L0 requested: wave sketch + in-LDS reduce) must agree with the general path and the oracle
    fz = rdb.index()
    assert fz.reads_literal == 0 and np.array_equal(fz.top, l2)
    assert np.array_equal(formats.mc_as_sorted_pairs(fz.top_mc), formats.mc_as_sorted_pairs(U.orc_count(l2)))
    for r in (2, 3, 24):
        assert np.array_equal(rdb.index(reduction=r).top, U.orc_reduce(U.orc_reduce(l0, r), r)), r
        assert np.array_equal(rdb.index(reduction=r, levels=1).top, U.orc_reduce(l0, r)), r
    # other parameters
    # (64/96/128, 16) have specialised closed-form kernels; every other (w, k) runs on the general closed-form kernel
    for (w, k, r) in ((24, 12, 3), (40, 15, 6), (100, 16, 4), (64, 16, 6), (96, 16, 5), (128, 16, 3), (255, 28, 6), (25, 13, 2),
                      (81, 17, 6)):
        a = rdb.index(window=w, kmer=k, reduction=r, want_l0=True)
        b0 = np.concatenate([U.orc_sketch_seqdb(db.seqdb[int(o):int(o) + int(n)], w, k, int(rr))
                             for rr, n, o in list(zip(db.rid, db.rlen, db.roff))])
        assert np.array_equal(a.l0, b0), (w, k)
        assert np.array_equal(a.top, U.orc_reduce(U.orc_reduce(b0, r), r)), (w, k, r)
        assert a.reads_literal == 0, (w, k, a.reads_literal)   # (only reads with ambiguous bases need the state machine)


@pytest.mark.parametrize("OT", [1, 3])
def test_small_dataset_overlap_vs_oracle(small, OT):
    db, rdb = small
    parts = [rdb.index(total_chunk=2, mychunk=c) for c in (1, 2)]
    mm = np.concatenate([p.top for p in parts])
    mc = np.concatenate([p.top_mc for p in parts])
    for c in range(1, OT + 1):
        got, st = rdb.overlap(mm, mc, total_chunk=OT, mychunk=c)
        want, ost = U.orc_overlap(db, mm, mc, mychunk=c, total=OT)
        assert len(want) > 1000
        assert formats.ovlp_fields_equal(got, want)
        assert st["n_align_needed"] == ost["n_align"] and st["n_pair_records"] == ost["n_records"]


def test_align_batch_vs_oracle(small):
    db, rdb = small
    rng = np.random.default_rng(9)
    n = 400
    keys = np.zeros(n, _lib.ALIGN_KEY_DTYPE)
    keys["rid0"] = rng.integers(0, db.n_reads, n)
    keys["rid1"] = rng.integers(0, db.n_reads, n)
    keys["q_off"] = rng.integers(0, 3000, n)
    keys["dir0"] = rng.integers(0, 2, n)
    keys["dir1"] = rng.integers(0, 2, n)
    for band in (100, 20, 130):
        got = rdb.align(keys, band)
        for i in range(n):
            a, b = int(keys["rid0"][i]), int(keys["rid1"][i])
            q = db.seqdb[int(db.roff[a]) + int(keys["q_off"][i]):int(db.roff[a]) + int(db.rlen[a])]
            t = db.seqdb[int(db.roff[b]):int(db.roff[b]) + int(db.rlen[b])]
            want = U.orc_ovlp_match(q, int(keys["dir0"][i]), t, int(keys["dir1"][i]), band)
            assert tuple(int(v) for v in got[i].tolist()) == want, (i, band)


def test_edge_cases(lib):
    # empty selections, a chunk that owns no read, reads shorter than a window, an all-N read
    g = simreads.make_genome(20000, 3)
    db = simreads.simulate_reads(g, n_reads=3, seed=1, mean_len=300, sd_len=200, wrap=0, min_len=20)
    sd = db.seqdb.copy()
    sd[int(db.roff[1]):int(db.roff[1]) + int(db.rlen[1])] = 0
    db.seqdb = sd
    rdb = ResidentDB(db, 0)
    ix = rdb.index(total_chunk=5, mychunk=4, want_l0=True)   # rid % 5 == 4: nobody
    assert len(ix.l0) == 0 and len(ix.top) == 0 and len(ix.top_mc) == 0 and ix.reads == 0
    ix = rdb.index(want_l0=True)
    l0 = np.concatenate([U.orc_sketch_seqdb(db.seqdb[int(o):int(o) + int(n)], 80, 16, int(r))
                         for r, n, o in zip(db.rid, db.rlen, db.roff)])
    assert np.array_equal(ix.l0, l0)
    assert np.array_equal(ix.top, U.orc_reduce(U.orc_reduce(l0, 6), 6))
    ov, st = rdb.overlap(ix.top, ix.top_mc)
    assert len(ov) == 0
    assert len(mm_reduce(np.zeros(0, formats.MM_DTYPE), 6)) == 0 and len(mm_count(np.zeros(0, formats.MM_DTYPE))) == 0
    # bad arguments return errors instead of aborting the process
    with pytest.raises(_lib.PgxError):
        rdb.index(window=10)
    with pytest.raises(_lib.PgxError):
        rdb.index(total_chunk=2, mychunk=3)
    rdb.close()


@pytest.mark.parametrize("piece,batch", [(None, None), ("64", "1000"), ("97", "1"), ("4096", "100000")])
def test_mkseqdb_matches_reference(tmp_path, monkeypatch, piece, batch):
    """row f1: FASTA / FASTQ / gz / CRLF / multi-line inputs -> seqdb + idx, byte-identical to the reference binary; the input is
    read piece by piece and encoded batch by batch (bounded host memory): tiny pieces / batches put record boundaries, header
    markers and quality strings on every piece border"""
    from peregrine_amd.shimmer import shmr_mkseqdb
    if piece:
        monkeypatch.setenv("PGX_MKSEQDB_PIECE", piece)
        monkeypatch.setenv("PGX_MKSEQDB_BATCH", batch)
    z = G.load("mkseqdb_cases.npz")
    paths = []
    for k in z["order"]:
        p = tmp_path / str(k)
        p.write_bytes(z["file_" + str(k)].tobytes())
        paths.append(str(p))
    (tmp_path / "seq.lst").write_text("\n".join(paths) + "\n")
    st = shmr_mkseqdb(str(tmp_path / "seq.lst"), str(tmp_path / "out"))
    assert (tmp_path / "out.seqdb").read_bytes() == z["seqdb"].tobytes()
    assert (tmp_path / "out.idx").read_bytes() == z["idx"].tobytes()
    assert st["bases"] == len(z["seqdb"])
    # and a simulated read set round-trips: FASTA -> seqdb equals the simulator's own encoding
    db = simreads.make_workload("tiny")
    simreads.seqdb_to_fasta(db, str(tmp_path / "reads.fa"))
    (tmp_path / "r.lst").write_text(str(tmp_path / "reads.fa") + "\n")
    shmr_mkseqdb(str(tmp_path / "r.lst"), str(tmp_path / "sd"))
    assert np.array_equal(np.fromfile(tmp_path / "sd.seqdb", np.uint8), db.seqdb)
    with pytest.raises(_lib.PgxError):
        shmr_mkseqdb(str(tmp_path / "missing.lst"), str(tmp_path / "x"))


def test_dedup_matches_reference(tmp_path):
    """row f2: cat ovlp*.dat | shmr_dedup -- GPU first-wins + coordinate transform, text byte-identical to the reference"""
    from peregrine_amd.shimmer import shmr_dedup
    z = G.load("tiny_stage.npz")
    d = G.load("dedup_cases.npz")
    for name in ("dd_t1", "dd_t2", "dd_t3", "dd_l1"):
        paths = []
        for k in d[name + "_keys"]:
            p = tmp_path / f"{name}_{k}.dat"
            z[str(k)].tofile(p)
            paths.append(str(p))
        text, nu = shmr_dedup(paths, str(tmp_path / f"{name}.ovl"))
        assert text == d[name].tobytes(), name
        assert (tmp_path / f"{name}.ovl").read_bytes() == text and nu == text.count(b"\n")
    assert shmr_dedup([])[0] == b""


def test_parallel_replay_equals_oracle(small, monkeypatch):
    """the multi-threaded replay (forced on for a small set) must reach the sequential fixed point, run after run"""
    db, rdb = small
    ix = rdb.index()
    want, ost = U.orc_overlap(db, ix.top, ix.top_mc)
    monkeypatch.setenv("PGX_PAR_MIN", "0")
    rng = np.random.default_rng(3)
    for it in range(60):   # (tools/replay_stress.py is the long version of this loop)
        threads, block = int(rng.choice([2, 3, 8, 16, 32, 64])), int(rng.choice([1, 3, 16, 64]))
        monkeypatch.setenv("PGX_THREADS", str(threads))
        monkeypatch.setenv("PGX_BLOCK", str(block))       # work-block size: 1 maximises the cross-thread conflicts
        monkeypatch.setenv("PGX_PIN", str(it & 1))
        got, st = rdb.overlap(ix.top, ix.top_mc)
        assert formats.ovlp_fields_equal(got, want), (threads, block)
        assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"]


def _enc(codes):
    codes = np.asarray(codes, np.uint8)
    return ((np.uint8(1) << codes) | ((np.uint8(8) >> codes[::-1]) << np.uint8(4))).astype(np.uint8)


def test_adversarial_reads_all_paths():
    """long contig-like reads (> 1024 minimizers), strand-ambiguous runs, homopolymers / tandem repeats (tie bursts, slab
    overflow), tiny reads and ambiguous bases: every dispatch path of the index stage against the oracle"""
    rng = np.random.default_rng(123)
    rnd = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    at = lambda n: np.resize(np.array([0, 3], np.uint8), n)
    reads = [
        rnd(300_000),                                                  # contig-like: k_reduce_read cannot hold it
        np.concatenate([rnd(4000), at(400), rnd(3000), at(39), rnd(800)]),   # long strand-ambiguous k-mer runs
        np.concatenate([rnd(2000), np.zeros(600, np.uint8), rnd(2500)]),      # poly-A: every window ties
        np.resize(rnd(3), 5000), np.resize(rnd(7), 6000), np.resize(rnd(16), 4000), np.resize(rnd(40), 9000),
        rnd(1), rnd(15), rnd(16), rnd(17), rnd(94), rnd(95), rnd(96), rnd(110), rnd(1023), rnd(1024), rnd(1025), rnd(2049),
        rnd(15000), rnd(15001),
    ]
    enc = [_enc(r) for r in reads]
    amb = _enc(rnd(7000)).copy()
    amb[[10, 3000, 6990]] &= 0xF0                                        # ambiguous forward bases -> run by run (pgx_sketch_n.hip)
    enc.append(amb)
    enc += [_enc(rnd(int(n))) for n in rng.integers(5000, 20000, 40)]
    rlen = np.array([len(e) for e in enc], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    from peregrine_amd.formats import SeqDB
    db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    l0 = np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc)])
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    a = rdb.index(want_l0=True)
    assert a.reads_literal >= 1
    assert np.array_equal(a.l0, l0) and np.array_equal(a.top, l2)
    assert np.array_equal(rdb.index().top, l2)                          # fused attempt -> general path
    assert np.array_equal(rdb.index(levels=1).top, l1)
    assert np.array_equal(rdb.index(total_chunk=3, mychunk=1).top,
                          U.orc_reduce(U.orc_reduce(np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc) if i % 3 == 1]), 6), 6))
    import os
    os.environ["PGX_FUSE"] = "1"   # read when the fused path is first consulted in this process; harmless if already cached
    try:
        assert np.array_equal(rdb.index().top, l2)
    finally:
        del os.environ["PGX_FUSE"]
    ov, _ = rdb.overlap(a.top, a.top_mc)
    want, _ = U.orc_overlap(db, l2, U.orc_count(l2))
    assert formats.ovlp_fields_equal(ov, want)
    rdb.close()


def _with_ambiguous(codes, positions, both=True):
    e = _enc(codes).copy()
    for p in positions:
        e[p] &= 0xF0 if not both else 0x00       # ambiguous forward base (and, `both`, its mirror image on the reverse strand's nibble)
        if both:
            e[len(e) - 1 - p] &= 0x0F
    return e


@pytest.mark.parametrize("tiny_slabs", [False, True])
def test_reads_with_ambiguous_bases_run_by_run(monkeypatch, tiny_slabs):
    """mm_sketch resets only its run length at an ambiguous base (src/mm_sketch.c:112-113): reads with ambiguous bases are cut into
    runs of unambiguous bases that the closed-form kernels sketch, minus each run's pending minimum, plus the element the end of the
    sequence emits (pgx_sketch_n.hip).  Patterns: single / many / clustered ambiguous bases, at the read ends, after strand-ambiguous
    and low-complexity stretches (stale k-mer state), runs shorter than a window / than a k-mer, a read of ambiguous bases only; every
    list level, both index paths, other (w, k) through the general kernel."""
    if tiny_slabs:
        monkeypatch.setenv("PGX_SLAB_DIV", "1000000")
        monkeypatch.setenv("PGX_SLAB_MIN", "8")
    rng = np.random.default_rng(2024)
    rnd = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    at = lambda n: np.resize(np.array([0, 3], np.uint8), n)
    enc = [
        _with_ambiguous(rnd(7000), [10, 3000, 6990]),
        _with_ambiguous(rnd(15000), [0]), _with_ambiguous(rnd(15000), [14999]), _with_ambiguous(rnd(15000), [14999 - 40]),
        _with_ambiguous(rnd(12000), list(range(5000, 5040))),                            # a run of 40 ambiguous bases
        _with_ambiguous(rnd(9000), sorted(rng.choice(9000, 60, replace=False))),        # every ~150 bases: most runs hold one window or none
        _with_ambiguous(rnd(9000), sorted(rng.choice(9000, 400, replace=False))),       # runs shorter than a window
        _with_ambiguous(np.concatenate([rnd(3000), at(500), rnd(3000)]), [3499, 3500, 3520]),   # stale strand-ambiguous k-mers across the break
        _with_ambiguous(np.concatenate([rnd(2000), np.zeros(700, np.uint8), rnd(2000)]), [2300, 2350, 2699]),   # inside a homopolymer
        _with_ambiguous(np.resize(rnd(5), 8000), [4000]),                                # inside a tandem array: bursts of ties + a break
        _with_ambiguous(rnd(40), [20]), _with_ambiguous(rnd(16), [15]), _with_ambiguous(rnd(17), [0]), _with_ambiguous(rnd(200), [100]),
        _with_ambiguous(rnd(300), list(range(300))),                                     # nothing but ambiguous bases
        _with_ambiguous(rnd(15000), [7000], both=False),                                 # ambiguous on the forward strand only
        _with_ambiguous(rnd(6000), [5999 - 90, 5999 - 30]),                              # the end element reaches back across breaks
        _with_ambiguous(rnd(6000), [5999 - 17]), _with_ambiguous(rnd(6000), [5999 - 16]), _with_ambiguous(rnd(6000), [5999 - 15]),
    ]
    n_amb = len(enc)
    enc += [_enc(rnd(int(n))) for n in rng.integers(3000, 20000, 30)]
    order = rng.permutation(len(enc))
    enc = [enc[i] for i in order]
    rlen = np.array([len(e) for e in enc], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    from peregrine_amd.formats import SeqDB
    db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    l0 = np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc)])
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    a = rdb.index(want_l0=True)                                   # general path (dev_sketch + list reduce)
    assert a.reads_literal >= n_amb - 1                           # (the forward-only one included; a read of 16 bases with base 15 ambiguous too)
    assert np.array_equal(a.l0, l0) and np.array_equal(a.top, l2)
    f2 = rdb.index()                                              # fused path: k_sketch_blk -> flagged -> run by run + per-read reduce
    assert np.array_equal(f2.top, l2) and f2.reads_literal >= n_amb - 1
    assert np.array_equal(rdb.index(levels=1).top, l1)
    sel = [i for i in range(len(enc)) if i % 3 == 2]
    assert np.array_equal(rdb.index(total_chunk=3, mychunk=2).top,
                          U.orc_reduce(U.orc_reduce(np.concatenate([U.orc_sketch_seqdb(enc[i], 80, 16, i) for i in sel]), 6), 6))
    for w, k in ((64, 16), (128, 16), (24, 12), (11, 13), (80, 15), (200, 28), (5, 4)):     # wave kernel at other windows; the general kernel
        if k < 12 or w < 24 or w <= k:
            got = rdb.sketch(np.arange(len(enc), dtype=np.uint32), w, k)                  # (shmr_index asserts w >= 24, k >= 12; mm_sketch itself does not)
        else:
            got = rdb.index(window=w, kmer=k, want_l0=True).l0
        want = np.concatenate([U.orc_sketch_seqdb(e, w, k, i) for i, e in enumerate(enc)])
        assert np.array_equal(got, want), (w, k)
    rdb.close()


def test_low_complexity_read_of_the_c4s_set():
    """read 279,270 of the c4s set (tests/golden/c4s_read_279270.npy: an (AC)n array of ~360 bases, then ~100 bases, then (AC)n again):
    found by tools/l2diff.py -- the one read of 600,080 whose final-level list differed from the reference's file in round 3's tree"""
    import os
    rb = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4s_read_279270.npy"))
    enc = [rb, rb[:9000], rb[9000:], rb[2:], rb]
    rlen = np.array([len(e) for e in enc], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    from peregrine_amd.formats import SeqDB
    db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    l0 = np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc)])
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    a = rdb.index(want_l0=True)
    assert np.array_equal(a.l0, l0) and np.array_equal(a.top, l2)
    assert np.array_equal(rdb.index(levels=1).top, l1)
    assert np.array_equal(rdb.index().top, l2)
    rdb.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sparse_entry_reads_random(seed):
    """Reads whose ENTRY stream is sparse: self-complementary tandem arrays ((GC)n, (AT)n, (ACGT)n, (GAATTC)n ...) yield no entries at all
    (every k-mer is its own reverse complement), so a few point mutations inside one make the first window of 80 entries span a whole
    tile or more, repeat the same mutated k-mers (ties with entry w-1), and put drops everywhere -- the geometry of c4s read 279,270.
    Random mixes of such arrays, plain random sequence and other short-period arrays, at many lengths, against the oracle."""
    rng = np.random.default_rng(9000 + seed)
    rnd = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    units = [np.array(u, np.uint8) for u in ([2, 1], [0, 3], [0, 1, 2, 3], [2, 0, 0, 3, 3, 1], [1, 2], [3, 0], [0, 0, 3, 3], [1, 1, 2, 2])]

    def piece():
        kind = rng.integers(0, 4)
        n = int(rng.integers(20, 2500))
        if kind == 0:
            return rnd(n)
        if kind == 3:
            return np.resize(rnd(int(rng.integers(1, 9))), n)
        a = np.resize(units[int(rng.integers(0, len(units)))], n).copy()
        nm = int(rng.integers(0, 1 + n // 60))
        for p in rng.integers(0, n, nm):
            a[p] = (a[p] + rng.integers(1, 4)) % 4
        if nm and rng.random() < 0.5:      # the same mutation again a period multiple later: equal mutated k-mers (ties)
            p = int(rng.integers(0, n)); d = 2 * len(units[0]) * int(rng.integers(8, 200))
            if p + d < n:
                a[p + d] = a[p] = (a[p] + 1) % 4
        return a

    reads = [np.concatenate([piece() for _ in range(int(rng.integers(1, 7)))]) for _ in range(260)]
    reads += [np.resize(units[i % len(units)], int(n)) for i, n in enumerate(rng.integers(16, 3000, 16))]   # pure arrays: no entry at all
    enc = [_enc(r) for r in reads]
    rlen = np.array([len(e) for e in enc], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    from peregrine_amd.formats import SeqDB
    db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    per = [U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc)]
    l0 = np.concatenate(per)
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    a = rdb.index(want_l0=True)
    if not np.array_equal(a.l0, l0):
        got_rid = (a.l0["y"] >> np.uint64(32)).astype(np.int64)
        for i, w in enumerate(per):
            g = a.l0[got_rid == i]
            assert np.array_equal(g, w), (i, len(enc[i]), len(g), len(w))
    assert np.array_equal(a.top, l2)
    assert np.array_equal(rdb.index().top, l2) and np.array_equal(rdb.index(levels=1).top, l1)
    for w in (64, 96, 128):
        assert np.array_equal(rdb.index(window=w, want_l0=True).l0, np.concatenate([U.orc_sketch_seqdb(e, w, 16, i) for i, e in enumerate(enc)])), w
    rdb.close()


@pytest.mark.parametrize("tiny_slabs", [False, True])
def test_low_complexity_reads_stay_on_the_fused_index_path(monkeypatch, tiny_slabs):
    """homopolymers and short-period tandem arrays make every position a tied minimizer (bursts of up to 1,024 per tile, thousands
    of top-level shimmers per read): the fused index path stages the bursts in pieces and redoes reads that outgrow their slab
    into exact slabs -- no read leaves the fused index path, and the lists equal the oracle's"""
    if tiny_slabs:   # 8 elements per read: nearly every read outgrows its slab and goes through the exact-slab pass
        monkeypatch.setenv("PGX_SLAB_DIV", "1000000")
        monkeypatch.setenv("PGX_SLAB_MIN", "8")
    rng = np.random.default_rng(77)
    rnd = lambda n: rng.integers(0, 4, n).astype(np.uint8)
    reads = [
        np.concatenate([rnd(3000), np.zeros(400, np.uint8), rnd(5000)]),            # poly-A run
        np.concatenate([rnd(2000), np.resize(rnd(2), 3000), rnd(4000)]),            # period 2
        np.concatenate([rnd(1000), np.resize(rnd(7), 2500), rnd(6000), np.resize(rnd(39), 3000), rnd(500)]),
        np.resize(rnd(3), 15000),                                                   # all tandem: ~5,000 shimmers on every level
        np.resize(rnd(5), 20000), np.full(9000, 2, np.uint8),                       # all tandem / one long homopolymer
        np.concatenate([np.full(1500, 1, np.uint8), rnd(9000), np.full(1500, 3, np.uint8)]),   # runs at both read ends
    ]
    reads += [rnd(int(n)) for n in rng.integers(5000, 20000, 60)]
    for i in range(0, 40, 4):                                                       # errors inside the arrays, like real reads
        r = reads[7 + i].copy()
        r[2000:2600] = np.resize(rnd(4), 600)
        r[2100] ^= 1
        r[2300] ^= 2
        reads[7 + i] = r
    enc = [_enc(r) for r in reads]
    rlen = np.array([len(e) for e in enc], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    from peregrine_amd.formats import SeqDB
    db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    l0 = np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc)])
    l1 = U.orc_reduce(l0, 6)
    l2 = U.orc_reduce(l1, 6)
    a = rdb.index()
    assert np.array_equal(a.top, l2)
    import os
    default_path = not (os.environ.get("PGX_SKETCH") or os.environ.get("PGX_FUSE"))   # (the round-1 kernels do fall back)
    assert a.reads_literal == 0 or not default_path                                 # nothing fell back to the general path
    b = rdb.index(levels=1)
    assert np.array_equal(b.top, l1) and (b.reads_literal == 0 or not default_path)
    assert np.array_equal(rdb.index(total_chunk=2, mychunk=2).top,
                          U.orc_reduce(U.orc_reduce(np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(enc) if i % 2 == 0]), 6), 6))
    rdb.close()


def _write_query_files(tmp_path):
    """the tiny set's seqdb / idx / two-chunk level-2 index as files (inputs of query_cases.npz)"""
    q, mmers, mc, rlen = G.query_fixture()
    t = G.load("tiny_stage.npz")
    pre, sp = str(tmp_path / "sd"), str(tmp_path / "ix-L2")
    (tmp_path / "sd.seqdb").write_bytes(t["seqdb"].tobytes())
    (tmp_path / "sd.idx").write_bytes(t["idx_text"].tobytes())
    for c in (1, 2):
        formats.write_mmlist(f"{sp}-{c:02d}-of-02.dat", t[f"ix2l2_L2_{c}"])
        m = np.zeros(len(t[f"ix2l2_L2MC_{c}"]), formats.MC_DTYPE)
        m["mer"], m["count"] = t[f"ix2l2_L2MC_{c}"][:, 0], t[f"ix2l2_L2MC_{c}"][:, 1]
        formats.write_mm_count(f"{sp}-MC-{c:02d}-of-02.dat", m)
    return q, mmers, mc, rlen, pre, sp


def test_query_helpers_match_reference(tmp_path):
    """row f4: build_shimmer_map4py + get_shimmer_hits / get_mmer_count / get_shimmers_for_read vs the compiled reference's
    answers (tests/golden/query_cases.npz) on every key of the tiny set, 5 chunk / bound settings"""
    from peregrine_amd.shimmer import ShimmerMap
    q, mmers, mc, rlen, pre, sp = _write_query_files(tmp_path)
    for (c, T, lo, hi) in ((1, 1, 2, 240), (1, 2, 2, 240), (2, 2, 2, 240), (1, 1, 1, 3), (2, 3, 2, 30)):
        tag = f"c{c}t{T}lo{lo}hi{hi}"
        m = ShimmerMap(pre, sp, c, T, lo, hi)
        assert np.array_equal(m.mmers, mmers)
        want, off = q[f"hits_{tag}"], q[f"hoff_{tag}"]
        for i, k in enumerate(q["qkeys"]):
            got = m.hits(int(k) >> 8, int(k) & 0xFF)
            assert got.tobytes() == want[off[i]:off[i + 1]].tobytes(), (tag, i)
        if tag == "c1t1lo2hi240":
            assert [m.mmer_count(int(k) >> 8) for k in q["qkeys"]] == q["counts"].tolist()
            for r, f, n in zip(q["qrids"], q["read_first"], q["read_count"]):
                gf, gn = m.read_range(int(r))
                assert gn == n and (n == 0 or gf == f)
                assert np.array_equal(m.shimmers_for_read(int(r)), mmers[f:f + n] if n else mmers[:0])
        m.close()
    with pytest.raises(_lib.PgxError):
        ShimmerMap(str(tmp_path / "missing"), sp)


def test_query_helpers_match_oracle_on_small_set(small, tmp_path):
    """row f4 on a larger set against the oracle: every hit list of 400 sampled keys (+ absent ones), chunked ownership"""
    from peregrine_amd.shimmer import ShimmerMap
    db, rdb = small
    formats.write_seqdb(str(tmp_path / "sd"), db)
    for c in (1, 2, 3):
        ix = rdb.index(total_chunk=3, mychunk=c)
        formats.write_mmlist(str(tmp_path / f"ix-L2-{c:02d}-of-03.dat"), ix.top)
        formats.write_mm_count(str(tmp_path / f"ix-L2-MC-{c:02d}-of-03.dat"), ix.top_mc)
    mm = np.concatenate([formats.read_mmlist(str(tmp_path / f"ix-L2-{c:02d}-of-03.dat")) for c in (1, 2, 3)])
    mcs = np.concatenate([formats.read_mm_count(str(tmp_path / f"ix-L2-MC-{c:02d}-of-03.dat")) for c in (1, 2, 3)])
    rl, _ = db.by_rid()
    rng = np.random.default_rng(8)
    keys = np.concatenate([rng.choice(np.unique(mm["x"]), 400, replace=False), np.array([12345 << 8 | 16], np.uint64)])
    for (c, T) in ((1, 1), (2, 2)):
        m, o = ShimmerMap(str(tmp_path / "sd"), str(tmp_path / "ix-L2"), c, T), U.OrcMap(mm, mcs, rl, c, T)
        nhit = 0
        for k in keys:
            got, want = m.hits(int(k) >> 8, int(k) & 0xFF), o.hits(int(k) >> 8, int(k) & 0xFF)
            assert got.tobytes() == want.tobytes(), (c, T, int(k))
            assert m.mmer_count(int(k) >> 8) == o.count(int(k) >> 8)
            nhit += len(got)
        assert nhit > 1000
        for r in rng.integers(0, db.n_reads, 50):
            assert m.read_range(int(r)) == o.read_shimmers(int(r))
        m.close(), o.close()


def test_map_matches_reference(tmp_path):
    """row f3: pgx_map_chunk / bin/shmr_map vs the compiled reference's shmr_map stdout (contigs cut from the reads' genome,
    one of them reverse-complemented), 3 chunk / bound settings"""
    import subprocess
    import sys
    from peregrine_amd.shimmer import map_reads_to_ref, shmr_map
    q, mmers, mc, rlen, pre, sp = _write_query_files(tmp_path)
    formats.write_mmlist(str(tmp_path / "ref-L2-01-of-01.dat"), q["ref_l2"])
    for (c, T, lo, hi) in ((1, 1, 1, 240), (2, 2, 1, 240), (1, 1, 2, 4)):
        want = q[f"map_c{c}t{T}lo{lo}hi{hi}"].tobytes()
        text, n = shmr_map(str(tmp_path / "ref-L2"), pre, sp, str(tmp_path / "ref"), T, c, lo, hi)
        assert text == want and n == want.count(b"\n"), (c, T, lo, hi)
        text2, _ = map_reads_to_ref(q["ref_l2"], mmers, mc, rlen, T, c, lo, hi)
        assert text2 == want
    cli = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "shmr_map"), "-r", str(tmp_path / "ref"), "-m", str(tmp_path / "ref-L2"),
                          "-p", pre, "-l", sp, "-t", "2", "-c", "2"], check=True, stdout=subprocess.PIPE).stdout
    assert cli == q["map_c2t2lo1hi240"].tobytes()


def test_map_matches_oracle_on_small_set(small):
    """row f3 at a larger size against the oracle: the reads' own genome as contigs is not available here, so the reads
    themselves (every 7th, as 'contigs') are mapped against all reads"""
    from peregrine_amd.shimmer import map_reads_to_ref
    db, rdb = small
    ix = rdb.index()
    rl, _ = db.by_rid()
    rid = (ix.top["y"] >> np.uint64(32)).astype(np.int64)
    ref = ix.top[rid % 7 == 0]
    for (c, T, lo, hi) in ((1, 1, 1, 240), (2, 3, 2, 30)):
        text, n = map_reads_to_ref(ref, ix.top, ix.top_mc, rl, T, c, lo, hi)
        want, wn = U.orc_map_reads_to_ref(ref, ix.top, ix.top_mc, rl, c, T, lo, hi)
        assert n == wn and text == want and n > 1000, (c, T, n, wn)


def test_repeat_rich_set_parallel_replay_vs_oracle():
    """a genome with planted repeat families and tandem arrays (deep buckets, multiplicity cut-offs, best-n saturation,
    many rejected candidates => long correction cascades), large enough for the multi-threaded replay by default; several
    parameter sets, every record compared with the oracle"""
    g = simreads.make_genome(2_000_000, 31, repeat_families=6, repeat_len=5000, repeat_copies=12, divergence=0.02, tandem=8)
    db = simreads.simulate_reads(g, coverage=24.0, seed=5, mean_len=9000, sd_len=2500, err=0.012)
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    for kw in (dict(), dict(bestn=2, mc_upper=40, ovlp_upper=60), dict(total_chunk=2, mychunk=2, bestn=8, align_bandwidth=60),
               dict(mc_lower=1, mc_upper=1000, ovlp_upper=400)):
        got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
        okw = dict(mychunk=kw.get("mychunk", 1), total=kw.get("total_chunk", 1), mc_lower=kw.get("mc_lower", 2),
                   mc_upper=kw.get("mc_upper", 240), bestn=kw.get("bestn", 4), ovlp_upper=kw.get("ovlp_upper", 120),
                   band=kw.get("align_bandwidth", 100))
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, **okw)
        assert len(want) > 5000 and formats.ovlp_fields_equal(got, want), kw
        assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"], kw
        assert st["n_align_needed"] > st["n_records"]          # the set does produce rejected candidates
    rdb.close()


def test_device_replay_equals_oracle(small, monkeypatch):
    """the greedy walk on the GPU (pgx_replay.hip; chosen by size in production, forced on here) reaches the sequential fixed
    point under every schedule: window sizes from one wavefront's worth of buckets to everything at once, 1-5 inner
    iterations; parameter corners; two overlap chunks; and the repeat-rich set (buckets holding a read twice, best-n
    saturation, ovlp_upper beyond the device encodings -> the host replay must take over silently)"""
    db, rdb = small
    ix = rdb.index()
    monkeypatch.setenv("PGX_GPU_REPLAY", "1")
    rng = np.random.default_rng(11)
    for kw in (dict(), dict(bestn=2, mc_upper=60), dict(bestn=1, ovlp_upper=40), dict(total_chunk=2, mychunk=2), dict(align_bandwidth=30), dict(bestn=0)):
        okw = dict(mychunk=kw.get("mychunk", 1), total=kw.get("total_chunk", 1), mc_upper=kw.get("mc_upper", 240), bestn=kw.get("bestn", 4),
                   ovlp_upper=kw.get("ovlp_upper", 120), band=kw.get("align_bandwidth", 100))
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, **okw)
        for it in range(5):
            monkeypatch.setenv("PGX_REPLAY_WIN", str(int(rng.choice([64, 1024, 16384, 1 << 22]))))
            monkeypatch.setenv("PGX_REPLAY_K", str(int(rng.choice([1, 2, 3, 5]))))
            got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
            assert formats.ovlp_fields_equal(got, want), (kw, it)
            assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"], (kw, it)
            # the checksum k_emit adds up while it writes the records = the numpy statement over the stream the caller got (and the reference's)
            assert st["device_replay"] == 1 and st["replay_attempts"] == 1 and st["stream_checksum"] == formats.stream_checksum(want), (kw, it)
    monkeypatch.delenv("PGX_REPLAY_WIN"), monkeypatch.delenv("PGX_REPLAY_K")
    # reader lists shared by 2 / 4 / 8 neighbouring hot slots (what the library does by itself when HBM is short): spurious re-evaluations, same walk
    want, ost = U.orc_overlap(db, ix.top, ix.top_mc)
    for shift in ("1", "2", "3"):
        monkeypatch.setenv("PGX_REPLAY_COLD_SHIFT", shift)
        got, st = rdb.overlap(ix.top, ix.top_mc)
        assert formats.ovlp_fields_equal(got, want) and st["device_replay"] == 1 and st["n_align_needed"] == ost["n_align"], shift
    monkeypatch.delenv("PGX_REPLAY_COLD_SHIFT")
    # undersized device tables: the walk is repeated with larger ones (hash tables x 4, arenas x 2 per attempt; four attempts), then handed to the host replay
    want, ost = U.orc_overlap(db, ix.top, ix.top_mc)
    seen = []
    for x in ("0.3", "0.02", "0.0005"):
        monkeypatch.setenv("PGX_REPLAY_PAIRS_X", x), monkeypatch.setenv("PGX_REPLAY_MEMO_X", x)
        got, st = rdb.overlap(ix.top, ix.top_mc)
        assert formats.ovlp_fields_equal(got, want) and st["n_align_needed"] == ost["n_align"], x
        assert st["stream_checksum"] == formats.stream_checksum(want), (x, st)
        assert (1 <= st["replay_attempts"] <= 4) if st["device_replay"] else st["replay_attempts"] == 0, (x, st)
        seen.append((st["device_replay"], st["replay_attempts"]))
    assert any(a > 1 for d, a in seen if d) or any(d == 0 for d, a in seen), seen   # some attempt overflowed: repeated, or handed to the host replay
    monkeypatch.delenv("PGX_REPLAY_PAIRS_X"), monkeypatch.delenv("PGX_REPLAY_MEMO_X")
    g = simreads.make_genome(2_000_000, 31, repeat_families=6, repeat_len=5000, repeat_copies=12, divergence=0.02, tandem=8)
    db2 = simreads.simulate_reads(g, coverage=24.0, seed=5, mean_len=9000, sd_len=2500, err=0.012)
    rdb2 = ResidentDB(db2, 0)
    ix2 = rdb2.index()
    for kw in (dict(), dict(bestn=2, mc_upper=40, ovlp_upper=60), dict(total_chunk=2, mychunk=2, bestn=8, align_bandwidth=60),
               dict(mc_lower=1, mc_upper=1000, ovlp_upper=400)):
        got, st = rdb2.overlap(ix2.top, ix2.top_mc, **kw)
        okw = dict(mychunk=kw.get("mychunk", 1), total=kw.get("total_chunk", 1), mc_lower=kw.get("mc_lower", 2),
                   mc_upper=kw.get("mc_upper", 240), bestn=kw.get("bestn", 4), ovlp_upper=kw.get("ovlp_upper", 120),
                   band=kw.get("align_bandwidth", 100))
        want, ost = U.orc_overlap(db2, ix2.top, ix2.top_mc, **okw)
        assert len(want) > 5000 and formats.ovlp_fields_equal(got, want), kw
        assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"], kw
    rdb2.close()


def test_asynchronous_record_delivery(small, monkeypatch):
    """pgx_results_async (round 4): the overlap stage returns once the copy of its records is enqueued on a stream of its own; the NEXT
    stage, pgx_free of the array and pgx_results_wait wait for it.  Two chunks back to back, arrays read after the wait, equal to the
    oracle's; an array dropped while its copy may be in flight; the setting restored."""
    db, rdb = small
    ix = rdb.index()
    l2, mc = ix.top, ix.top_mc
    monkeypatch.setenv("PGX_GPU_REPLAY", "1")
    want = [U.orc_overlap(db, l2, mc, mychunk=c, total=2)[0] for c in (1, 2)]
    assert _lib.results_async(True) is False
    try:
        for rep in range(3):                                   # (from the second use of a size class on the result arrays are pinned)
            a, _ = rdb.overlap(l2, mc, total_chunk=2, mychunk=1)
            b, _ = rdb.overlap(l2, mc, total_chunk=2, mychunk=2)   # (its emit waited for a's copy)
            _lib.results_wait()
            assert formats.ovlp_fields_equal(a, want[0]) and formats.ovlp_fields_equal(b, want[1]), rep
            c, _ = rdb.overlap(l2, mc, total_chunk=2, mychunk=1)
            del c                                              # freed right away: pgx_free waits for the copy before the block is reused
            d, _ = rdb.overlap(l2, mc, total_chunk=2, mychunk=2)
            _lib.results_wait()
            assert formats.ovlp_fields_equal(d, want[1]), rep
    finally:
        assert _lib.results_async(False) is True
    e, _ = rdb.overlap(l2, mc, total_chunk=2, mychunk=1)       # synchronous again
    assert formats.ovlp_fields_equal(e, want[0])


def test_query_and_map_edge_cases(tmp_path):
    """rows f3/f4 at the edges: unrelated contigs (no line), a chunk that owns nothing, reads unknown to the index (error, not a
    crash), an empty shimmer list, repeated / interleaved queries on one map"""
    from peregrine_amd.shimmer import ShimmerMap, map_reads_to_ref
    q, mmers, mc, rlen, pre, sp = _write_query_files(tmp_path)
    rng = np.random.default_rng(3)
    alien = np.zeros(500, formats.MM_DTYPE)                     # shimmers whose hashes the reads never produced
    alien["x"] = (rng.integers(1 << 40, 1 << 41, 500).astype(np.uint64) << np.uint64(8)) | np.uint64(16)
    alien["y"] = (np.uint64(7) << np.uint64(32)) | (np.arange(500, dtype=np.uint64) * np.uint64(400) << np.uint64(1))
    assert map_reads_to_ref(alien, mmers, mc, rlen) == (b"", 0)
    text, n = map_reads_to_ref(q["ref_l2"], mmers[:0], mc, rlen)  # no read shimmers at all
    assert (text, n) == (b"", 0)
    with pytest.raises(_lib.PgxError):                            # a shimmer of read 10^6 with 160 known reads
        bad = mmers.copy()
        bad["y"][5] = (np.uint64(1_000_000) << np.uint64(32)) | np.uint64(10)
        map_reads_to_ref(q["ref_l2"], bad, mc, rlen)
    with pytest.raises(_lib.PgxError):                            # a hash that is missing from the MC files
        map_reads_to_ref(q["ref_l2"], mmers, mc[: len(mc) // 2], rlen)
    m = ShimmerMap(pre, sp, 1, 1, 2, 240)
    keys = q["qkeys"][:40]
    first = [m.hits(int(k) >> 8, int(k) & 0xFF).tobytes() for k in keys]
    for _ in range(3):                                            # the per-map scratch table is reused across queries
        for i in rng.permutation(len(keys)):
            assert m.hits(int(keys[i]) >> 8, int(keys[i]) & 0xFF).tobytes() == first[i]
    assert m.mmer_count(0) == 0 and m.read_range(4_000_000_000) == (0, 0)
    m.close()
    m.close()                                                     # idempotent
    (tmp_path / "e-L2-01-of-01.dat").write_bytes(np.uint64(0).tobytes())
    (tmp_path / "e-L2-MC-01-of-01.dat").write_bytes(np.uint64(0).tobytes())
    e = ShimmerMap(pre, str(tmp_path / "e-L2"))
    assert len(e.mmers) == 0 and len(e.hits(123, 16)) == 0 and e.mmer_count(5) == 0 and e.read_range(1) == (0, 0)
    e.close()


def test_fused_index_overlap_equals_the_two_stages(small):
    """pgx_index_overlap_resident (list and counts handed over in HBM) == pgx_index_resident + pgx_overlap_resident"""
    db, rdb = small
    ix = rdb.index()
    ov, st = rdb.overlap(ix.top, ix.top_mc)
    for want in (False, True):
        ix2, ov2, st2 = rdb.index_overlap(want_index_arrays=want)
        assert formats.ovlp_fields_equal(ov, ov2) and st2["n_align_needed"] == st["n_align_needed"]
        assert ix2.bases == ix.bases and ix2.reads == ix.reads
        if want:
            assert np.array_equal(ix2.top, ix.top) and np.array_equal(ix2.top_mc, ix.top_mc)
        else:
            assert ix2.top is None and ix2.top_mc is None
    ov3, _ = rdb.overlap(ix.top, ix.top_mc, bestn=2, mc_upper=60)
    _, ov4, _ = rdb.index_overlap(bestn=2, mc_upper=60)
    assert formats.ovlp_fields_equal(ov3, ov4)


def test_one_level_index_through_both_stages(small):
    """-l 1 (run_test_one_level.sh): the three-times denser L1 list through the join and the multi-threaded replay, two overlap
    chunks, against the oracle"""
    db, rdb = small
    ix = rdb.index(levels=1)
    l1 = np.concatenate([U.orc_reduce(U.orc_sketch_seqdb(db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])], 80, 16, r), 6)
                         for r in range(db.n_reads)])
    assert np.array_equal(ix.top, l1) and len(l1) > 3 * len(rdb.index().top)
    for c in (1, 2):
        got, st = rdb.overlap(ix.top, ix.top_mc, total_chunk=2, mychunk=c)
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, mychunk=c, total=2)
        assert len(want) > 5000 and formats.ovlp_fields_equal(got, want), c
        assert st["n_align_needed"] == ost["n_align"]


def test_overlap_argument_edges(small):
    """degenerate overlap-stage arguments through the C-ABI: nothing to do, hostile lists, parameter corners (the reference
    asserts / exits on most of these; here they are errors or empty results, never a crash), each compared with the oracle where
    the oracle defines a result"""
    db, rdb = small
    ix = rdb.index()
    empty_mm, empty_mc = np.zeros(0, formats.MM_DTYPE), np.zeros(0, formats.MC_DTYPE)
    ov, st = rdb.overlap(empty_mm, empty_mc)
    assert len(ov) == 0 and st["n_records"] == 0
    ov, st = rdb.overlap(ix.top[:1], ix.top_mc)                       # a single shimmer: no pair
    assert len(ov) == 0
    with pytest.raises(_lib.PgxError):                                # a hash without a count (the reference asserts)
        rdb.overlap(ix.top, ix.top_mc[:10])
    with pytest.raises(_lib.PgxError):
        rdb.overlap(ix.top, ix.top_mc, total_chunk=2, mychunk=3)
    for kw in (dict(bestn=0), dict(ovlp_upper=2), dict(mc_lower=5, mc_upper=4), dict(mc_lower=0, mc_upper=1), dict(bestn=255, ovlp_upper=30),
               dict(align_bandwidth=1), dict(align_bandwidth=400)):
        got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, mc_lower=kw.get("mc_lower", 2), mc_upper=kw.get("mc_upper", 240),
                                  bestn=kw.get("bestn", 4), ovlp_upper=kw.get("ovlp_upper", 120), band=kw.get("align_bandwidth", 100))
        assert formats.ovlp_fields_equal(got, want), (kw, len(got), len(want))
        assert st["n_align_needed"] == ost["n_align"], kw
    dup = np.concatenate([ix.top, ix.top])                            # the same list twice (two copies of every read's run)
    dmc = np.concatenate([ix.top_mc, ix.top_mc])
    got, st = rdb.overlap(dup, dmc)
    want, ost = U.orc_overlap(db, dup, dmc)
    assert formats.ovlp_fields_equal(got, want) and st["n_align_needed"] == ost["n_align"]


# ---- every alignment kernel variant against the oracle ------------------------------------------------------------
def _with_long_reads():
    """the `small` read set + two 100 kb reads of the same genome region: a read beyond 65,535 bases switches the whole stage to the
    32-bit V rings (k_align4<8, int32_t>, pgx_align.hip)"""
    g = simreads.make_genome(400_000, 11)
    a = simreads.simulate_reads(g, coverage=12.0, seed=5, wrap=0)
    b = simreads.simulate_reads(g[100_000:260_000], n_reads=2, seed=6, mean_len=100_000, sd_len=0, wrap=0)
    seq = np.concatenate([a.seqdb, b.seqdb])
    rlen = np.concatenate([a.rlen, b.rlen])
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    return formats.SeqDB(seq, np.arange(len(rlen), dtype=np.uint32), rlen, roff, None)


def _keys_of(db, rdb, n_random, seed):
    ix = rdb.index()
    ov, _ = rdb.overlap(ix.top, ix.top_mc)
    keys = np.zeros(len(ov) + n_random, _lib.ALIGN_KEY_DTYPE)
    k = keys[:len(ov)]
    k["rid0"] = ov["y0"] >> np.uint64(32); k["rid1"] = ov["y1"] >> np.uint64(32)
    k["q_off"] = (((ov["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)) - ((ov["y1"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1))).astype(np.uint32)
    k["dir0"] = ov["strand0"]; k["dir1"] = ov["strand1"]
    rng = np.random.default_rng(seed)
    r = keys[len(ov):]
    r["rid0"] = rng.integers(0, db.n_reads, n_random); r["rid1"] = rng.integers(0, db.n_reads, n_random)
    rl = db.rlen[r["rid0"]].astype(np.int64)
    r["q_off"] = np.where(rng.random(n_random) < 0.2, np.maximum(rl - rng.integers(0, 40, n_random), 0), rng.integers(0, rl))   # incl. queries of < 40 bases
    r["dir0"] = rng.integers(0, 2, n_random); r["dir1"] = rng.integers(0, 2, n_random)
    return keys, ov


def _oracle_matches(db, keys, band):
    out = np.zeros(len(keys), _lib.MATCH_DTYPE)
    for i in range(len(keys)):
        a, b = int(keys["rid0"][i]), int(keys["rid1"][i])
        q = db.seqdb[int(db.roff[a]) + int(keys["q_off"][i]):int(db.roff[a]) + int(db.rlen[a])]
        t = db.seqdb[int(db.roff[b]):int(db.roff[b]) + int(db.rlen[b])]
        out[i] = U.orc_ovlp_match(q, int(keys["dir0"][i]), t, int(keys["dir1"][i]), band)
    return out


ALIGN_VARIANTS = [   # (id, environment, read set): every form dev_align dispatches to (pgx_align.hip)
    ("ph8-packed", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0"), "small"),   # k_align_ph<8, u16, packed>: the default of large launches
    ("ph8-packed-ambiguous", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0"), "withN"),   # ... reads with N: handed on to the byte-wise launch
    ("ph8-packed-stragglers", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ITER_LIMIT="150"), "small"),   # most candidates outlast
                                                                                    # 150 iterations: handed on to k_align1_list, a wavefront each
    ("ph8-packed-ordered", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ORDER_MIN="0"), "small"),   # round 6: the packs of this database are laid
                                                                                    # out by locus key (its overlap stage ran first) and the requests are taken in layout order
    ("ph8-packed-ordered-nw4", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ORDER_MIN="0", PGX_ALIGN_NW="4", PGX_ALIGN_SEG="24"), "small"),   # narrower
                                                                                    # workgroups, a segment of three chunks: every segment boundary crossed many times
    ("ph8-packed-nw1", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_NW="1", PGX_ALIGN_SEG="8"), "small"),   # rounds 2-5's form: a wavefront per workgroup
    ("ph8-packed-file-order", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ORDER_MIN="0"), "small3"),   # a database that never saw an overlap stage:
                                                                                    # packs in the order of the seqdb file, no order list
    ("ph8-bytes", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="-1"), "small2"),  # k_align_ph<8, u16> on the seqdb bytes (a database without packs)
    ("one-per-wave", dict(PGX_ALIGN_SMALL="1000000000"), "small"),                  # k_align1 on a LARGE launch (round 6: from the 2-bit packs -- this database has them by now)
    ("one-per-wave-bytes", dict(PGX_ALIGN_SMALL="1000000000", PGX_ALIGN1_PACKED="0"), "small"),   # ... on the seqdb bytes
    ("one-per-wave-ambiguous", dict(PGX_ALIGN_SMALL="1000000000"), "withN"),         # a database with reads that have no 2-bit codes: bytes
    ("ph8-packed-stragglers-bytes", dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ITER_LIMIT="150", PGX_ALIGN1_PACKED="0"), "small"),   # k_align1_list on the bytes
    ("long-reads-int32", dict(PGX_ALIGN_SMALL="0"), "long"),                        # a 100 kb read in the set: k_align4<8, int32>
]


@pytest.fixture(scope="module")
def variant_sets():
    sets = {}
    withn = simreads.make_workload("small")
    sd = withn.seqdb.copy()
    rng = np.random.default_rng(23)
    for r in rng.choice(withn.n_reads, 40, replace=False):        # 40 reads with a few ambiguous bases (nibble 0 on both strands)
        o, n = int(withn.roff[r]), int(withn.rlen[r])
        sd[o + rng.integers(0, n, 3)] = 0
    withn.seqdb = sd
    # ("small2": the same reads as a second database -- the packs stay with a database once built, so the byte-wise form needs its own)
    for name, db in (("small", simreads.make_workload("small")), ("small2", simreads.make_workload("small")), ("long", _with_long_reads()), ("withN", withn)):
        rdb = ResidentDB(db, 0)
        keys, ov = _keys_of(db, rdb, 3000, 17)
        want = {band: _oracle_matches(db, keys, band) for band in (100, 20)}
        rep = -(-20000 // len(keys))               # a launch of >= 20 k keys: beyond every small-launch threshold
        sets[name] = (db, rdb, np.tile(keys, rep), {band: np.tile(w, rep) for band, w in want.items()})
    db3 = simreads.make_workload("small")          # the same reads once more, aligned without an overlap stage before (no locus keys)
    sets["small3"] = (db3, ResidentDB(db3, 0), sets["small"][2], sets["small"][3])
    yield sets
    for _, rdb, _, _ in sets.values():
        rdb.close()


@pytest.mark.parametrize("vid,env,which", ALIGN_VARIANTS, ids=[v[0] for v in ALIGN_VARIANTS])
def test_align_variants_vs_oracle(variant_sets, vid, env, which):
    db, rdb, keys, want = variant_sets[which]
    if which == "long":
        assert int(db.rlen.max()) > 65535
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for band in (100, 20):
            got = rdb.align(keys, band)
            bad = np.flatnonzero(got != want[band])
            assert len(bad) == 0, (vid, band, len(bad), keys[bad[:3]], got[bad[:3]], want[band][bad[:3]])
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("env,which", [(dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0"), "small"), (dict(PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0"), "withN"),
                                       (dict(PGX_ALIGN_SMALL="0"), "long")], ids=["ph8", "ph8-ambiguous", "align4"])
def test_grouped_alignment_launches_of_every_size(variant_sets, env, which):
    """The grouped kernels take the work counter in chunks of 8 per wavefront (round 4: the per-candidate add on one address was their
    floor): launches smaller than a chunk, one past a chunk, around a wavefront's 8 groups and around 64, each candidate exactly once."""
    db, rdb, keys, want = variant_sets[which]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        for n in (1, 3, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 513, 4099):
            got = rdb.align(keys[:n], 100)
            bad = np.flatnonzero(got != want[100][:n])
            assert len(bad) == 0, (which, n, len(bad), keys[bad[:3]], got[bad[:3]], want[100][bad[:3]])
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def test_overlap_stage_with_long_reads_equals_oracle(variant_sets):
    """the whole overlap stage on the set with 100 kb reads (32-bit V rings in every launch), record for record"""
    db, rdb, _, _ = variant_sets["long"]
    ix = rdb.index()
    want, _ = U.orc_overlap(db, ix.top, ix.top_mc)
    for env in (dict(), dict(PGX_ALIGN_SMALL="0"), dict(PGX_GPU_REPLAY="1", PGX_ALIGN_SMALL="0")):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ov, st = rdb.overlap(ix.top, ix.top_mc)
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert len(want) > 1000 and formats.ovlp_fields_equal(ov, want), env


def test_workgroup_evaluation_of_repeat_buckets_equals_oracle(monkeypatch):
    """k_eval_big (round 3): a workgroup per bucket -- four rows x 128 partners per step, the rows committed in order through LDS
    masks -- also for the buckets that hold a read TWICE (tandem arrays: a read pair met several times within one evaluation; the
    pairs the evaluation has inserted sit in an LDS set, duplicates inside a step cut the step).  Record sequence and counters
    against the oracle with the thresholds from "nearly every bucket" (2) to the defaults, under several schedules."""
    g = simreads.make_genome(600_000, 9, repeat_families=2, repeat_len=4000, repeat_copies=6, tandem=60)
    db = simreads.simulate_reads(g, seed=43, coverage=30, mean_len=9000, sd_len=1500)
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    monkeypatch.setenv("PGX_GPU_REPLAY", "1")
    rng = np.random.default_rng(3)
    for kw in (dict(), dict(bestn=2), dict(align_bandwidth=30), dict(total_chunk=2, mychunk=2, bestn=8)):
        okw = dict(mychunk=kw.get("mychunk", 1), total=kw.get("total_chunk", 1), bestn=kw.get("bestn", 4), band=kw.get("align_bandwidth", 100))
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, **okw)
        assert len(want) > 10000
        for big, dup in ((2, 2), (5, 2), (0, 2), (2, 0), (48, 12), (0, 0)):
            monkeypatch.setenv("PGX_REPLAY_BIG", str(big)), monkeypatch.setenv("PGX_REPLAY_DUP", str(dup))
            monkeypatch.setenv("PGX_REPLAY_WIN", str(int(rng.choice([64, 16384, 1 << 22]))))
            monkeypatch.setenv("PGX_REPLAY_K", str(int(rng.choice([1, 3]))))
            got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
            assert st["device_replay"] == 1 and formats.ovlp_fields_equal(got, want), (kw, big, dup)
            assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"], (kw, big, dup)
    rdb.close()


def test_device_visit_order_equals_oracle_and_host_visit(monkeypatch):
    """pgx_visit.hip (round 3): the inner khash tables of the first-key groups replayed on the device -- a lane per group of up
    to 48 buckets, a wavefront (64 probe positions per step) per larger one -- and the groups' visited buckets placed in the
    host's outer-table slot order.  A genome with one repeat family of 90 exact copies and the multiplicity cut-off lifted makes
    first shimmers with ~100 different successors (groups well beyond 48 buckets, through several table resizes with kick-outs);
    record SEQUENCE against the oracle, and against the host-thread form of the visit order (PGX_DEV_VISIT=0).
    Reference: shmr_overlap.c:206-216 over khash.h:232-336."""
    g = simreads.make_genome(1_200_000, 21, repeat_families=1, repeat_len=3000, repeat_copies=90, divergence=0.0)
    db = simreads.simulate_reads(g, seed=5, coverage=24, mean_len=9000, sd_len=1500)
    monkeypatch.setenv("PGX_EARLY_OUTER_MIN", "1")   # (small sets replay the outer table after the join: not here)
    monkeypatch.setenv("PGX_GPU_REPLAY", "1")
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    for kw in (dict(mc_upper=100000), dict(mc_upper=100000, ovlp_upper=60, bestn=2), dict()):
        want, ost = U.orc_overlap(db, ix.top, ix.top_mc, mc_upper=kw.get("mc_upper", 240), ovlp_upper=kw.get("ovlp_upper", 120),
                                  bestn=kw.get("bestn", 4))
        assert len(want) > 5000
        monkeypatch.setenv("PGX_DEV_VISIT", "1")
        got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
        assert st["device_replay"] == 1 and st["device_visit"] >= 1, st
        if "mc_upper" in kw:
            assert st["device_visit"] > 1, st   # some groups took the wavefront kernel
        assert formats.ovlp_fields_equal(got, want), kw
        assert st["n_align_needed"] == ost["n_align"] and st["n_seen_skip"] == ost["n_seen_skip"], kw
        monkeypatch.setenv("PGX_DEV_VISIT", "0")
        got0, st0 = rdb.overlap(ix.top, ix.top_mc, **kw)
        assert st0["device_visit"] == 0 and st0["n_buckets"] == st["n_buckets"] and formats.ovlp_fields_equal(got0, want), kw
        # a group beyond what a wavefront replays (here: the limit lowered): the tables left on the device are fetched and the host
        # builds the visit order
        monkeypatch.setenv("PGX_DEV_VISIT", "1"), monkeypatch.setenv("PGX_VISIT_WAVE_MAX", "20")
        got1, st1 = rdb.overlap(ix.top, ix.top_mc, **kw)
        monkeypatch.delenv("PGX_VISIT_WAVE_MAX")
        assert st1["device_visit"] == 0 and st1["n_buckets"] == st["n_buckets"] and formats.ovlp_fields_equal(got1, want), kw
        # the visit order built on the device, then the device replay gives up (tables far too small): the host replay takes over
        # from the join's tables, fetched at that point
        monkeypatch.setenv("PGX_REPLAY_PAIRS_X", "0.0002"), monkeypatch.setenv("PGX_REPLAY_MEMO_X", "0.0002")
        got2, st2 = rdb.overlap(ix.top, ix.top_mc, **kw)
        monkeypatch.delenv("PGX_REPLAY_PAIRS_X"), monkeypatch.delenv("PGX_REPLAY_MEMO_X")
        assert st2["device_replay"] == 0 and formats.ovlp_fields_equal(got2, want), kw
    rdb.close()


def _env(**kw):
    import contextlib

    @contextlib.contextmanager
    def cm():
        saved = {k: os.environ.get(k) for k in kw}
        os.environ.update(kw)
        try:
            yield
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return cm()


def test_index_and_overlap_from_the_packs_and_with_the_bytes_released():
    """Round 6 (VERDICT r5 task 5): once a database has its 2-bit packs the closed-form sketch kernels (k_sketch_blk, and k_sketch_wave for the reads
    it flags: short reads, drops, tie bursts of tandem arrays) and the one-candidate-per-wavefront alignment kernels read THEM, and
    pgx_seqdb_release_bytes gives the bytes' HBM back.  Lists and streams equal the byte forms' and the oracle's before and after the release, under
    the device and the host replay; what needs bytes says so; a database with an ambiguous base keeps its bytes."""
    g = simreads.make_genome(500_000, 19, repeat_families=2, repeat_len=4000, repeat_copies=6, tandem=60)
    db = simreads.simulate_reads(g, seed=47, coverage=24, mean_len=6000, sd_len=3000, min_len=200)
    assert int(db.rlen.min()) < 295 and int(db.rlen.max()) > 12000       # reads shorter than a window + 200 (flag 1) among them
    rdb = ResidentDB(db, 0)
    T = 2
    byte_lists = [rdb.index(total_chunk=T, mychunk=c) for c in (1, 2)]     # no packs yet: the byte kernels
    mm, mc = np.concatenate([p.top for p in byte_lists]), np.concatenate([p.top_mc for p in byte_lists])
    for c, p in enumerate(byte_lists, 1):                                   # ... equal to the oracle's lists
        mine = np.flatnonzero(db.rid % T == c % T)
        want = np.concatenate([U.orc_reduce(U.orc_reduce(U.orc_sketch_seqdb(db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])], 80, 16, int(db.rid[r])), 6), 6) for r in mine])
        assert np.array_equal(p.top, want), c
    want_l1 = {c: np.concatenate([U.orc_reduce(U.orc_sketch_seqdb(db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])], 80, 16, int(db.rid[r])), 6)
                                  for r in np.flatnonzero(db.rid % T == c % T)]) for c in (1, 2)}
    want_ov = {c: U.orc_overlap(db, mm, mc, mychunk=c, total=3)[0] for c in (1, 3)}
    with _env(PGX_ALIGN_PACKED_MIN="0", PGX_GPU_REPLAY="1"):
        ov, st = rdb.overlap(mm, mc, total_chunk=3, mychunk=1)                # the first large launch builds the packs (laid out by locus key)
    assert st["device_replay"] == 1 and len(ov) > 2000 and formats.ovlp_fields_equal(ov, want_ov[1])
    assert rdb.has_bytes

    def check_lists(tag):
        for c, p in enumerate(byte_lists, 1):
            q = rdb.index(total_chunk=T, mychunk=c)
            assert np.array_equal(q.top, p.top), (tag, c)
            assert np.array_equal(formats.mc_as_sorted_pairs(q.top_mc), formats.mc_as_sorted_pairs(p.top_mc)), (tag, c)
            q1 = rdb.index(total_chunk=T, mychunk=c, levels=1)                # the fused form with one reduce level
            assert np.array_equal(q1.top, want_l1[c]), (tag, c, "levels=1")

    def check_streams(tag):
        for env in (dict(PGX_GPU_REPLAY="1"), dict(PGX_GPU_REPLAY="1", PGX_ALIGN_SMALL="0", PGX_ALIGN_PACKED_MIN="0", PGX_ALIGN_ITER_LIMIT="300"),
                    dict(PGX_GPU_REPLAY="0"), dict(PGX_GPU_REPLAY="1", PGX_ALIGN_SMALL="1000000000")):
            with _env(**env):
                for c in (1, 3):
                    got, _ = rdb.overlap(mm, mc, total_chunk=3, mychunk=c)
                    assert formats.ovlp_fields_equal(got, want_ov[c]), (tag, env, c)

    check_lists("packs, bytes still there")                                   # the sketch kernels read the packs now
    with _env(PGX_SKETCH_PACKED="0"):
        check_lists("bytes")
    check_streams("bytes still there")
    # ---- the bytes out of HBM
    import torch
    used0 = torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]
    assert rdb.release_bytes() is True and not rdb.has_bytes
    assert torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0] <= used0       # (the library's copy of the bytes went back to its block cache or the driver)
    assert rdb.release_bytes() is True                                         # (idempotent)
    check_lists("released")
    check_streams("released")
    ix, ov1, st1 = rdb.index_overlap()                                          # the one-chunk pipeline (lists stay in HBM between the stages)
    assert st1["n_records"] == len(ov1) and len(ov1) > 5000
    for call in (lambda: rdb.index(want_l0=True), lambda: rdb.index(window=64), lambda: rdb.index(kmer=15)):   # what needs the bytes says so
        with pytest.raises(_lib.PgxError, match="released"):
            call()
    check_lists("released, after the refused calls")
    rdb.close()
    # ---- a database with an ambiguous base keeps its bytes
    sd = db.seqdb.copy()
    o = int(db.roff[5])
    sd[o + 100] = 0
    sd[o + int(db.rlen[5]) - 1 - 100] &= 0x0F
    dbn = formats.SeqDB(sd, db.rid, db.rlen, db.roff, None)
    rn = ResidentDB(dbn, 0)
    assert rn.release_bytes() is False and rn.has_bytes                         # (the packs are built by the call; read 5 is flagged)
    pn = rn.index(total_chunk=T, mychunk=2)
    wantn = np.concatenate([U.orc_reduce(U.orc_reduce(U.orc_sketch_seqdb(sd[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])], 80, 16, int(db.rid[r])), 6), 6)
                            for r in np.flatnonzero(db.rid % T == 0)])
    assert np.array_equal(pn.top, wantn)
    rn.close()
