"""The CPU oracle against the committed outputs of the REAL reference (tests/golden/).  Runs everywhere (no GPU,
no /root/reference)."""
import os

import numpy as np
import pytest

import golden_util as G
import oracle_util as U
from peregrine_amd import formats


def test_sketch_matches_reference_vectors():
    n = 0
    for i, s, w, k, want in G.sketch_cases():
        got = U.orc_sketch_ascii(s, w, k, 7 + i)
        assert np.array_equal(got, want), f"case {i} w={w} k={k} len={len(s)}"
        n += 1
    assert n > 200


def test_reduce_matches_reference_vectors():
    for i, inp, rs, want in G.reduce_cases():
        assert np.array_equal(U.orc_reduce(inp, rs), want), f"case {i} rs={rs}"


def test_ovlp_match_matches_reference_vectors():
    kinds = set()
    for i, q, qs, t, ts, band, want in G.match_cases():
        got = U.orc_ovlp_match(q, qs, t, ts, band)
        assert got == want, f"case {i}"
        kinds.add(want[3] > 0)
    assert kinds == {True, False}  # both matched and unmatched/band-break cases are covered


def test_seqdb_path_equals_ascii_path():
    db = G.tiny_db()
    for r in (0, 3, 17, 50):
        b = db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])]
        lut = np.full(16, ord("N"), np.uint8)
        lut[[1, 2, 4, 8]] = [ord(c) for c in "ACGT"]
        assert np.array_equal(U.orc_sketch_seqdb(b, 80, 16, r), U.orc_sketch_ascii(lut[b & 15].tobytes(), 80, 16, r))


@pytest.mark.parametrize("T", [1, 2])
def test_index_stage_matches_reference(tmp_path, T):
    z = G.load("tiny_stage.npz")
    db = G.tiny_db(z)
    pre = str(tmp_path / "sd")
    formats.write_seqdb(pre, db)
    assert open(pre + ".idx", "rb").read() == z["idx_text"].tobytes()
    for c in range(1, T + 1):
        for lv in (2, 1):
            o = str(tmp_path / f"o{lv}")
            U.orc_index_chunk(pre, o, T, c, lv, 6, 1, 80, 16)
            tag = f"{c:02d}-of-{T:02d}"
            for L in (("L0", "L2") if lv == 2 else ("L1",)):
                assert np.array_equal(formats.read_mmlist(f"{o}-{L}-{tag}.dat"), z[f"ix{T}l{lv}_{L}_{c}"])
                mc = formats.mc_as_sorted_pairs(formats.read_mm_count(f"{o}-{L}-MC-{tag}.dat"))
                assert np.array_equal(mc, z[f"ix{T}l{lv}_{L}MC_{c}"])


@pytest.mark.parametrize("name", sorted(G.OVERLAP_RUNS))
def test_overlap_stage_matches_reference(tmp_path, name):
    z = G.load("tiny_stage.npz")
    db = G.tiny_db(z)
    IT, lv, OT, kw = G.OVERLAP_RUNS[name]
    pre = str(tmp_path / "sd")
    formats.write_seqdb(pre, db)
    for c in range(1, IT + 1):
        U.orc_index_chunk(pre, str(tmp_path / "ix"), IT, c, lv, 6, 0, 80, 16)
    for c in range(1, OT + 1):
        out = str(tmp_path / f"ov.{c}")
        n, st = U.orc_overlap_chunk(pre, str(tmp_path / f"ix-L{lv}"), out, OT, c, **kw)
        got = formats.read_ovlp(out)
        want = z[f"{name}_{c}"]
        assert len(got) == n == len(want)
        assert formats.ovlp_fields_equal(got, want), name


def test_dedup_matches_reference_text():
    """row f2: oracle's shmr_dedup restatement against the reference binary's stdout on concatenated chunk streams"""
    z = G.load("tiny_stage.npz")
    d = G.load("dedup_cases.npz")
    for name in ("dd_t1", "dd_t2", "dd_t3", "dd_l1"):
        recs = np.concatenate([z[str(k)] for k in d[name + "_keys"]])
        text, nu = U.orc_dedup(recs)
        assert text == d[name].tobytes(), name
        assert nu == text.count(b"\n")


def test_query_helpers_match_reference():
    """row f4: get_shimmer_hits / get_mmer_count / get_shimmers_for_read of the compiled reference (shimmer4py.c)"""
    q, mmers, mc, rlen = G.query_fixture()
    for (c, T, lo, hi) in ((1, 1, 2, 240), (1, 2, 2, 240), (2, 2, 2, 240), (1, 1, 1, 3), (2, 3, 2, 30)):
        tag = f"c{c}t{T}lo{lo}hi{hi}"
        m = U.OrcMap(mmers, mc, rlen, c, T, lo, hi)
        want, off = q[f"hits_{tag}"], q[f"hoff_{tag}"]
        for i, k in enumerate(q["qkeys"]):
            got = m.hits(int(k) >> 8, int(k) & 0xFF)
            assert got.tobytes() == want[off[i]:off[i + 1]].tobytes(), (tag, i)
        if tag == "c1t1lo2hi240":
            assert [m.count(int(k) >> 8) for k in q["qkeys"]] == q["counts"].tolist()
            for r, f, n in zip(q["qrids"], q["read_first"], q["read_count"]):
                gf, gn = m.read_shimmers(r)
                assert gn == n and (n == 0 or gf == f)
        m.close()


def test_map_matches_reference_text():
    """row f3: stdout of the compiled reference's shmr_map for contigs cut from the reads' genome"""
    q, mmers, mc, rlen = G.query_fixture()
    for (c, T, lo, hi) in ((1, 1, 1, 240), (2, 2, 1, 240), (1, 1, 2, 4)):
        text, n = U.orc_map_reads_to_ref(q["ref_l2"], mmers, mc, rlen, c, T, lo, hi)
        want = q[f"map_c{c}t{T}lo{lo}hi{hi}"].tobytes()
        assert text == want and n == want.count(b"\n"), (c, T, lo, hi)


def test_golden_provenance_and_qsort_assumption():
    """The record order of the fixtures (and of the reference itself) rests on qsort being a stable descending sort under the
    reference's 0/1 comparator (shmr_overlap.c:46-50,217): true for glibc's merge-sort path.  The fixtures carry the libc they
    were produced on; the same property is checked on the libc running this test, so a failure here explains a record-order
    mismatch against reference binaries built on this host."""
    import ctypes
    import json
    import os
    info = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "provenance.json")))
    assert info["qsort_with_0_1_comparator_is_stable_descending"] is True and info["libc"].startswith("glibc")
    libc = ctypes.CDLL(None)
    CMP = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32))
    rng = np.random.default_rng(3)
    for n in (5, 120, 3000):
        pos = rng.integers(0, max(2, n // 4), n).astype(np.uint32)
        arr = np.stack([pos, np.arange(n, dtype=np.uint32)], axis=1).copy()
        libc.qsort(arr.ctypes.data_as(ctypes.c_void_p), n, 8, CMP(lambda a, b: 1 if a[0] < b[0] else 0))
        want = arr[np.lexsort((arr[:, 1], -arr[:, 0].astype(np.int64)))]
        assert np.array_equal(arr, want), "this libc's qsort is not stable under the reference's comparator"
