"""Parity at BASELINE.json sizes through size-independent properties (the oracle cannot finish these sizes in seconds):
 * configs[2]-like set: 150 Mb genome x 30x, ~300 k reads, 4.5 Gbases, generated on the GPU;
 * a 10 Mb x 25x set that the oracle CAN finish is compared record-for-record.
Properties: sampled reads' L0 equal the oracle's; chunked index runs concatenate to the single-chunk lists; every
sampled ovlp_t record re-derives bit-exactly from the oracle's ovlp_match and passes the reference's acceptance test;
each read pair is reported once per chunk; the stage is idempotent; overlap chunks partition the single-chunk buckets."""
import numpy as np
import pytest

import oracle_util as U
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

pytestmark = pytest.mark.gpu


def _read(db, r):
    return db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])]


def test_medium_set_full_oracle_comparison():
    db = simreads.simulate_reads_torch(10_000_000, 77, 25.0, seed=9)   # ~16.7 k reads, 250 Mbases
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    ov, st = rdb.overlap(ix.top, ix.top_mc)
    l2 = []
    for r in range(db.n_reads):
        l2.append(U.orc_reduce(U.orc_reduce(U.orc_sketch_seqdb(_read(db, r), 80, 16, r), 6), 6))
    l2 = np.concatenate(l2)
    assert np.array_equal(ix.top, l2)
    want, ost = U.orc_overlap(db, l2, U.orc_count(l2))
    assert len(want) > 100_000 and formats.ovlp_fields_equal(ov, want)
    assert st["n_align_needed"] == ost["n_align"]
    rdb.close()


def test_c3_scale_properties():
    db = simreads.simulate_reads_torch(150_000_000, 1003, 30.0)       # BASELINE configs[2]: ~300 k reads, 4.5 Gbases
    assert db.n_bases > 4.4e9
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    rng = np.random.default_rng(5)
    # (1) L2 of sampled reads equals the oracle's; the list is grouped by rid in idx order and position-sorted
    rid = (ix.top["y"] >> np.uint64(32)).astype(np.int64)
    assert np.all(np.diff(rid) >= 0)
    pos = ((ix.top["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
    same = np.diff(rid) == 0
    assert np.all(np.diff(pos)[same] > 0)
    starts = np.searchsorted(rid, np.arange(db.n_reads + 1))
    for r in rng.integers(0, db.n_reads, 200):
        want = U.orc_reduce(U.orc_reduce(U.orc_sketch_seqdb(_read(db, r), 80, 16, int(r)), 6), 6)
        assert np.array_equal(ix.top[starts[r]:starts[r + 1]], want), int(r)
    # (2) counts: sum of multiplicities == list length; sorted unique mers
    assert int(ix.top_mc["count"].sum()) == len(ix.top) and np.all(np.diff(ix.top_mc["mer"].astype(np.int64)) > 0)
    # (3) chunked index == single chunk, re-ordered (rid % T == c % T)
    parts = [rdb.index(total_chunk=3, mychunk=c) for c in (1, 2, 3)]
    cat = np.concatenate([p.top for p in parts])
    assert len(cat) == len(ix.top)
    order = np.argsort((cat["y"] >> np.uint64(32)).astype(np.int64), kind="stable")
    assert np.array_equal(cat[order], ix.top)
    # (4) overlap stage: sampled records re-derive from the oracle's ovlp_match and pass the acceptance rule
    ov, st = rdb.overlap(ix.top, ix.top_mc)
    assert len(ov) > 4_000_000 and st["n_records"] == len(ov)
    r0 = (ov["y0"] >> np.uint64(32)).astype(np.int64)
    r1 = (ov["y1"] >> np.uint64(32)).astype(np.int64)
    pair = np.minimum(r0, r1) << 32 | np.maximum(r0, r1)
    assert len(np.unique(pair)) == len(pair)                       # first-wins per chunk (process-global seen table)
    assert np.array_equal(ov["rl0"], db.rlen[r0]) and np.array_equal(ov["rl1"], db.rlen[r1])
    for i in rng.integers(0, len(ov), 300):
        o = ov[i]
        p0 = ((int(o["y0"]) & 0xFFFFFFFF) >> 1) + 1
        p1 = ((int(o["y1"]) & 0xFFFFFFFF) >> 1) + 1
        assert p0 >= p1
        q = _read(db, r0[i])[p0 - p1:]
        t = _read(db, r1[i])
        m = U.orc_ovlp_match(q, int(o["strand0"]), t, int(o["strand1"]), 100)
        assert m == tuple(int(o[f]) for f in formats.MATCH_FIELDS), int(i)
        q_bgn, q_end, t_bgn, t_end = m[2], m[3], m[4], m[5]
        assert q_bgn < 48 and t_bgn < 48 and (abs(len(q) - q_end) < 48 or abs(len(t) - t_end) < 48) and q_end > 500 and t_end > 500
        contain = abs(int(o["rl0"]) - (q_end - q_bgn)) < 96 or abs(int(o["rl1"]) - (t_end - t_bgn)) < 96
        assert int(o["ovlp_type"]) == ((1 if o["rl0"] >= o["rl1"] else 2) if contain else 0)
    # (5) idempotence, and the multi-threaded replay reaches the same fixed point as the sequential one
    import os
    ov2, st2 = rdb.overlap(ix.top, ix.top_mc)
    assert formats.ovlp_fields_equal(ov, ov2)
    os.environ["PGX_THREADS"] = "1"
    try:
        ov1, st1 = rdb.overlap(ix.top, ix.top_mc)
    finally:
        del os.environ["PGX_THREADS"]
    assert formats.ovlp_fields_equal(ov, ov1)
    assert st1["n_align_needed"] == st["n_align_needed"] == st2["n_align_needed"]
    assert st1["n_seen_skip"] == st["n_seen_skip"]
    rdb.close()
