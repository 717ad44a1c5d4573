#!/usr/bin/env python3
"""Parity + speed probe of the lane-per-candidate alignment kernel (pgx_align_lane.hip) against the byte-wise kernels and the
oracle: keys of a real overlap run + unrelated read pairs (band breaks, no match) + short / empty queries, bands 20 / 100 / 130.
usage: tools/lanecheck.py [genome_Mb=4.6] [n_oracle=300]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB
gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.6
n_orc = int(sys.argv[2]) if len(sys.argv) > 2 else 300
db = simreads.simulate_reads_torch(int(gmb * 1e6), 1003, 30.0)
rdb = ResidentDB(db, 0)
ix = rdb.index()
ov, st = rdb.overlap(ix.top, ix.top_mc)
keys = np.zeros(len(ov), _lib.ALIGN_KEY_DTYPE)
keys["rid0"] = ov["y0"] >> np.uint64(32); keys["rid1"] = ov["y1"] >> np.uint64(32)
p0 = ((ov["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)); p1 = ((ov["y1"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1))
keys["q_off"] = (p0 - p1).astype(np.uint32); keys["dir0"] = ov["strand0"]; keys["dir1"] = ov["strand1"]
rng = np.random.default_rng(7)
nx = max(2000, len(keys) // 50)
extra = np.zeros(nx, _lib.ALIGN_KEY_DTYPE)          # unrelated pairs, random offsets and strands, some queries at the very end of the read
extra["rid0"] = rng.integers(0, db.n_reads, nx); extra["rid1"] = rng.integers(0, db.n_reads, nx)
rl = db.rlen[extra["rid0"]]
extra["q_off"] = np.where(rng.random(nx) < 0.1, rl - rng.integers(0, 40, nx).clip(0, rl), rng.integers(0, rl))
extra["dir0"] = rng.integers(0, 2, nx); extra["dir1"] = rng.integers(0, 2, nx)
allk = np.concatenate([keys, extra])
allk = allk[rng.permutation(len(allk))]
print(f"{len(keys)} keys of accepted overlaps + {nx} unrelated pairs", flush=True)
bad = 0
for band in (100, 20, 130):
    os.environ["PGX_ALIGN_LANE_MIN"] = "-1"
    old = rdb.align(allk, band)
    os.environ["PGX_ALIGN_LANE_MIN"] = "0"
    _lib.timing_reset()
    new = rdb.align(allk, band)
    same = np.array_equal(old, new)
    nd = int((old != new).sum()) if not same else 0
    print(f"band {band}: lane kernel == byte-wise kernels on {len(allk)} keys: {same}" + ("" if same else f" ({nd} differ)"), flush=True)
    if not same:
        bad += 1
        w = np.flatnonzero(old != new)[:8]
        import oracle_util as U
        for i in w:
            a, b = int(allk["rid0"][i]), int(allk["rid1"][i])
            q = db.seqdb[int(db.roff[a]) + int(allk["q_off"][i]):int(db.roff[a]) + int(db.rlen[a])]
            t = db.seqdb[int(db.roff[b]):int(db.roff[b]) + int(db.rlen[b])]
            print("  key", allk[i], "old", old[i], "new", new[i], "oracle", U.orc_ovlp_match(q, int(allk["dir0"][i]), t, int(allk["dir1"][i]), band))
if n_orc:
    import oracle_util as U
    sel = rng.choice(len(allk), min(n_orc, len(allk)), replace=False)
    os.environ["PGX_ALIGN_LANE_MIN"] = "0"
    got = rdb.align(allk, 100)[sel]
    ok = True
    for j, i in enumerate(sel):
        a, b = int(allk["rid0"][i]), int(allk["rid1"][i])
        q = db.seqdb[int(db.roff[a]) + int(allk["q_off"][i]):int(db.roff[a]) + int(db.rlen[a])]
        t = db.seqdb[int(db.roff[b]):int(db.roff[b]) + int(db.rlen[b])]
        want = U.orc_ovlp_match(q, int(allk["dir0"][i]), t, int(allk["dir1"][i]), 100)
        if tuple(int(v) for v in got[j].tolist()) != want:
            ok = False
            print("  oracle mismatch: key", allk[i], "got", got[j], "want", want)
    bad += not ok
    print(f"lane kernel == oracle on {len(sel)} sampled keys: {ok}", flush=True)
# speed, the accepted-overlap keys only (what a replay sweep asks for)
for mode, name in (("-1", "byte-wise (k_align_ph / k_align1)"), ("0", "lane-per-candidate")):
    os.environ["PGX_ALIGN_LANE_MIN"] = mode
    os.environ["PGX_TRACE"] = "1"
    rdb.align(keys, 100)
    os.environ.pop("PGX_TRACE")
    _lib.timing_reset()
    t = time.perf_counter()
    for _ in range(3): rdb.align(keys, 100)
    wall = (time.perf_counter() - t) / 3
    ms = sum(_lib.timing(k)[0] for k in ("align", "align1")) / 3
    pk = _lib.timing("align_pack")[0] / 3
    print(f"[{name}] {len(keys)} alignments: kernels {ms:.2f} ms (+ pack {pk:.2f} ms) = {len(keys)/ms/1e3:.2f} M aln/s; call wall {wall*1e3:.1f} ms", flush=True)
print("LANECHECK", "FAILED" if bad else "ok")
