#!/usr/bin/env python3
"""Does the order of the candidates matter for a mid-size k_align4 launch (persistent groups pull in index order)?
The bench set's ~53 k alignments in request order, longest-estimate first, and shortest first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB
db = simreads.workload("ecoli") if hasattr(simreads, "workload") else None
if db is None:
    c = dict(simreads.WORKLOADS["ecoli"]); g = simreads.make_genome(c.pop("genome_len"), c.pop("genome_seed")); db = simreads.simulate_reads(g, seed=42, **c)
rdb = ResidentDB(db, 0)
ix = rdb.index()
ov, st = rdb.overlap(ix.top, ix.top_mc)
keys = np.zeros(len(ov), _lib.ALIGN_KEY_DTYPE)
keys["rid0"] = ov["y0"] >> np.uint64(32); keys["rid1"] = ov["y1"] >> np.uint64(32)
p0 = ((ov["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)); p1 = ((ov["y1"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1))
keys["q_off"] = (p0 - p1).astype(np.uint32); keys["dir0"] = ov["strand0"]; keys["dir1"] = ov["strand1"]
est = np.minimum(ov["rl0"].astype(np.int64) - keys["q_off"], ov["rl1"].astype(np.int64))
for name, order in (("request order", np.arange(len(keys))), ("longest first", np.argsort(-est, kind="stable")), ("shortest first", np.argsort(est, kind="stable")),
                    ("random", np.random.default_rng(1).permutation(len(keys)))):
    k = keys[order]
    best = 1e9
    for it in range(4):
        _lib.timing_reset()
        rdb.align(k, 100)
        ms, n, u = _lib.timing("align")
        best = min(best, ms)
    print(f"{name:15s}: {len(k)} alignments, best of 4: {best:.3f} ms = {len(k)/best/1e3:.1f} M aln/s", flush=True)
