#!/usr/bin/env python3
"""Steady-state probe of the alignment kernel: one big batch of candidate keys taken from a real overlap run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB
gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 20
db = simreads.simulate_reads_torch(int(gmb * 1e6), 1003, 30.0)
rdb = ResidentDB(db, 0)
ix = rdb.index()
ov, st = rdb.overlap(ix.top, ix.top_mc)
keys = np.zeros(len(ov), _lib.ALIGN_KEY_DTYPE)
keys["rid0"] = ov["y0"] >> np.uint64(32); keys["rid1"] = ov["y1"] >> np.uint64(32)
p0 = ((ov["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)); p1 = ((ov["y1"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1))
keys["q_off"] = (p0 - p1).astype(np.uint32); keys["dir0"] = ov["strand0"]; keys["dir1"] = ov["strand1"]
for it in range(3):
    _lib.timing_reset()
    res = rdb.align(keys, 100)
    ms, n, u = _lib.timing("align")
    print(f"[align] {u} alignments in {ms:.2f} ms = {u/ms/1e3:.2f} M aln/s; mean dist {res['dist'].mean():.1f}", flush=True)
assert np.array_equal(res["q_end"], ov["q_end"])
