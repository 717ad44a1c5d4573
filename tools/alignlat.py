#!/usr/bin/env python3
"""Latency probe of the alignment kernel: batches of 1 .. 64 k candidate keys taken from a real overlap run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB
cfg = dict(simreads.WORKLOADS["ecoli"])
g = simreads.make_genome(cfg.pop("genome_len"), cfg.pop("genome_seed"))
db = simreads.simulate_reads(g, seed=42, **cfg)
rdb = ResidentDB(db, 0)
ix = rdb.index()
ov, st = rdb.overlap(ix.top, ix.top_mc)
keys = np.zeros(len(ov), _lib.ALIGN_KEY_DTYPE)
keys["rid0"] = ov["y0"] >> np.uint64(32); keys["rid1"] = ov["y1"] >> np.uint64(32)
p0 = ((ov["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)); p1 = ((ov["y1"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1))
keys["q_off"] = (p0 - p1).astype(np.uint32); keys["dir0"] = ov["strand0"]; keys["dir1"] = ov["strand1"]
rng = np.random.default_rng(2)
for n in [int(a) for a in sys.argv[1:]] or (1, 32, 512, 2048, 8192, 16384, len(keys)):
    sel = keys[rng.choice(len(keys), n, replace=False)] if n < len(keys) else keys
    rdb.align(sel, 100)
    _lib.timing_reset()
    for _ in range(5):
        res = rdb.align(sel, 100)
    ms, launches, u = _lib.timing("align")
    ms1, l1, u1 = _lib.timing("align1")   # (launches of at most 8 k alignments run k_align1)
    ms, launches, u = ms + ms1, launches + l1, u + u1
    print(f"[alignlat] n={n:6d}: {ms/launches*1e3:8.1f} us per launch = {u/ms/1e3:6.2f} M aln/s", flush=True)
