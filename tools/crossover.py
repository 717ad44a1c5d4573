#!/usr/bin/env python3
"""Overlap-stage wall time, host replay vs device replay, over a range of set sizes (genome Mb at 30x).
usage: tools/crossover.py [Mb ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

for gmb in [float(x) for x in (sys.argv[1:] or ["2", "5", "10", "20", "40"])]:
    db = simreads.simulate_reads_torch(int(gmb * 1e6), 1003, 30.0)
    rdb = ResidentDB(db, 0)
    ix = rdb.index()
    res = {}
    for mode in ("host", "device"):
        if mode == "device": os.environ["PGX_GPU_REPLAY"] = "1"
        else: os.environ["PGX_GPU_REPLAY"] = "0"
        best = 1e9
        for it in range(4):
            t = time.perf_counter(); ov, st = rdb.overlap(ix.top, ix.top_mc); dt = time.perf_counter() - t
            if it: best = min(best, dt)
        res[mode] = (best, ov, st)
    same = formats.ovlp_fields_equal(res["host"][1], res["device"][1])
    print(f"{gmb:6.1f} Mb: {st['n_pair_records']:9d} pair records, {len(ov):8d} overlaps; host {res['host'][0]*1e3:8.2f} ms, device {res['device'][0]*1e3:8.2f} ms, identical {same}", flush=True)
    del rdb
