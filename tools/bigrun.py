#!/usr/bin/env python3
"""Large-scale probe: BASELINE configs[2]-like set (genome_len x coverage) generated on the GPU, index + overlap,
timings and a few invariants.  usage: tools/bigrun.py [genome_Mb] [coverage]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB

gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 150
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30
t = time.perf_counter()
db = simreads.simulate_reads_torch(int(gmb * 1e6), 1003, cov)
print(f"simulated {db.n_reads} reads {db.n_bases/1e9:.3f} Gbases in {time.perf_counter()-t:.1f}s", flush=True)
t = time.perf_counter(); rdb = ResidentDB(db, 0); print(f"upload {time.perf_counter()-t:.2f}s", flush=True)
for it in range(2):
    _lib.timing_reset()
    t = time.perf_counter(); ix = rdb.index(); ti = time.perf_counter() - t
    ms, n, u = _lib.timing("sketch")
    print(f"index: {ti*1e3:.1f} ms wall = {db.n_bases/ti/1e9:.1f} Gbases/s; sketch kernel {ms:.2f} ms = {u/ms/1e6:.1f} Gbases/s; L2 {len(ix.top)} literal {ix.reads_literal}", flush=True)
os.environ["PGX_TRACE"] = "1"
t = time.perf_counter(); ov, st = rdb.overlap(ix.top, ix.top_mc); to = time.perf_counter() - t
ms, n, u = _lib.timing("align")
ms1, n1, u1 = _lib.timing("align1")
ms, n, u = ms + ms1, n + n1, u + u1
print(f"overlap: {to:.2f}s wall, {len(ov)} records = {len(ov)/to/1e3:.1f} k rec/s; align kernel {ms:.1f} ms / {u} aln = {u/ms/1e3:.2f} M aln/s; stats {st}", flush=True)
pair = np.minimum(ov['y0'] >> 32, ov['y1'] >> 32) << 32 | np.maximum(ov['y0'] >> 32, ov['y1'] >> 32)
print("unique pairs:", len(np.unique(pair)) == len(pair), "types", np.bincount(ov['ovlp_type']))
