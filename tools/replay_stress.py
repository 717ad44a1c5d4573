#!/usr/bin/env python3
"""Race hunt for the multi-threaded replay: the same overlap stage many times with varying thread counts / block sizes,
every output compared field-for-field with the sequential replay's.  usage: tools/replay_stress.py [iterations] [workload]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
wl = sys.argv[2] if len(sys.argv) > 2 else "ecoli"
cfg = dict(simreads.WORKLOADS[wl])
g = simreads.make_genome(cfg.pop("genome_len"), cfg.pop("genome_seed"))
db = simreads.simulate_reads(g, seed=42, **cfg)
rdb = ResidentDB(db, 0)
ix = rdb.index()
os.environ["PGX_THREADS"] = "1"
want, wst = rdb.overlap(ix.top, ix.top_mc)
print(f"{wl}: {len(want)} records sequentially, n_align_needed {wst['n_align_needed']}", flush=True)
os.environ["PGX_PAR_MIN"] = "0"
bad = 0
rng = np.random.default_rng(1)
t0 = time.time()
for it in range(iters):
    thr = int(rng.choice([2, 3, 5, 8, 16, 24, 32, 64]))
    blk = int(rng.choice([1, 3, 16, 64, 200]))
    os.environ["PGX_THREADS"] = str(thr)
    os.environ["PGX_BLOCK"] = str(blk)
    os.environ["PGX_PIN"] = str(int(rng.integers(0, 2)))
    got, st = rdb.overlap(ix.top, ix.top_mc)
    ok = formats.ovlp_fields_equal(got, want) and st["n_align_needed"] == wst["n_align_needed"] and st["n_seen_skip"] == wst["n_seen_skip"]
    if not ok:
        bad += 1
        print(f"MISMATCH iteration {it}: threads {thr} block {blk}: {len(got)} records, stats {st}", flush=True)
print(f"{iters} iterations in {time.time()-t0:.1f}s, {bad} mismatches")
sys.exit(1 if bad else 0)
