#!/usr/bin/env python3
"""Device replay (pgx_replay.hip) against the host replay: the same overlap stage on several simulated sets and parameter
corners, the device run with random window sizes / inner iteration counts (different fixed-point schedules), every output
compared field-for-field with the host's.  usage: tools/devreplay_stress.py [iterations per set] [workload ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
wls = sys.argv[2:] or ["tiny", "repeats", "small", "ecoli"]
rng = np.random.default_rng(7)
bad = 0
t0 = time.time()
for wl in wls:
    seeds = [42] if wl == "c3" else [42, 43]
    for seed in seeds:
        if wl == "tandem":   # many tandem arrays and homopolymer runs: large buckets that hold a read several times
            g = simreads.make_genome(600_000, 9, repeat_families=2, repeat_len=4000, repeat_copies=6, tandem=60)
            db = simreads.simulate_reads(g, seed=seed, coverage=30, mean_len=9000, sd_len=1500)
        elif wl == "repeats":  # planted repeat families + a tandem array: buckets holding a read twice, long reader lists
            g = simreads.make_genome(400_000, 5, repeat_families=3, repeat_len=5000, repeat_copies=4, tandem=1)
            db = simreads.simulate_reads(g, seed=seed, coverage=25, mean_len=9000, sd_len=1500)
        elif wl == "c3":
            c = dict(simreads.WORKLOADS[wl])
            db = simreads.simulate_reads_torch(c["genome_len"], c["genome_seed"], c["coverage"], seed=seed)
        else:
            c = dict(simreads.WORKLOADS[wl])
            g = simreads.make_genome(c.pop("genome_len"), c.pop("genome_seed"))
            db = simreads.simulate_reads(g, seed=seed, **c)
        rdb = ResidentDB(db, 0)
        ix = rdb.index()
        corners = [dict(), dict(bestn=2), dict(bestn=1, ovlp_upper=40), dict(mc_upper=60), dict(total_chunk=2, mychunk=2), dict(align_bandwidth=30)]
        for kw in (corners if wl != "c3" else [dict()]):
            os.environ.pop("PGX_GPU_REPLAY", None)
            want, wst = rdb.overlap(ix.top, ix.top_mc, **kw)
            for it in range(iters if wl != "c3" else 1):
                os.environ["PGX_GPU_REPLAY"] = "1"
                os.environ["PGX_REPLAY_WIN"] = str(int(rng.choice([64, 1024, 16384, 262144, 1 << 22])))
                os.environ["PGX_REPLAY_K"] = str(int(rng.choice([1, 2, 3, 5])))
                os.environ["PGX_REPLAY_BIG"] = str(int(rng.choice([0, 2, 5, 24])))   # 2: nearly every bucket goes to the workgroup kernel
                os.environ["PGX_REPLAY_DUP"] = str(int(rng.choice([0, 2, 2, 12])))   # the same for the buckets that hold a read twice
                got, st = rdb.overlap(ix.top, ix.top_mc, **kw)
                ok = formats.ovlp_fields_equal(got, want) and st["n_align_needed"] == wst["n_align_needed"] and st["n_seen_skip"] == wst["n_seen_skip"]
                if not ok:
                    bad += 1
                    print(f"MISMATCH {wl} seed {seed} {kw}: window {os.environ['PGX_REPLAY_WIN']} k {os.environ['PGX_REPLAY_K']} big {os.environ['PGX_REPLAY_BIG']} dup {os.environ['PGX_REPLAY_DUP']}: {len(got)} vs {len(want)} records, {st} vs {wst}", flush=True)
            print(f"{wl} seed {seed} {kw}: {len(want)} records, ok so far ({bad} mismatches, {time.time()-t0:.0f}s)", flush=True)
print(f"done in {time.time()-t0:.1f}s, {bad} mismatches")
sys.exit(1 if bad else 0)
