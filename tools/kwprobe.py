import sys, time
sys.path.insert(0, "/root/repo")
from peregrine_amd import simreads, _lib
from peregrine_amd.shimmer import ResidentDB
cfg = dict(simreads.WORKLOADS["ecoli"])
g = simreads.make_genome(cfg.pop("genome_len"), cfg.pop("genome_seed"))
db = simreads.simulate_reads(g, seed=42, **cfg)
rdb = ResidentDB(db, 0)
for (w, k) in ((80, 16), (96, 16), (80, 15), (100, 16), (60, 14), (80, 20)):
    rdb.index(window=w, kmer=k)
    _lib.timing_reset()
    t = time.perf_counter(); ix = rdb.index(window=w, kmer=k); dt = time.perf_counter() - t
    tm = {n: round(_lib.timing(n)[0], 2) for n in ("sketch", "sketch_general", "sketch_literal", "sketch_gather", "reduce", "count")}
    print(f"w={w} k={k}: {dt*1e3:.2f} ms = {db.n_bases/dt/1e9:.1f} Gbases/s, literal reads {ix.reads_literal}/{ix.reads}, L2 {len(ix.top)} kernels ms {tm}")
