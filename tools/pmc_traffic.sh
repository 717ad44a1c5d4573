#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes, kernel-trace only), per launch, ON ONE WORKLOAD.
#   usage: tools/pmc_traffic.sh [workload=c3] [steps=2]   ->  gpurun_out/r03_traffic_<workload>.json
# FETCH_SIZE on gfx950 reports exactly 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): doubled below.
W=${1:-c3}; S=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench_$W
# the simulated set is made once, outside the profiler (PGX_BENCH_CACHE): rocprofv3 --pmc aborts inside the torch kernels of the
# repeat-planting genome builder of c4s / c5s ("AQL packet is malformed")
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache
timeout 600 python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT.cache.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python bench.py --workload $W --steps $S --warmup 1 --no-cpu-baseline > $OUT.$c.log 2>&1
done
python - <<PY
import csv, collections, json, re
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"$OUT/{c}/p_counter_collection.csv")):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])   # "void pgx::k_align4<16>(...)" -> k_align4
        if m:
            acc[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, v in res.items():
    # read side: FETCH_SIZE counts 64 B per fabric request; a request moves 128 B for wide coalesced streams (x2: the sketch / pack /
    # join kernels) and 64 B for scattered 8- / 16-byte loads (x1: the alignment kernels' probes; their 16-byte extension loads are
    # dword-aligned pieces of 2-bit packs now, no longer whole lines) -- profiles/r03_fetch_calib.txt.  Write side as reported.
    rf = 1 if k.startswith("k_align") else 2
    v["read_factor"] = rf
    v["hbm_bytes_per_launch"] = (rf * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
res["_workload"] = "$W"
res["_command"] = "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --workload $W --steps $S --warmup 1 --no-cpu-baseline"
json.dump(res, open("gpurun_out/r03_traffic_$W.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k in ("k_align_ph", "k_align1", "k_align1_list", "k_sketch_blk", "k_eval", "k_eval_rows", "k_eval_big", "k_update", "k_file")}, indent=1))
PY
