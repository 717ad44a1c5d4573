#!/bin/bash
# SQ counters of k_eval_big on a repeat-rich workload (the simulated set cached outside the profiler): tools/pmc_big.sh [workload=c4s]
W=${1:-c4s}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache
OUT=gpurun_out/pmc_big_$W
rm -rf $OUT; mkdir -p $OUT
timeout 600 python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $OUT/cache.log
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -o p -- python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $OUT/a.log 2>&1
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_GDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -o p -- python bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > $OUT/b.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for nm in ("k_eval_big", "k_eval_rows", "k_eval(", "k_file"):
                if nm in r["Kernel_Name"]:
                    acc[nm][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[nm].add(r["Dispatch_Id"])
    for nm, cs in acc.items():
        for name, v in sorted(cs.items()):
            print(f"{nm:12s} {name:24s} per step: {v:16.0f}   ({len(cnt[nm])} launches)")
PY
find $OUT -type f -size +1M -delete; rm -rf /dev/shm/pgx_bench_cache
