#!/bin/bash
# The evidence of the tree round 5 ends with.  (In round 5 the steps were finally run as SEPARATE short calls: a call dies with its client.)
# In ONE call (order: counters first, so that the graded line attaches the counters of THIS tree):
#   1. tools/pmc_profile.sh c4 r05 1      -> profiles/r05_{kernel_stats_bench,traffic,valu}_c4.*
#   2. the driver's command               -> profiles/r05_bench_c4_default.{json,log}   (whole-workload reference CPU leg, hashes, pins)
#   3. configs[4] at full size, twice     -> profiles/r05_bench_c5_run{1,2}.json        (run 1 with the sample CPU leg)
#   4. the end-to-end leg at c4           -> profiles/r05_bench_c4_end_to_end.json
#   5. the one-chunk shapes               -> profiles/r05_bench_{c3,c4s,c5s}.json
#   6. pytest -m gpu                      -> profiles/r05_pytest_gpu.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/prof; mkdir -p $P profiles
bash tools/pmc_profile.sh c4 r05 1 > $P/pmc_c4.log 2>&1; tail -12 $P/pmc_c4.log
cp profiles/r05_*c4* $P/ 2>/dev/null
( time timeout -k 5 1790 python bench.py --gpus 1 --steps 20 --warmup 5 > $P/r05_bench_c4_default.json ) 2> $P/r05_bench_c4_default.log; tail -4 $P/r05_bench_c4_default.log
( time timeout -k 5 1500 python bench.py --workload c5 --steps 3 --warmup 1 --cpu-baseline sample > $P/r05_bench_c5_run1.json ) 2> $P/r05_bench_c5_run1.log; tail -3 $P/r05_bench_c5_run1.log
( time timeout -k 5 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $P/r05_bench_c5_run2.json ) 2> $P/r05_bench_c5_run2.log; tail -3 $P/r05_bench_c5_run2.log
( time timeout -k 5 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --end-to-end > $P/r05_bench_c4_end_to_end.json ) 2> $P/r05_bench_c4_end_to_end.log; tail -3 $P/r05_bench_c4_end_to_end.log
for w in c3 c4s c5s; do timeout 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $P/r05_bench_$w.json 2> $P/r05_bench_$w.log; done
timeout 1100 python -m pytest tests -m gpu -q --durations=15 > $P/r05_pytest_gpu.log 2>&1; tail -3 $P/r05_pytest_gpu.log
# (7. the whole-workload CPU leg as 48 processes over 48 chunks -- PGX_BENCH_CPU_PROCS=48 -- is NOT part of this script any more: 48 x ~13 GB of
#  reference processes beside the 93 GB seqdb file in /dev/shm took the GPU box down in round 5; bench.py now checks the host's memory first.)
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/prof/r05_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    cb = d.get("cpu_baseline") or {}
    print("%-46s %9.1f ms/step %7.2f M/s  hbm %.1f GB  pins %s  cks-equal %s  cpu %s %s  e2e %s" % (f.split("/")[-1], d["ms_per_step"], d["value"] / 1e6, d.get("hbm_bytes_in_use", 0) / 1e9,
          d.get("streams_match_pins"), d.get("stream_checksums_equal_in_every_timed_step"), cb.get("mode"), ("%.0f k/s match %s" % (cb["value"] / 1e3, cb.get("records_match_gpu"))) if cb.get("value") else "", (d.get("gpu_end_to_end") or {}).get("overlaps_per_s")))
PY
