#!/bin/bash
# round 6: parity of the settle-time fan-out and of the served path, then the c4 line per environment setting (3 steps + 1 warm-up, no CPU leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then shift
timeout -k 5 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu -k "served or replay or overlap_stage or small_dataset or scatter or visit or pipeline_like" > gpurun_out/r06_second_tests.log 2>&1
tail -8 gpurun_out/r06_second_tests.log
fi
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 700 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab2_c4_$i.json 2> gpurun_out/ab2_c4_$i.err
  python - "$cfg" gpurun_out/ab2_c4_$i.json <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2])); k = d["kernels"]
except Exception as e:
    print(sys.argv[1], "FAILED", e); sys.exit(0)
st = d["overlap_stats_rank0"]
print("%-44s %8.1f ms/step  %6.2f M/s  index %6.1f  overlap %7.1f | " % (sys.argv[1], d["ms_per_step"], d["value"] / 1e6, d["index_ms_per_step"], d["overlap_ms_per_step"]) +
      "  ".join("%s %.0f" % (n, v["ms_total"] / v["steps"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_total"] / kv[1]["steps"])[:5]),
      " sweeps %d  evals %.2f M  aligned %.1f M" % (st["rounds"], st["n_evaluations"] / 1e6, st["n_align_gpu"] / 1e6), " pins", d.get("streams_match_pins"),
      " hbm %.1f GB released %s" % (d.get("hbm_bytes_in_use", 0) / 1e9, d.get("hbm_ledger", {}).get("seqdb_bytes_released_after_warmup")))
P
  tail -2 gpurun_out/ab2_c4_$i.err
done
