#!/bin/bash
# round 6: alignment parity tests of the new kernel form, then the c4 line (3 steps + 1 warm-up, no CPU leg) per environment setting
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then shift
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "align or ovlp_match or overlap_stage_with_long or small_dataset_overlap or device_replay" > gpurun_out/r06_align_tests.log 2>&1
tail -5 gpurun_out/r06_align_tests.log
fi
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 700 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_c4_$i.json 2> gpurun_out/ab_c4_$i.err
  python - "$cfg" gpurun_out/ab_c4_$i.json <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[2])); k = d["kernels"]
except Exception as e:
    print(sys.argv[1], "FAILED", e); sys.exit(0)
print("%-40s %8.1f ms/step  %6.2f M/s  index %6.1f  overlap %7.1f | " % (sys.argv[1], d["ms_per_step"], d["value"] / 1e6, d["index_ms_per_step"], d["overlap_ms_per_step"]) +
      "  ".join("%s %.0f" % (n, v["ms_total"] / v["steps"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_total"] / kv[1]["steps"])[:7]), " hbm %.1f GB" % (d.get("hbm_bytes_in_use", 0) / 1e9), " pins", d.get("streams_match_pins"), " aln/s %.1f M" % (d["roofline_all"]["align"]["alignments_per_s"] / 1e6))
P
  tail -3 gpurun_out/ab_c4_$i.err
done
