#!/bin/bash
# round 6: the driver's command on the final tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
date > gpurun_out/r06_bench_c4_default.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_c4_default.json ) 2>> gpurun_out/r06_bench_c4_default.log
echo "rc $?"; tail -30 gpurun_out/r06_bench_c4_default.log
python - <<'P'
import json
d = json.load(open("gpurun_out/r06_bench_c4_default.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "index_ms_per_step", "overlap_ms_per_step", "streams_match_pins", "hbm_bytes_in_use")})
cb = d.get("cpu_baseline", {}); print({k: cb.get(k) for k in ("value", "cores", "mode", "index_s", "overlap_s", "records_match_gpu", "cgroup_limit_gb", "fallback_reason")})
print(d.get("gpu_over_cpu")); print(d.get("gpu_end_to_end")); print(d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("valu"))
P
