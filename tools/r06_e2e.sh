#!/bin/bash
# round 6: the served / end-to-end leg at full size (no CPU leg), with the server's trace; arguments = overlap commands in flight, one run each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for fl in "$@"; do
PGX_BENCH_E2E_INFLIGHT=$fl PGX_BENCH_E2E_TRACE=gpurun_out/r06_e2e_server_trace_$fl.log PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 900 python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu-baseline --end-to-end > gpurun_out/r06_e2e_$fl.json 2> gpurun_out/r06_e2e_$fl.err
python - gpurun_out/r06_e2e_$fl.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); print("resident %.1f ms/step" % d["ms_per_step"], json.dumps(d.get("gpu_end_to_end")))
P
tail -4 gpurun_out/r06_e2e_$fl.err
grep -E "overlap chunk|seqdb load" gpurun_out/r06_e2e_server_trace_$fl.log | head -30
done
