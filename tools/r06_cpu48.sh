#!/bin/bash
# round 6 (VERDICT r5 task 6b): ONE whole-workload reference leg with 48 processes over 48 + 48 chunks at full-size configs[3] (host memory priced first:
# bench.py's guard; the reference mmaps the seqdb file MAP_SHARED, a process holds ~13 GB of lists and tables)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(nproc; free -g; df -h /dev/shm) > gpurun_out/r06_cpu48_host.txt 2>&1
PGX_BENCH_CPU_PROCS=48 PGX_BENCH_BUDGET_S=4000 PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 2400 python bench.py --workload c4 --steps 2 --warmup 1 --cpu-baseline full --no-end-to-end > gpurun_out/r06_bench_c4_cpu48.json 2> gpurun_out/r06_bench_c4_cpu48.err
echo "rc $?"; tail -12 gpurun_out/r06_bench_c4_cpu48.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r06_bench_c4_cpu48.json")); cb = d["cpu_baseline"]
print(d["ms_per_step"], d["value"], d.get("streams_match_pins"), {k: cb.get(k) for k in ("value", "cores", "mode", "chunking", "index_s", "overlap_s", "records", "records_match_gpu", "host_ram_gb", "procs_limit_reason")}, d.get("gpu_over_cpu"))
P
