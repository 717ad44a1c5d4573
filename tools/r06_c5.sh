#!/bin/bash
# round 6: BASELINE configs[4] at full size with its reference leg (8 whole overlap chunks of the job's 24, hashed -> the pins' source)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(nproc; free -g; grep -E "MemTotal|MemAvailable" /proc/meminfo; df -h /dev/shm) > gpurun_out/r06_host.txt 2>&1
PGX_BENCH_BUDGET_S=4000 timeout -k 5 2600 python bench.py --workload c5 --steps 3 --warmup 1 --cpu-baseline whole_chunks > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err
echo "rc $?"; tail -25 gpurun_out/r06_bench_c5.err; cat gpurun_out/r06_host.txt
