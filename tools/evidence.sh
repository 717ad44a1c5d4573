cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01e
timeout -k 5 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r01e/bench_ecoli.json 2> gpurun_out/r01e/bench_ecoli.err
timeout -k 5 300 python bench.py --workload c3 --steps 3 --warmup 1 > gpurun_out/r01e/bench_c3.json 2> gpurun_out/r01e/bench_c3.err
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01e/prof -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01e/bench_under_rocprof.json 2> gpurun_out/r01e/rocprof.err
ls -R gpurun_out/r01e/prof | head -20
python tools/rocpd_summary.py $(find gpurun_out/r01e/prof -name "*.db" | head -1) > gpurun_out/r01e/kernel_stats_bench_ecoli.txt 2>&1
head -30 gpurun_out/r01e/kernel_stats_bench_ecoli.txt
timeout -k 5 400 bash tools/pmc_traffic.sh < /dev/null | tail -40
