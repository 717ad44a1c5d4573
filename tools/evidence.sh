cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r01d
python bench.py --steps 10 --warmup 2 > gpurun_out/r01d/bench_ecoli.json 2> gpurun_out/r01d/bench_ecoli.err
python bench.py --workload c3 --steps 3 --warmup 1 > gpurun_out/r01d/bench_c3.json 2> gpurun_out/r01d/bench_c3.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r01d/prof -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01d/bench_under_rocprof.json 2> gpurun_out/r01d/rocprof.err
ls -R gpurun_out/r01d/prof | head -20
python tools/rocpd_summary.py $(find gpurun_out/r01d/prof -name "*.db" | head -1) > gpurun_out/r01d/kernel_stats_bench_ecoli.txt 2>&1
head -30 gpurun_out/r01d/kernel_stats_bench_ecoli.txt
bash tools/pmc_traffic.sh | tail -30
