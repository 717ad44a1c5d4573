#!/bin/bash
# Round evidence on the GPU box: the graded bench line on the default workload (c3), the same command under
# rocprofv3 --kernel-trace --stats, and the PMC traffic passes.   usage: tools/evidence.sh <tag, e.g. r02a> [workload=c3]
T=${1:-r02}; W=${2:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$T
timeout -k 5 1500 python bench.py --workload $W > gpurun_out/$T/bench_$W.json 2> gpurun_out/$T/bench_$W.err
tail -c 600 gpurun_out/$T/bench_$W.err
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$T/prof_$W -o p -- python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/$T/bench_${W}_under_rocprof.json 2> gpurun_out/$T/rocprof_$W.err
python tools/rocpd_summary.py $(find gpurun_out/$T/prof_$W -name "*.db" | head -1) > gpurun_out/$T/kernel_stats_bench_$W.txt 2>&1
head -40 gpurun_out/$T/kernel_stats_bench_$W.txt
rm -rf gpurun_out/$T/prof_$W      # the database is hundreds of MB; the summary is what is kept
timeout -k 5 1300 bash tools/pmc_traffic.sh $W 2 < /dev/null | tail -60
