#!/bin/bash
# round-3 evidence collection on the GPU box (one gpurun call): microbenchmarks, the RCCL path as a one-rank job, and the
# repeat-rich workloads' bench lines + kernel traces.  usage: tools/r03_evidence.sh <tag> [parts...]
TAG=${1:-r03a}; shift
PARTS=${@:-"valu calib rccl c4s c5s"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
prof() {  # prof <name> <bench args...>: rocprofv3 kernel trace of a bench command, summarised
  local name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout -k 5 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o p -- python bench.py "$@" --no-cpu-baseline > gpurun_out/${TAG}_bench_${name}_under_rocprof.json 2> gpurun_out/prof_$name.err
  local db=$(find gpurun_out/prof_$name -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/rocpd_summary.py "$db" gpurun_out/${TAG}_kernel_stats_bench_$name.txt > /dev/null; fi
  find gpurun_out/prof_$name -type f -size +2M -delete
}
for p in $PARTS; do
  case $p in
    valu)  timeout 600 tools/build/valu_issue > gpurun_out/${TAG}_valu_issue.txt 2>&1 ;;
    calib) timeout 900 tools/fetch_calib.sh > gpurun_out/${TAG}_fetch_calib.txt 2>&1 ;;
    rccl)  timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q > gpurun_out/${TAG}_pytest_parallel.txt 2>&1
           PGX_FORCE_EXCHANGE=1 timeout 900 python bench.py --workload ecoli --steps 5 --warmup 1 > gpurun_out/${TAG}_bench_forced_rccl_ecoli.json 2> gpurun_out/${TAG}_bench_forced_rccl_ecoli.err ;;
    c3|c4s|c5s|ecoli)
           timeout 900 python bench.py --workload $p --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_$p.json 2> gpurun_out/${TAG}_bench_$p.err
           prof $p --workload $p --steps 2 --warmup 1 ;;
    traffic_*) W=${p#traffic_}; timeout 1000 tools/pmc_traffic.sh $W 1 > gpurun_out/${TAG}_traffic_$W.log 2>&1; find gpurun_out/pmc_bench_$W -type f -size +1M -delete ;;
    pmc_lane) timeout 1300 tools/pmc_lane.sh 150 > gpurun_out/${TAG}_pmc_align_lane.txt 2>&1 ;;
    pmc_align) timeout 900 tools/pmc_align_c3.sh 8 > gpurun_out/${TAG}_pmc_align.txt 2>&1; find gpurun_out/pmc_align_m8 -type f -size +1M -delete ;;
    full_c3) timeout 1200 python bench.py > gpurun_out/${TAG}_bench_c3_full.json 2> gpurun_out/${TAG}_bench_c3_full.err ;;
  esac
done
ls -la gpurun_out | tail -30
