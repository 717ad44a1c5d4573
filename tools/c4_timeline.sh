#!/bin/bash
# rocprofv3 kernel trace of a c4 step -> chunk timeline (profiles/<tag>_chunk_timeline_c4.txt) and kernel stats
TAG=${1:-r04y}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/c4tl; rm -rf $OUT; mkdir -p $OUT
export PGX_BENCH_NO_REPLAY_TIMING=1
timeout -k 5 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/chunk_timeline.py $f > gpurun_out/${TAG}_chunk_timeline_c4.txt 2>&1
head -50 gpurun_out/${TAG}_chunk_timeline_c4.txt
find $OUT -type f -size +1M -delete
