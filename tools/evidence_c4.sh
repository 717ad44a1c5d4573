#!/bin/bash
# c4 evidence in the order that lets the graded line carry the counters of THIS tree: PMC passes first (they rewrite
# profiles/<tag>_{traffic,valu}_c4.json, which bench.py attaches), then the driver's command; everything copied to gpurun_out/prof/
T=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof profiles
bash tools/pmc_profile.sh c4 $T 1 > gpurun_out/prof/pmc_c4.log 2>&1
tail -25 gpurun_out/prof/pmc_c4.log
( time timeout -k 5 2400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/bench_c4_default.json ) 2> gpurun_out/prof/bench_c4_default.log
tail -6 gpurun_out/prof/bench_c4_default.log
cp profiles/${T}_*c4* gpurun_out/prof/
