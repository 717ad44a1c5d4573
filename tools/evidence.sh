#!/bin/bash
# Round evidence on the GPU box for one workload: the graded bench line (as the driver runs it for the default workload), then
# tools/pmc_profile.sh (kernel stats + HBM traffic + VALU issue -> profiles/).   usage: tools/evidence.sh <tag, e.g. r04> [workload=c4]
T=${1:-r04}; W=${2:-c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$T profiles
if [ "$W" = c4 ]; then STEPS="--gpus 1 --steps 20 --warmup 5"; else STEPS=""; fi
( time timeout -k 5 2400 python bench.py --workload $W $STEPS > gpurun_out/$T/bench_$W.json ) 2> gpurun_out/$T/bench_$W.err
tail -12 gpurun_out/$T/bench_$W.err
cp gpurun_out/$T/bench_$W.json profiles/${T}_bench_${W}_default.json
bash tools/pmc_profile.sh $W $T 1 | tail -30
