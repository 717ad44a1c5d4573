#!/bin/bash
# replay schedule knobs on one workload: ms per step of bench.py under each setting (the simulated set is cached once).  Round 4 ran it with
# every schedule knob of round 3 (profiles/r04f_knob_sweep_c4s.txt: all within noise) and then turned all but the ones below into constants.
#   usage: tools/knob_sweep.sh [workload=c4s] [steps=3]  -> gpurun_out/knob_sweep_<workload>.txt
W=${1:-c4s}; S=${2:-3}
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache PGX_BENCH_NO_REPLAY_TIMING=1
OUT=gpurun_out/knob_sweep_$W.txt
: > $OUT
run() {
  local tag="$1"; shift
  local line
  line=$(env "$@" python bench.py --workload $W --steps $S --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$tag" <<PY >> $OUT
import json, sys
try:
    p = json.loads('''$line''')
    st = p["overlap_stats_rank0"]
    print("%-44s %8.1f ms/step  overlap %8.1f  evals %9d  sweeps %3d  aligned %9d  records %d" % (sys.argv[1], p["ms_per_step"], p["overlap_ms_per_step"], st["n_evaluations"], st["rounds"], st["n_align_gpu"], p["records_per_step"]))
except Exception as e:
    print("%-44s FAILED %s" % (sys.argv[1], e))
PY
}
run baseline X=1
run baseline-again X=1
run WIN=524288 PGX_REPLAY_WIN=524288
run WIN=131072 PGX_REPLAY_WIN=131072
run K=2 PGX_REPLAY_K=2
run K=4 PGX_REPLAY_K=4
run DUP=24 PGX_REPLAY_DUP=24
run DUP=6 PGX_REPLAY_DUP=6
run BIG=48 PGX_REPLAY_BIG=48
cat $OUT
