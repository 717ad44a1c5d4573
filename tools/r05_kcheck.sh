#!/bin/bash
# does the record stream depend on the replay's schedule?  (round 5: PGX_REPLAY_K = 1 / 2 gave other streams than 3 on c3 / c5s)
# usage: tools/r05_kcheck.sh <tree dir> <workload> "<env A>" "<env B>" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=$1; W=$2; shift 2
for cfg in "$@"; do
  ( cd $T && env $cfg PGX_BENCH_NO_REPLAY_TIMING=1 PGX_BENCH_NO_STREAM_HASH=1 timeout 400 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null ) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['overlap_stats_rank0']
print('$T $W %-28s records %d  needed %d  skips %d  evals %d  sweeps %d  cks %s' % ('$cfg', d['records_per_step'], s['n_align_needed'], s['n_seen_skip'], s['n_evaluations'], s['rounds'], s.get('stream_checksum')))"
done
