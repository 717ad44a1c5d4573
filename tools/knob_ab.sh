#!/bin/bash
# one environment knob over a list of values on a bench workload: tools/knob_ab.sh PGX_ALIGN_SMALL "13000 30000 60000" c4s
K=$1; VALS=$2; W=${3:-c4s}
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache
for v in $VALS; do
  if [ "$v" = default ]; then unset $K; else export $K=$v; fi
  python bench.py --workload $W --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$W $K=$v ms/step', round(d['ms_per_step'],1), 'align', round(k['align']['ms_total']/k['align']['steps'],1), 'align1', round(k.get('align1',{'ms_total':0,'steps':1})['ms_total']/k.get('align1',{'steps':1})['steps'],1), 'n_align_gpu', d['overlap_stats_rank0']['n_align_gpu'])"
done
