#!/bin/bash
# A/B of library builds on the bench workloads: LIBS="base new x1 ..." -> peregrine_amd/libpgx_<name>.so (new = libpgx.so); WL="c3 c4s"
mkdir -p gpurun_out
if [ -z "$NOPARITY" ]; then
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "align_variants or ovlp_match_symbol or long_reads" > gpurun_out/ab_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/ab_parity.log
tail -3 gpurun_out/ab_parity.log
fi
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache
for w in ${WL:-c3 c4s}; do
for lib in ${LIBS:-base new}; do
  unset PGX_ALIGN_Q
  if [ $lib = new ]; then unset PGX_LIB; else export PGX_LIB=$PWD/peregrine_amd/libpgx_$lib.so; fi
  timeout 400 python bench.py --workload $w --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline > gpurun_out/ab_${w}_$lib.json 2> gpurun_out/ab_${w}_$lib.err
  python - <<P
import json
d=json.load(open("gpurun_out/ab_${w}_$lib.json"))
k=d["kernels"]
print("$w $lib ms/step", round(d["ms_per_step"],1), "align ms/step", round(k["align"]["ms_total"]/k["align"]["steps"],1), "aln/s", round(d["roofline_all"]["align"]["alignments_per_s"]/1e6,1), "nrec", d["records_per_step"])
P
done; done
