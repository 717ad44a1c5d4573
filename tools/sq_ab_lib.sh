#!/bin/bash
# SQ counters of the alignment kernels for two builds of the library (A/B of a kernel change): LIBS="base new" -> peregrine_amd/libpgx_base.so / libpgx.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${W:-c3}
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache PGX_BENCH_NO_REPLAY_TIMING=1
CMD="python bench.py --workload $W --steps 2 --warmup 0 --no-cpu-baseline"
timeout 600 $CMD > /dev/null 2>&1
for lib in ${LIBS:-base new}; do
  if [ $lib = new ]; then unset PGX_LIB; else export PGX_LIB=$PWD/peregrine_amd/libpgx_$lib.so; fi
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" ${GROUPS_EXTRA}; do
    i=$((i+1)); OUT=gpurun_out/sqlib_${lib}_$i; rm -rf $OUT
    timeout -k 5 600 rocprofv3 --kernel-trace --kernel-include-regex "k_align" --pmc $grp --output-format csv -d $OUT -o p -- $CMD > $OUT.json 2> $OUT.err || echo "pass $i ($grp) failed: $(tail -2 $OUT.err | tr '\n' ' ')"
    python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-40:]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in acc: print("$lib", k, "ms=%.1f"%dur[k], {c: "%.4g"%v for c,v in acc[k].items()})
PY
    find $OUT -type f -size +1M -delete
  done
done
