#!/bin/bash
# SQ / TCP / TCC / TA counters of the alignment kernels at c3, one --pmc group per pass (profiles/r04v_align_counters_c3.txt; QS="0 1 2" walked the
# forms of commit 269861a: PGX_ALIGN_Q=1 k_align_q, 2 with LDS windows -- the knob does nothing in later trees)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache PGX_BENCH_NO_REPLAY_TIMING=1
CMD="python bench.py --workload c3 --steps 2 --warmup 0 --no-cpu-baseline"
timeout 600 $CMD > /dev/null 2>&1
for q in ${QS:-0}; do
  export PGX_ALIGN_Q=$q
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_ACCESSES_sum"; do
    i=$((i+1)); OUT=gpurun_out/sq_ab_${q}_$i; rm -rf $OUT
    timeout -k 5 600 rocprofv3 --kernel-trace --kernel-include-regex "k_align_(ph|q)" --pmc $grp --output-format csv -d $OUT -o p -- $CMD > $OUT.json 2> $OUT.err || echo "pass $i ($grp) failed: $(tail -2 $OUT.err | tr '\n' ' ')"
    python - <<PY
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); dur=collections.defaultdict(float)
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-40:]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in acc: print("Q=$q", k, "ms=%.1f"%dur[k], {c: "%.4g"%v for c,v in acc[k].items()})
PY
    find $OUT -type f -size +1M -delete
  done
done
