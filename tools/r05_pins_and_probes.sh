#!/bin/bash
# ONE call on the GPU box: the full-size configs[3] stream pins (the reference on 8 host cores, ~35 min) in the background, and beside it the
# GPU-side evidence of the round's tree: the whole -m gpu suite, a c4 chunk timeline, alignment statistics c4 vs c3, memory-side counters of
# the alignment kernel at c4 vs c3.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( python tests/golden/make_c4_stream_pins.py --workload c4 --out gpurun_out/c4_stream_pins.json > gpurun_out/pins.log 2>&1; echo "pins rc=$?" >> gpurun_out/pins.log ) &
PINS=$!
sleep 150     # (the generator + file write of the pins job use the GPU first)
( timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_pytest_gpu.log )
tail -4 gpurun_out/r05_pytest_gpu.log
bash tools/r05_probe_c4.sh r05e > gpurun_out/r05e_probe.log 2>&1
# alignment statistics (PGX_ALIGN_STATS build): c4 (one step) and c3
for w in c4 c3; do
  PGX_LIB=$PWD/peregrine_amd/libpgx_stats.so PGX_TRACE=1 PGX_BENCH_NO_REPLAY_TIMING=1 PGX_BENCH_NO_STREAM_HASH=1 timeout 300 python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "align stats" | tail -12 > gpurun_out/r05e_align_stats_$w.txt
done
# memory-side counters of k_align_ph: c4 vs c3
for w in c4 c3; do
  OUT=gpurun_out/pmc_alignmem_$w; rm -rf $OUT; mkdir -p $OUT
  i=0
  for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    PGX_BENCH_NO_REPLAY_TIMING=1 PGX_BENCH_NO_STREAM_HASH=1 PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache timeout -k 5 400 rocprofv3 --kernel-trace --kernel-include-regex "k_align_ph" --pmc $grp --output-format csv -d $OUT/$i -o p -- python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$i.json 2> $OUT/$i.err || echo "pass $i failed"
  done
  python - $OUT $w <<'PY' > gpurun_out/r05e_pmc_alignmem_$2.txt 2>&1
import csv, glob, collections, sys
OUT, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = 0.0; nl = 0
for f in glob.glob(OUT + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob(OUT + "/1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; nl += 1
print(w, "k_align_ph launches", nl, "total ms %.1f" % dur)
for k in sorted(acc): print("  %-40s %18.0f  (%d dispatches)" % (k, acc[k], n[k]))
a = acc
if a.get("TCP_TCC_READ_REQ_sum"): print("  L1->L2 read latency (cycles) %.0f" % (a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"]))
if a.get("TCP_UTCL1_REQUEST_sum"): print("  UTCL1 miss rate %.4f" % (a["TCP_UTCL1_TRANSLATION_MISS_sum"] / a["TCP_UTCL1_REQUEST_sum"]))
if a.get("TCC_HIT_sum"): print("  L2 hit rate %.3f" % (a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
if a.get("SQ_WAVE_CYCLES"): print("  waiting %.3f of wave cycles; VALU/SALU per wave %.0f / %.0f" % (a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_INSTS_VALU"] / max(a["SQ_WAVES"], 1), a["SQ_INSTS_SALU"] / max(a["SQ_WAVES"], 1)))
PY
  cat gpurun_out/r05e_pmc_alignmem_$w.txt | tail -8
  find $OUT -type f -size +1M -delete
done
wait $PINS
tail -5 gpurun_out/pins.log
