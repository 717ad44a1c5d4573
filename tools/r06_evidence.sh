#!/bin/bash
# round 6 evidence on the final tree: (1) tools/pmc_profile.sh c4 r06 (kernel stats, FETCH_SIZE / WRITE_SIZE, SQ passes of the bench command -> profiles/r06_{kernel_stats_bench,
# traffic,valu}_c4.*), (2) the memory-side counters of k_align_ph at full size (UTCL1 hit / miss, L2, latency, waits) -> profiles/r06_pmc_align_memory_c4.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out profiles
bash tools/pmc_profile.sh c4 r06 2 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=c4
OUT=gpurun_out/pmc_alignmem_$w; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  PGX_BENCH_NO_REPLAY_TIMING=1 PGX_BENCH_NO_STREAM_HASH=1 timeout -k 5 400 rocprofv3 --kernel-trace --kernel-include-regex "k_align_ph" --pmc $grp --output-format csv -d $OUT/$i -o p -- python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$i.json 2> $OUT/$i.err || echo "pass $i failed"
done
python - $OUT $w <<'PY' > profiles/r06_pmc_align_memory_c4.txt 2>&1
import csv, glob, collections, sys
OUT, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = 0.0; nl = 0
for f in glob.glob(OUT + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for f in glob.glob(OUT + "/1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; nl += 1
print("# memory-side and SQ counters of k_align_ph<8, u16, packed> at full-size configs[3] (c4: 47 GB of 2-bit packs laid out by locus key, requests in layout order,")
print("# workgroups of 16 wavefronts), one step, separate rocprofv3 --pmc passes with --kernel-trace only (tools/r06_evidence.sh); round 5's figures: profiles/r05e_pmc_align_memory_c4_vs_c3.txt")
print(w, "k_align_ph launches", nl, "total ms %.1f" % dur)
for k in sorted(acc): print("  %-40s %18.0f  (%d dispatches)" % (k, acc[k], n[k]))
a = acc
if a.get("TCP_TCC_READ_REQ_sum"): print("  L1->L2 read latency (cycles) %.0f" % (a["TCP_TCC_READ_REQ_LATENCY_sum"] / a["TCP_TCC_READ_REQ_sum"]))
if a.get("TCP_UTCL1_REQUEST_sum"): print("  UTCL1 (per-CU TLB) miss rate %.4f   (misses / requests; hits / requests %.4f)" % (a["TCP_UTCL1_TRANSLATION_MISS_sum"] / a["TCP_UTCL1_REQUEST_sum"], a["TCP_UTCL1_TRANSLATION_HIT_sum"] / a["TCP_UTCL1_REQUEST_sum"]))
if a.get("TCC_HIT_sum"): print("  L2 hit rate %.3f" % (a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
if a.get("SQ_WAVE_CYCLES"): print("  wavefronts waiting %.3f of their cycles; VALU issue per cycle and SIMD %.3f" % (a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], a["SQ_INSTS_VALU"] / 1024 / (a["SQ_BUSY_CYCLES"] / 32)))
PY
cat profiles/r06_pmc_align_memory_c4.txt | tail -12
find $OUT -type f -size +1M -delete
# (only gpurun_out/ travels back from the GPU box: the summaries written under profiles/ go along in a directory of their own)
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06_kernel_stats_bench_c4.txt profiles/r06_traffic_c4.json profiles/r06_valu_c4.json profiles/r06_bench_c4_nocpu.json profiles/r06_pmc_align_memory_c4.txt gpurun_out/profiles_r06/ 2>/dev/null
ls -la gpurun_out/profiles_r06
