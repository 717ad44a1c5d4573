#!/bin/bash
# SQ counters of the alignment kernels on tools/lanecheck.py's c3-size batch (4.68 M alignments per launch).  usage: tools/pmc_lane.sh [genome_Mb=150]
G=${1:-150}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lane
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 420 rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python tools/lanecheck.py $G 0 > $OUT/trace.log 2>&1
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python tools/lanecheck.py $G 0 > $OUT/a.log 2>&1
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/b -o p -- python tools/lanecheck.py $G 0 > $OUT/b.log 2>&1
python - <<PY
import csv, collections, glob, sqlite3
for db in glob.glob("$OUT/t/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    for name, calls, tot, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if "k_align" in name or "k_pack2" in name or "k_lane" in name:
            print("%-60s calls %4d avg %10.1f us" % (name.split("(")[0][-60:], calls, avg / 1e3))
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(f"$OUT/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for kn in ("k_align_lane", "k_align_ph"):
                if kn in r["Kernel_Name"]:
                    acc[kn][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for kn, cs in acc.items():
        for name, disp in sorted(cs.items()):
            vals = list(disp.values())
            big = [v for v in vals if v > 0.5 * max(vals)] or vals
            print(f"{kn:14s} {name:24s} per big launch: {sum(big)/len(big):18.0f}   ({len(big)} launches)")
PY
find $OUT -type f -size +1M -delete
