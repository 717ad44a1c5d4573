#!/bin/bash
# SQ counters of the alignment kernels on a bench step of any workload: occupancy over the launch (is a launch waiting for a few long
# candidates?) and the instruction counts.  usage: tools/pmc_align_any.sh <workload> [tag]
W=${1:-c4s}; TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_align_$W
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -o p -- python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
python - <<PY > gpurun_out/${TAG}_pmc_align_$W.txt
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
dur = {}
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_align" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"]]["name"] = r["Kernel_Name"].split("(")[0][-40:]
            acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("# alignment launches of one bench step ($W): duration, wavefronts, VALU / SALU wavefront-instructions, and the mean number of resident wavefronts")
print("# (mean resident wavefronts = 4 x SQ_WAVE_CYCLES / (duration x 2.1 GHz), approximate: far below the launch's grid => the launch spent its time waiting for a few long candidates)")
print("%-44s %9s %9s %14s %14s %12s" % ("kernel", "ms", "waves", "VALU", "SALU", "mean waves"))
for d, c in sorted(acc.items(), key=lambda kv: int(kv[0])):
    if dur.get(d, 0) < 0.2: continue
    gui = dur.get(d, 0) * 1e-3 * 2.1e9
    print("%-44s %9.2f %9d %14.0f %14.0f %12.1f" % (c["name"], dur.get(d, 0), c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0),
          4 * c.get("SQ_WAVE_CYCLES", 0) / gui if gui else 0))
PY
cat gpurun_out/${TAG}_pmc_align_$W.txt
find $OUT -type f -size +1M -delete
