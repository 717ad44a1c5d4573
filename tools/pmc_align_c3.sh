#!/bin/bash
# SQ counters of the large-launch alignment kernel on the bench's c3 step, per alignment.  usage: tools/pmc_align_c3.sh <mode> (PGX_ALIGN_MODE)
M=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_align_m$M
rm -rf $OUT
PGX_ALIGN_MODE=$M timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1
PGX_ALIGN_MODE=$M timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/b -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline >> $OUT.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align4" in r["Kernel_Name"] or "k_align_ph" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, disp in sorted(acc.items()):
        vals = list(disp.values())
        big = [v for v in vals if v > 0.5 * max(vals)]
        print(f"mode $M {name:24s} per big launch (the step's first: 4.66 M alignments at c3): {sum(big)/len(big):16.0f}   ({len(big)} launches)")
PY
