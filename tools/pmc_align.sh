#!/bin/bash
# SQ counters of the alignment kernels at steady state (tools/alignbench.py: 630 k alignments per launch, 20 Mb x 30x).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_align
rm -rf $OUT
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python tools/alignbench.py 20 > $OUT.log 2>&1
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o p -- python tools/alignbench.py 20 >> $OUT.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_align4" in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
    for name, disp in sorted(acc.items()):
        vals = [sum(v) for v in disp.values()]
        big = [v for v in vals if v > 0.5 * max(vals)]    # the 630 k-alignment launches (the overlap run before them has small ones)
        print(f"{name:24s} per big launch: {sum(big)/len(big):16.0f}   ({len(big)} launches)")
PY
tail -3 $OUT.log
