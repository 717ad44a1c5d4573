#!/bin/bash
# SQ counters of the sketch kernel of the index stage (default: k_sketch_blk; PGX_SKETCH=wave for round 1's k_sketch_wave):
#   tools/pmc_sketch.sh [tag]      4 x the E. coli-size set = 299 Mbases, 19,936 waves per launch, one wave per read
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-blk}
OUT=gpurun_out/pmc_sketch_$T
rm -rf $OUT
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/a -o p -- python tools/kbench.py 4 sketch > $OUT.log 2>&1
timeout -k 5 420 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o p -- python tools/kbench.py 4 sketch >> $OUT.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for nm in ("k_sketch_blk", "k_sketch_wave", "k_reduce_read"):
                if nm in k:
                    acc[nm][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for nm, cs in acc.items():
        for name, disp in sorted(cs.items()):
            vals = list(disp.values())
            big = [v for v in vals if v > 0.5 * max(vals)] if max(vals) > 0 else vals   # (the redo launch of the general kernel is small)
            print(f"{nm:14s} {name:24s} per launch: {sum(big)/len(big):16.0f}   ({len(big)} of {len(vals)} launches)")
PY
grep sketch $OUT.log | tail -4
