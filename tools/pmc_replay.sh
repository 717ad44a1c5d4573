#!/bin/bash
# SQ counters of the device replay's kernels on the bench set (separate passes, kernel-trace only): what a step of k_eval costs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_replay
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT.log 2>&1 < /dev/null
timeout -k 5 180 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $OUT/b -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline >> $OUT.log 2>&1 < /dev/null
python - <<PY
import csv, collections, glob, re
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"\b(k_eval_rows|k_eval|k_update|k_file|k_settle|k_count)\b", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, cs in sorted(acc.items()):
        for name, disp in sorted(cs.items()):
            vals = list(disp.values())
            print(f"{k:12s} {name:22s} launches {len(vals):5d}  mean {sum(vals)/len(vals):14.0f}  max {max(vals):14.0f}  total {sum(vals):16.0f}")
PY
tail -2 $OUT.log | cut -c1-200
