#!/bin/bash
# Memory-side counters of the device replay's kernels at the bench default (separate passes, kernel-trace only): request counts,
# average L1->L2 read latency, L2 hit rate, address-translation misses, atomics.   tools/pmc_replay_mem.sh [workload]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=${1:-c3}
OUT=gpurun_out/pmc_replay_mem
rm -rf $OUT; mkdir -p $OUT
run() { timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$N -o p -- python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline >> $OUT.log 2>&1 < /dev/null; }
: > $OUT.log
N=a run TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
N=b run TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
N=c run TCC_HIT_sum TCC_MISS_sum
N=d run TCC_EA0_RDREQ_sum TCC_ATOMIC_sum
N=e run TCP_TOTAL_ATOMIC_WITH_RET_sum TCC_REQ_sum
N=f run TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE
N=g run SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_BUSY_CYCLES
python - <<PY
import csv, collections, glob, re
for d in "abcdefg":
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(f"$OUT/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"\b(k_eval_rows|k_eval|k_update|k_file|k_settle|k_count|k_align_ph)\b", r["Kernel_Name"])
            if m:
                acc[m.group(1)][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, cs in sorted(acc.items()):
        for name, disp in sorted(cs.items()):
            vals = list(disp.values())
            print(f"{k:12s} {name:40s} launches {len(vals):5d}  mean {sum(vals)/len(vals):14.0f}  max {max(vals):14.0f}  total {sum(vals):16.0f}")
PY
rm -rf $OUT/?
