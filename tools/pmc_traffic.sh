#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes, kernel-trace only), per launch.
# FETCH_SIZE on gfx950 reports exactly 1/2 of a wide coalesced read stream (MI355X_MICROARCH.md, HBM): doubled below.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import csv, collections, json, re
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"$OUT/{c}/p_counter_collection.csv")):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])   # "void pgx::k_align4<16>(...)" -> k_align4
        if m:
            acc[m.group(1)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if k in ("k_align4", "k_align1", "k_sketch_wave", "k_reduce_read", "k_eval", "k_eval_rows", "k_update"):
            res.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
            res[k]["launches"] = len(v)
for k, v in res.items():
    # read side doubled (gfx950 FETCH_SIZE = 1/2 of wide coalesced reads; calibrated on k_sketch_wave: 599.8 MB reported
    # for a 1197.4 MB seqdb scan), write side as reported
    v["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
json.dump(res, open("gpurun_out/r01_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
