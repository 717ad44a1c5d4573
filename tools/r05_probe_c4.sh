#!/bin/bash
# round 5 probe: one c4 step with the library's stage trace (per-sweep lines), then a rocprofv3 kernel trace of a step -> chunk timeline,
# launch list and the full (start, end, kernel) rows of the last chunk for offline analysis
TAG=${1:-r05a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PGX_TRACE=1 PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 900 python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_trace_bench.json 2> gpurun_out/${TAG}_trace_c4.err
grep -c . gpurun_out/${TAG}_trace_c4.err
OUT=gpurun_out/c4tl; rm -rf $OUT; mkdir -p $OUT
export PGX_BENCH_NO_REPLAY_TIMING=1
timeout -k 5 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/chunk_timeline.py $f > gpurun_out/${TAG}_chunk_timeline_c4.txt 2>&1
head -45 gpurun_out/${TAG}_chunk_timeline_c4.txt
python - "$f" gpurun_out/${TAG}_chunk_rows_c4.csv <<'P'
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else re.sub(r"[^A-Za-z_:]", "", r["Kernel_Name"])[:40], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
keeps = [i for i, r in enumerate(rows) if r[2] == "k_keep"]
lo, hi = keeps[-2], keeps[-1]
t0 = rows[lo][0]
with open(sys.argv[2], "w") as o:
    for s, e, n, g, w in rows[lo:hi]:
        o.write("%d,%d,%s,%s,%s\n" % (s - t0, e - t0, n, g, w))
print("rows", hi - lo)
P
find $OUT -type f -size +1M -delete
