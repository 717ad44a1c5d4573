#!/bin/bash
# rocprofv3 kernel trace of a c4 step -> chunk timeline (profiles/<tag>_chunk_timeline_c4.txt) and kernel stats
TAG=${1:-r04y}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/c4tl; rm -rf $OUT; mkdir -p $OUT
export PGX_BENCH_NO_REPLAY_TIMING=1
timeout -k 5 1200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python bench.py --workload c4 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/chunk_timeline.py $f > gpurun_out/${TAG}_chunk_timeline_c4.txt 2>&1
head -50 gpurun_out/${TAG}_chunk_timeline_c4.txt
python - "$f" <<'P' | tee gpurun_out/${TAG}_launch_list_c4.txt
# the launches of the kernels that run once per sweep, in order, for the LAST chunk of the trace: where a tail sweep's time goes
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
keeps = [i for i, r in enumerate(rows) if r[2] == "k_keep"]
lo = keeps[-2] if len(keeps) > 1 else 0
hi = keeps[-1] if len(keeps) > 1 else len(rows)
t0 = rows[lo][0]
names = ("k_file", "k_settle", "k_align_ph", "k_align1", "k_align1_list", "k_emit")
npass = 0
for s, e, n in rows[lo:hi]:
    if n == "k_update": npass += 1
    if n in names:
        print("%9.2f ms  %-14s %9.1f us   (%d passes before)" % ((s - t0) / 1e6, n, (e - s) / 1e3, npass)); npass = 0
P
find $OUT -type f -size +1M -delete
