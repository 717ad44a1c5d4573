#!/bin/bash
# FETCH_SIZE calibration (VERDICT r2 #5 / housekeeping #9): runs tools/build/fetch_calib under rocprofv3 --pmc FETCH_SIZE and
# prints, per access pattern, the counter against the distinct 32- / 64- / 128-byte granules the loads touch.
#   usage (GPU box): tools/fetch_calib.sh > gpurun_out/r03_fetch_calib.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT
timeout -k 5 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p -o p -- tools/build/fetch_calib > $OUT/cal.txt 2> $OUT/err.txt
python3 - <<PY
import csv, glob, re
cal = {}
for line in open("$OUT/cal.txt"):
    if line.startswith("CAL"):
        m = re.match(r"CAL (\S+(?: \d+>)?)\s+requested_bytes (\d+) distinct32 (\d+) distinct64 (\d+) distinct128 (\d+)", line)
        cal[m.group(1)] = [int(m.group(i)) for i in (2, 3, 4, 5)]
pmc = {}
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            pmc[k] = pmc.get(k, 0.0) + float(r["Counter_Value"]) * 1024
print("# FETCH_SIZE (raw, KB x 1024) of one launch per pattern vs the bytes of the distinct granules its loads touch (2 GiB buffer)")
print("%-22s %14s %14s | raw/requested raw/distinct32 raw/distinct64 raw/distinct128" % ("kernel", "requested B", "FETCH_SIZE B"))
for k, (req, d32, d64, d128) in cal.items():
    v = next((pmc[n] for n in pmc if n.replace(" ", "") == k.replace(" ", "")), None)
    if v is None:
        print("%-22s %14d %14s" % (k, req, "no counter row")); continue
    print("%-22s %14d %14.0f | %13.3f %14.3f %14.3f %15.3f" % (k, req, v, v / req, v / d32, v / d64, v / d128))
PY
