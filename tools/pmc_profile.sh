#!/bin/bash
# The counters bench.py attaches to its roofline objects, for ONE workload, from separate rocprofv3 passes of the bench command
# (kernel-trace + one --pmc group per pass; never combined with the other trace domains):
#   pass 1, 2: FETCH_SIZE, WRITE_SIZE            -> profiles/<tag>_traffic_<workload>.json   (HBM bytes per launch)
#   pass 3   : SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES -> profiles/<tag>_valu_<workload>.json (wave64 VALU instructions per unit,
#              issue rate per SIMD and cycle: the roofline that binds the sketch and alignment kernels)
#   pass 0   : --kernel-trace --stats            -> profiles/<tag>_kernel_stats_bench_<workload>.txt
# usage: tools/pmc_profile.sh [workload=c3] [tag=r05] [steps=2] [extra bench args...]
W=${1:-c3}; TAG=${2:-r05}; S=${3:-2}; shift 3 2>/dev/null
EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$W
rm -rf $OUT; mkdir -p $OUT profiles
# the simulated set of the one-chunk workloads is made once, outside the profiler (rocprofv3 --pmc aborts inside torch's generator kernels of
# the repeat-planting genome builder); the extra replay-timing step is switched off so that the profiled launches are exactly the timed ones
export PGX_BENCH_CACHE=/dev/shm/pgx_bench_cache PGX_BENCH_NO_REPLAY_TIMING=1 PGX_BENCH_NO_STREAM_HASH=1
CMD="python bench.py --workload $W --steps $S --warmup 0 --no-cpu-baseline $EXTRA"
timeout 900 $CMD > $OUT/plain.json 2> $OUT/plain.err
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 900 rocprofv3 --kernel-trace --kernel-include-regex pgx --pmc $c --output-format csv -d $OUT/$c -o p -- $CMD > $OUT/$c.json 2> $OUT/$c.err
done
timeout -k 5 900 rocprofv3 --kernel-trace --kernel-include-regex pgx --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/SQ -o p -- $CMD > $OUT/SQ.json 2> $OUT/SQ.err
python - <<PY
import csv, collections, glob, json, re, os
W, TAG, S, OUT = "$W", "$TAG", int("$S"), "$OUT"
def kname(n):
    m = re.search(r"\b(k_[a-z0-9_]+)", n)
    if not m: return None
    k = m.group(1)
    # round 6: the sketch kernels have a byte form and a form on the 2-bit packs (template argument PACKED); with --warmup 0 the first step's index
    # stage still runs on the bytes (the packs are built by its first overlap stage): the counters of the two forms are kept apart
    if k in ("k_sketch_blk", "k_sketch_wave") and not re.search(r"true>?\s*[\(>]|, true>", n.split("(")[0]): k += "_bytes"
    return k
def rows(d, pat):
    for f in glob.glob(f"{OUT}/{d}/**/*{pat}.csv", recursive=True):
        yield from csv.DictReader(open(f))
# ---- kernel stats
st = collections.defaultdict(lambda: [0, 0.0])
for r in rows("stats", "kernel_trace"):
    k = kname(r["Kernel_Name"]) or r["Kernel_Name"].split("(")[0][:60]
    st[k][0] += 1; st[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in st.values()) or 1
with open(f"profiles/{TAG}_kernel_stats_bench_{W}.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {W} --steps {S} --warmup 0 --no-cpu-baseline $EXTRA (generation kernels of the read simulator included)\n")
    f.write("%-48s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for k, (c, t) in sorted(st.items(), key=lambda kv: -kv[1][1])[:60]:
        f.write("%-48s %8d %14.1f %12.2f %6.2f%%\n" % (k[:48], c, t, t / c, 100 * t / tot))
# ---- traffic
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in rows(c, "counter_collection"):
        k = kname(r["Kernel_Name"])
        if k: acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res.setdefault(k, {})[c + "_KB_per_launch"] = sum(v) / len(v)
        res[k]["launches"] = len(v)
for k, v in res.items():
    # FETCH_SIZE counts 64 B per fabric request; a request moves 128 B for wide coalesced streams (x2: sketch / pack / join kernels) and 64 B for
    # scattered 8- / 16-byte loads (x1: the alignment kernels' probes) -- profiles/r03_fetch_calib.txt.  Write side as reported.
    rf = 1 if k.startswith("k_align") else 2
    v["read_factor"] = rf
    v["hbm_bytes_per_launch"] = (rf * v.get("FETCH_SIZE_KB_per_launch", 0) + v.get("WRITE_SIZE_KB_per_launch", 0)) * 1024
for k in ("k_sketch_blk", "k_sketch_wave"):
    if k in res and k + "_bytes" in res: res[k]["steps"] = S - 1     # (the first step's index stage ran on the bytes)
res["_workload"] = W
res["_steps"] = S
res["_command"] = f"rocprofv3 --kernel-trace --pmc {{FETCH_SIZE|WRITE_SIZE}} -- python bench.py --workload {W} --steps {S} --warmup 0 --no-cpu-baseline $EXTRA"
json.dump(res, open(f"profiles/{TAG}_traffic_{W}.json", "w"), indent=1)
# ---- VALU issue
try:
    line = json.loads(open(f"{OUT}/SQ.json").read().strip().splitlines()[-1])
except Exception:
    line = {}
dur = {}
for r in rows("SQ", "kernel_trace"):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows("SQ", "counter_collection"):
    k = kname(r["Kernel_Name"])
    if k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
units = {"k_sketch_blk": (line.get("kernels", {}).get("sketch", {}).get("units"), "base"),
         "k_align_ph": (line.get("kernels", {}).get("align", {}).get("units"), "alignment")}
vv = {}
for k, c in acc.items():
    secs = sum(dur.get(d, 0) for d in disp[k])
    e = {"launches": len(disp[k]), "seconds": secs, **{n: v for n, v in c.items()}}
    # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): per engine = the kernel's busy cycles
    e["SQ_BUSY_CYCLES_per_se"] = c.get("SQ_BUSY_CYCLES", 0) / 32
    e["clock_hz"] = e["SQ_BUSY_CYCLES_per_se"] / secs if secs else None
    if k in units and units[k][0]:
        e["units"], e["unit_name"] = units[k][0], units[k][1]
        if k == "k_sketch_blk" and "k_sketch_blk_bytes" in acc:   # bench.py counted the bases of every step; these launches are the packed ones
            e["units"] = units[k][0] * len(disp[k]) / (len(disp[k]) + len(disp["k_sketch_blk_bytes"]))
    vv[k] = e
vv["_workload"] = W
vv["_steps"] = S
vv["_command"] = f"rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES -- python bench.py --workload {W} --steps {S} --warmup 0 --no-cpu-baseline $EXTRA"
vv["_note"] = "units = what bench.py's timers counted over the same launches (warmup 0, no extra replay-timing step: PGX_BENCH_NO_REPLAY_TIMING=1)"
json.dump(vv, open(f"profiles/{TAG}_valu_{W}.json", "w"), indent=1)
print(json.dumps({k: {n: v for n, v in e.items() if n in ("launches", "seconds", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_BUSY_CYCLES_per_se", "units")} for k, e in vv.items() if k in ("k_align_ph", "k_sketch_blk", "k_eval_big", "k_eval", "k_eval_rows")}, indent=1))
PY
cp $OUT/plain.json profiles/${TAG}_bench_${W}_nocpu.json 2>/dev/null
find $OUT -type f -size +2M -delete
