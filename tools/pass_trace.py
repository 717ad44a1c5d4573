#!/usr/bin/env python3
"""The evaluate / update passes of ONE step from a rocprofv3 --kernel-trace csv (bench.py --workload c4s --steps 1 --warmup 0):
per pass (a k_update launch closes it) the durations of k_eval / k_eval_rows / k_eval_big / k_update / counts and the gap to the next
pass; then histograms.  usage: tools/pass_trace.py p_kernel_trace.csv"""
import collections, csv, re, sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
# the last step: from the last k_keep on
keeps = [i for i, r in enumerate(rows) if r[2] == "k_keep"]
rows = rows[keeps[-1]:]
t0 = rows[0][0]
passes = []
cur = collections.defaultdict(float); start = None
for s, e, n in rows:
    if n in ("k_eval", "k_eval_rows", "k_eval_big", "k_update", "k_count_a", "k_count_b"):
        if start is None: start = s
        cur[n] += (e - s) / 1e3
        cur["end"] = max(cur.get("end", 0), e)
        if n == "k_count_b" or (n == "k_update" and False):
            passes.append((start, dict(cur))); cur = collections.defaultdict(float); start = None
    elif n in ("k_file", "k_settle", "k_align_ph", "k_align1", "k_align1_list", "k_emit"):
        passes.append((s, {"marker": n, "dur": (e - s) / 1e3}))
print("t_ms    eval   rows    big  update counts   wall")
tot = collections.Counter()
for i, (s, p) in enumerate(passes):
    if "marker" in p:
        print(f"{(s - t0) / 1e6:8.2f}  -- {p['marker']} {p['dur']:.0f} us")
        continue
    wall = (p["end"] - s) / 1e3
    for k in ("k_eval", "k_eval_rows", "k_eval_big", "k_update"): tot[k] += p.get(k, 0)
    tot["wall"] += wall; tot["n"] += 1
    print(f"{(s - t0) / 1e6:8.2f} {p.get('k_eval', 0):6.0f} {p.get('k_eval_rows', 0):6.0f} {p.get('k_eval_big', 0):6.0f} {p.get('k_update', 0):6.0f} {p.get('k_count_a', 0) + p.get('k_count_b', 0):6.0f} {wall:7.0f}")
print("totals (us):", dict(tot))
