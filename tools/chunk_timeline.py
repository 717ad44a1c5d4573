#!/usr/bin/env python3
"""One overlap chunk of a multi-chunk step (bench.py --workload c4) from a rocprofv3 --kernel-trace csv: where its wall time goes.
The chunk = from a `k_keep` launch (the join's first kernel) to the next one (or the end of the trace); the LAST complete chunk is
analysed.  Prints: wall, union busy time, idle split by the kernel that preceded the gap, per-kernel totals / launches, and the
time during which ONLY k_eval_big ran (the side stream's chain holding the pass).
usage: tools/chunk_timeline.py p_kernel_trace.csv [chunk_index_from_end=2]"""
import collections, csv, re, sys


def short(n):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", n)
    m = re.search(r"rocprim::detail::(?:trampoline_kernel<rocprim::detail::wrapped_)?(\w+)", n)
    if "rocprim" in n and m:
        return "rocprim::" + m.group(1)
    return n.replace("void ", "").replace("pgx::", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
keeps = [i for i, r in enumerate(rows) if r[2] == "k_keep"]
a, b = keeps[-back], (keeps[-back + 1] if back > 1 else len(rows))
ch = rows[a:b]
t0, t1 = ch[0][0], max(r[1] for r in ch)
print(f"chunk: {len(ch)} launches, wall {(t1 - t0) / 1e6:.1f} ms")
ev = []
for s, e, n in ch:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active = collections.Counter()
last = t0
busy = 0
only = collections.Counter()
gap_by = collections.Counter()
prev_end_name = ch[0][2]
for t, d, n in ev:
    if t > last:
        tot = sum(active.values())
        if tot:
            busy += t - last
            names = [k for k, v in active.items() if v > 0]
            if len(names) == 1:
                only[names[0]] += t - last
        else:
            gap_by[prev_end_name] += t - last
        last = t
    active[n] += d
    if d < 0:
        prev_end_name = n
print(f"busy (union of kernel intervals) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms")
print("idle time by the kernel that ended before the gap:")
for k, v in gap_by.most_common(12):
    print(f"   {k:40s} {v / 1e6:8.2f} ms")
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in ch:
    tot[n] += e - s; cnt[n] += 1
print("per kernel: total ms, launches, avg us, ms while it was the only kernel running")
for k, v in tot.most_common(28):
    print(f"   {k:40s} {v / 1e6:9.2f} {cnt[k]:7d} {v / cnt[k] / 1e3:9.1f} {only[k] / 1e6:9.2f}")
