#!/usr/bin/env python3
"""GPU busy / idle timeline of ONE bench step from a rocprofv3 --kernel-trace csv (memory copies are not in it: a gap
may be a copy).  A step = the interval between the last two k_sketch_blk launches that are followed by alignment work.
usage: tools/timeline.py p_kernel_trace.csv [min_gap_us [sequence_until_ms [sequence_from_ms]]]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", n)
    m = re.search(r"rocprim::detail::(?:trampoline_kernel<rocprim::detail::wrapped_)?(\w+)", n)
    if "rocprim" in n and m:
        return "rocprim::" + m.group(1)
    return n.replace("void ", "").replace("pgx::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sketch_blk")]
    if len(starts) < 2:
        print("fewer than two steps in the trace"); return
    a, b = starts[-2], starts[-1]
    step = rows[a:b]
    t0, t1 = step[0][0], rows[b][0]
    busy = 0
    cur_s, cur_e = step[0][0], step[0][1]
    gaps = []
    prev = step[0][2]
    for s, e, n in step[1:] + [(t1, t1, "next step")]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    span = t1 - t0
    print(f"step span {span/1e6:.2f} ms, {len(step)} kernels, busy {busy/1e6:.2f} ms, idle {(span-busy)/1e6:.2f} ms")
    small = sum(g for g, *_ in gaps if g < min_gap * 1e3)
    print(f"gaps below {min_gap:.0f} us: {sum(1 for g,*_ in gaps if g < min_gap*1e3)} totalling {small/1e6:.2f} ms")
    print("gaps above: us, at ms, after kernel -> before kernel")
    for g, at, p, n in gaps:
        if g >= min_gap * 1e3:
            print(f"  {g/1e3:9.1f}  {at/1e6:8.2f}  {p} -> {n}")
    if len(sys.argv) > 3:   # the sequence up to this many ms: runs of the same kernel merged
        lim = float(sys.argv[3]) * 1e6
        lo = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 0.0
        print("sequence: at ms, busy us, launches, kernel")
        run = None
        for s, e, n in step:
            if s - t0 > lim:
                break
            if s - t0 < lo:
                continue
            if run and run[3] == n:
                run[1] += e - s; run[2] += 1
            else:
                if run: print(f"  {run[0]/1e6:8.3f} {run[1]/1e3:9.1f} {run[2]:4d}  {run[3]}")
                run = [s - t0, e - s, 1, n]
        if run: print(f"  {run[0]/1e6:8.3f} {run[1]/1e3:9.1f} {run[2]:4d}  {run[3]}")
    agg = {}
    for s, e, n in step:
        x = agg.setdefault(n, [0, 0]); x[0] += 1; x[1] += e - s
    print("kernels of the step:")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {t/1e6:8.3f} ms {c:5d}  {n}")


if __name__ == "__main__":
    main()
