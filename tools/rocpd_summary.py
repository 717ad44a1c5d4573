#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd sqlite database as a --stats style table.
usage: tools/rocpd_summary.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"rocprim::detail::(?:trampoline_kernel<rocprim::detail::wrapped_)?(\w+)", name)
    if name.startswith("void rocprim") and m:
        return "rocprim::" + m.group(1)
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-44s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    agg = {}
    for name, calls, tot, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls; a[1] += tot; a[2] += pct
    for k, (calls, tot, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-44s %8d %14.1f %12.2f %6.2f%%" % (k[:44], calls, tot, tot / calls, pct))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
