#!/bin/bash
# A/B of environment settings on the c4 line: tools/r05_ab_c4.sh "NAME=VAL ..." "NAME=VAL ..." ...   (3 steps + 1 warm-up each, no CPU leg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg PGX_BENCH_NO_REPLAY_TIMING=1 timeout -k 5 600 python bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_c4_$i.json 2> gpurun_out/ab_c4_$i.err
  python - "$cfg" gpurun_out/ab_c4_$i.json <<'P'
import json, sys
d = json.load(open(sys.argv[2])); k = d["kernels"]
print("%-40s %8.1f ms/step  %6.2f M/s  index %6.1f  overlap %7.1f | " % (sys.argv[1], d["ms_per_step"], d["value"] / 1e6, d["index_ms_per_step"], d["overlap_ms_per_step"]) +
      "  ".join("%s %.0f" % (n, v["ms_total"] / v["steps"]) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_total"] / kv[1]["steps"])[:6]), " hbm %.1f GB" % (d.get("hbm_bytes_in_use", 0) / 1e9), " evals", d["overlap_stats_rank0"].get("n_evaluations"), " cks", d["overlap_stats_rank0"].get("stream_checksum"))
L = d.get("hbm_ledger")
if L: print("     ledger: peak live %.1f GB, cached free at peak %.1f GB, torch reserved %.1f GB | " % (L["peak_live_bytes"] / 1e9, L["cached_free_bytes_at_peak"] / 1e9, L["torch_reserved_bytes"] / 1e9) + "  ".join("%s %.1f" % (k, v / 1e9) for k, v in sorted(L["peak_by_tag"].items(), key=lambda kv: -kv[1]) if v > 2e8))
P
done
