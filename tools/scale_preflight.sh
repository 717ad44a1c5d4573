#!/bin/bash
# Pre-flight of a multi-GPU node BEFORE the full-size scaling runs (VERDICT r5 task 7a): the c4 recipe on a 300 Mb genome, the job's 8 index + 8
# overlap chunks dealt to N = 2, 4, 8 ranks over RCCL, every rank's ovlp_t stream of each of its chunks compared field by field with
# oracle/_ref/shmr_overlap -t 8 -c c (bench.py --check-ref).  N = 2 / 4 take the all-gather form, N = 8 the all-to-all(v) of pair records.
# Exit code 0 only if every stream of every N equals the reference's; the first mismatch or failed launch stops the script.
#   tools/scale_preflight.sh [N ...]        (default: 2 4 8, those that the node's GPU count allows)
cd "$(dirname "$0")/.." || exit 2
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
[ $# -gt 0 ] && NS="$*" || NS="2 4 8"
mkdir -p gpurun_out
rc=0
for N in $NS; do
  if [ "$N" -gt "$NG" ]; then echo "[preflight] N=$N skipped: the node has $NG GPU(s)"; continue; fi
  out=gpurun_out/preflight_n$N.json
  port=$((29600 + N))
  timeout -k 10 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus "$N" --workload c4 --genome-mb 300 --check-ref --steps 1 --warmup 1 --no-cpu-baseline > "$out" 2> "gpurun_out/preflight_n$N.err"
  st=$?
  python - "$out" "$N" "$st" <<'P'
import json, sys
path, n, st = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
except Exception as e:
    print(f"[preflight] N={n}: no JSON line (launcher exit code {st}): {e}"); sys.exit(1)
chk = d.get("check_vs_reference") or {}
rows = chk.get("chunks", [])
ok = bool(chk.get("all_equal")) and d.get("world_size") == n and d.get("read_set_hash_equal_on_all_ranks") is True
print(f"[preflight] N={n}: world_size {d.get('world_size')}, {d['config']['parallelism']}, {d['ms_per_step']:.0f} ms/step, "
      f"{len(rows)} chunk streams compared with the reference, all equal: {chk.get('all_equal')}; read set equal on all ranks: {d.get('read_set_hash_equal_on_all_ranks')}")
for r in rows:
    if not r["equal_to_reference"]:
        print(f"[preflight]   MISMATCH chunk {r['chunk']} ({r['records']} reference records)")
sys.exit(0 if ok else 1)
P
  if [ $? -ne 0 ]; then echo "[preflight] FAILED at N=$N (see $out, gpurun_out/preflight_n$N.err)"; rc=1; break; fi
done
[ $rc -eq 0 ] && echo "[preflight] every compared stream equals the reference's: the full-size command may run"
exit $rc
