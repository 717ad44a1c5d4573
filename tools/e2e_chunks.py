#!/usr/bin/env python3
"""The drop-in path at scale (VERDICT r3 task 9): a job's N index + N overlap chunk commands (what pg_run.py launches,
/root/reference/py/scripts/pg_run.py:232-244,305-317) through bin/native/shmr_index / shmr_overlap on files in /dev/shm,
  (a) stand-alone: every command its own process = process start + HIP context + seqdb file -> HBM, N + N times;
  (b) served: `pgx_cli serve -p <prefix>` holds the database, the same commands attach to it;
  (c) resident: the same chunks through the library in ONE process (ResidentDB), files written by the caller -- the floor.
Outputs of (a) and (b) are compared byte for byte.   usage: tools/e2e_chunks.py [workload=c4s] [chunks=8] [--out profiles/x.json]"""
import json, os, shutil, signal, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if args else "c4s"
N = int(args[1]) if len(args) > 1 else 8
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
sp = dict(levels=2, mc_upper=240); sp.update(simreads.STAGE_PARAMS.get(wl, {}))
lv = sp["levels"]
db = simreads.make_workload_torch(wl) if wl in simreads.TORCH_WORKLOADS else simreads.make_workload(wl)
d = tempfile.mkdtemp(prefix="pgx_e2e_", dir="/dev/shm")
cli = os.path.join(ROOT, "bin", "native", "pgx_cli")
res = {"workload": wl, "chunks": N, "reads": db.n_reads, "bases": db.n_bases}
try:
    pre = os.path.join(d, "sd")
    formats.write_seqdb(pre, db)

    def commands(tag):
        os.makedirs(os.path.join(d, tag), exist_ok=True)
        t0 = time.perf_counter()
        for c in range(1, N + 1):
            subprocess.run([cli, "shmr_index", "-p", pre, "-t", str(N), "-c", str(c), "-m", "0", "-l", str(lv), "-o", os.path.join(d, tag, "ix")], check=True, capture_output=True)
        t1 = time.perf_counter()
        for c in range(1, N + 1):
            subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", os.path.join(d, tag, "ix-L%d" % lv), "-t", str(N), "-c", str(c), "-M", str(sp["mc_upper"]),
                            "-o", os.path.join(d, tag, "ov.%02d" % c)], check=True, capture_output=True)
        t2 = time.perf_counter()
        nrec = sum(os.path.getsize(os.path.join(d, tag, "ov.%02d" % c)) // 64 for c in range(1, N + 1))
        return {"index_s": t1 - t0, "overlap_s": t2 - t1, "records": nrec, "overlaps_per_s": nrec / (t2 - t0)}

    res["stand_alone"] = commands("alone")
    print("stand-alone:", res["stand_alone"], flush=True)
    t0 = time.perf_counter()
    srv_log = open(os.path.join(ROOT, "gpurun_out", "e2e_server_trace.log"), "w") if os.environ.get("E2E_SERVER_TRACE") else subprocess.DEVNULL
    srv = subprocess.Popen([cli, "serve", "-p", pre], stderr=srv_log, env=dict(os.environ, PGX_TRACE="1") if os.environ.get("E2E_SERVER_TRACE") else None)
    while not os.path.exists(pre + ".pgx.sock") and srv.poll() is None:
        time.sleep(0.05)
    t_up = time.perf_counter() - t0
    try:
        res["served"] = commands("served")
    finally:
        srv.send_signal(signal.SIGTERM); srv.wait()
    res["served"]["server_start_s"] = t_up
    res["served"]["overlaps_per_s_incl_server_start"] = res["served"]["records"] / (res["served"]["index_s"] + res["served"]["overlap_s"] + t_up)
    print("served:", res["served"], flush=True)
    same = all(open(os.path.join(d, "alone", f), "rb").read() == open(os.path.join(d, "served", f), "rb").read() for f in sorted(os.listdir(os.path.join(d, "alone"))))
    res["files_identical"] = bool(same and sorted(os.listdir(os.path.join(d, "alone"))) == sorted(os.listdir(os.path.join(d, "served"))))
    # (c) one process, library calls
    rdb = ResidentDB(db, 0)
    for rep in range(2):
        t0 = time.perf_counter()
        parts = [rdb.index(total_chunk=N, mychunk=c, levels=lv) for c in range(1, N + 1)]
        mm = np.concatenate([p.top for p in parts]); mc = np.concatenate([p.top_mc for p in parts])
        t1 = time.perf_counter()
        nrec = 0
        for c in range(1, N + 1):
            ov, _ = rdb.overlap(mm, mc, total_chunk=N, mychunk=c, mc_upper=sp["mc_upper"])
            nrec += len(ov)
        t2 = time.perf_counter()
    res["resident"] = {"index_s": t1 - t0, "overlap_s": t2 - t1, "records": nrec, "overlaps_per_s": nrec / (t2 - t0)}
    res["served_over_resident"] = (res["served"]["index_s"] + res["served"]["overlap_s"]) / (t2 - t0)
    res["stand_alone_over_resident"] = (res["stand_alone"]["index_s"] + res["stand_alone"]["overlap_s"]) / (t2 - t0)
    print(json.dumps(res))
    if out:
        json.dump(res, open(out, "w"), indent=1)
finally:
    shutil.rmtree(d, ignore_errors=True)
