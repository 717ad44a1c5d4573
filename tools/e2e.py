#!/usr/bin/env python3
"""End-to-end timing of the native drop-in executables (process start, HIP context, file -> HBM, kernels, D2H, file write) on a
c3-size file set in /dev/shm: usage tools/e2e.py [genome_Mb=150] [threads...]"""
import os, sys, time, subprocess, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peregrine_amd import formats, simreads
gmb = float(sys.argv[1]) if len(sys.argv) > 1 else 150
db = simreads.simulate_reads_torch(int(gmb * 1e6), 1003, 30.0)
d = tempfile.mkdtemp(prefix="pgx_e2e_", dir="/dev/shm")
try:
    pre = os.path.join(d, "sd")
    formats.write_seqdb(pre, db)
    exe = os.path.join(ROOT, "bin", "native")
    if os.environ.get("E2E_TRACE"):
        env = dict(os.environ, PGX_TRACE="1")
        for rep in range(2):
            t0 = time.perf_counter()
            subprocess.run([os.path.join(exe, "shmr_index"), "-p", pre, "-t", "1", "-c", "1", "-m", "0", "-o", os.path.join(d, "gx")], check=True, env=env)
            print(f"== shmr_index wall {time.perf_counter()-t0:.3f} s", flush=True)
            t0 = time.perf_counter()
            subprocess.run([os.path.join(exe, "shmr_overlap"), "-p", pre, "-l", os.path.join(d, "gx-L2"), "-t", "1", "-c", "1", "-o", os.path.join(d, "gov")], check=True, env=env)
            print(f"== shmr_overlap wall {time.perf_counter()-t0:.3f} s", flush=True)
    if os.environ.get("E2E_PARENT_STEP"):   # the state bench.py's parent process is in when it times the executables
        from peregrine_amd.shimmer import ResidentDB
        rdb = ResidentDB(db, 0)
        for _ in range(2):
            ix, ov, st = rdb.index_overlap()
        print("parent ran", len(ov), "records; affinity", len(os.sched_getaffinity(0)), "cpus", flush=True)
        if os.environ["E2E_PARENT_STEP"] == "close":
            rdb.close()
    for thr in (sys.argv[2:] or ["1", "4", "8", "16"]):
        env = dict(os.environ, PGX_LOAD_THREADS=thr)
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            subprocess.run([os.path.join(exe, "shmr_index"), "-p", pre, "-t", "1", "-c", "1", "-m", "0", "-o", os.path.join(d, "gx")], check=True, env=env)
            t1 = time.perf_counter()
            subprocess.run([os.path.join(exe, "shmr_overlap"), "-p", pre, "-l", os.path.join(d, "gx-L2"), "-t", "1", "-c", "1", "-o", os.path.join(d, "gov")], check=True, env=env)
            t2 = time.perf_counter()
            if best is None or t2 - t0 < sum(best): best = (t1 - t0, t2 - t1)
        print(f"load threads {thr}: shmr_index {best[0]:.3f} s ({db.n_bases/best[0]/1e9:.1f} Gbases/s), shmr_overlap {best[1]:.3f} s, {os.path.getsize(os.path.join(d, 'gov'))//64} records", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
