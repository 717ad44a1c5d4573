#!/usr/bin/env python3
"""Where does a SERVED overlap command at full-size configs[3] spend its time?  The read set to /dev/shm, `pgx_cli serve` with PGX_TRACE=1,
the 8 index commands, then N overlap commands; prints the server's stage lines.   usage: tools/r05_e2e_trace.py [n_overlap=2]"""
import os, signal, subprocess, sys, tempfile, time, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peregrine_amd import simreads
n_ov = int(sys.argv[1]) if len(sys.argv) > 1 else 2
seq, total, rlen = simreads.make_workload_resident("c4")
rid = np.arange(len(rlen), dtype=np.uint32)
roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
d = tempfile.mkdtemp(prefix="pgx_e2e_", dir="/dev/shm")
try:
    pre = os.path.join(d, "sd")
    simreads.write_seqdb_from_device(pre, seq, total, rid, rlen, roff)
    del seq
    torch.cuda.empty_cache()
    cli = os.path.join(ROOT, "bin", "native", "pgx_cli")
    log = open(os.path.join(ROOT, "gpurun_out", "e2e_server_trace.log"), "w")
    srv = subprocess.Popen([cli, "serve", "-p", pre], stderr=log, env=dict(os.environ, PGX_TRACE="1"))
    while not os.path.exists(pre + ".pgx.sock") and srv.poll() is None:
        time.sleep(0.05)
    try:
        for c in range(1, 9):
            subprocess.run([cli, "shmr_index", "-p", pre, "-t", "8", "-c", str(c), "-m", "0", "-l", "2", "-o", os.path.join(d, "ix")], check=True)
        for c in range(1, n_ov + 1):
            t0 = time.perf_counter()
            subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", os.path.join(d, "ix-L2"), "-t", "8", "-c", str(c), "-M", "240", "-o", os.path.join(d, "ov.%02d" % c)], check=True)
            print("overlap command %d: %.2f s" % (c, time.perf_counter() - t0), flush=True)
    finally:
        srv.send_signal(signal.SIGTERM); srv.wait()
    log.close()
    for line in open(os.path.join(ROOT, "gpurun_out", "e2e_server_trace.log")):
        if any(k in line for k in ("stage total", "overlap chunk", "GPU join", "device replay", "gave up", "host replay", "note", "sweep 1:", "tables:", "visit on")):
            print(line.rstrip()[:260])
finally:
    shutil.rmtree(d, ignore_errors=True)
