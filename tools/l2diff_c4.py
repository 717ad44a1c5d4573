#!/usr/bin/env python3
"""Index-list parity at FULL size: every final-level shimmer list of BASELINE configs[3] (93 Gbases, 6.2 M reads) from the resident GPU index
against the reference binary's files (oracle/_ref/shmr_index as T processes over T chunks), chunk by chunk, element by element.
(tools/l2diff.py found the one-read first-window bug of round 3 on the 9-Gbase set; this is the same hunt over ten times the reads.)
  python tools/l2diff_c4.py [T=24] [levels=2] [genome_mb=3100]"""
import os, sys, tempfile, shutil, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import concurrent.futures as cf
import torch
import oracle_util as U
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB

T = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lv = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gmb = float(sys.argv[3]) if len(sys.argv) > 3 else None
seq, total, rlen = simreads.make_workload_resident("c4", genome_mb=gmb)
rid = np.arange(len(rlen), dtype=np.uint32)
roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
rdb = ResidentDB.adopt_device(seq, total, rid, rlen, roff, 0)
d = tempfile.mkdtemp(prefix="l2diff_", dir="/dev/shm")
try:
    pre = os.path.join(d, "sd")
    t = time.time()
    simreads.write_seqdb_from_device(pre, seq, total, rid, rlen, roff)
    print("files written in %.1f s" % (time.time() - t), flush=True)
    t = time.time()
    with cf.ThreadPoolExecutor(min(T, 24)) as ex:
        list(ex.map(lambda c: U.ref_run("shmr_index", "-p", pre, "-t", T, "-c", c, "-m", 0, "-l", lv, "-o", os.path.join(d, "ref")), range(1, T + 1)))
    print("reference index (%d processes): %.1f s" % (min(T, 24), time.time() - t), flush=True)
    bad = 0
    nel = 0
    nsecond = 0
    for c in range(1, T + 1):
        ref = formats.read_mmlist(os.path.join(d, "ref-L%d-%02d-of-%02d.dat" % (lv, c, T)))
        ix = rdb.index(total_chunk=T, mychunk=c, levels=lv)
        got = ix.top
        nsecond += ix.reads_literal
        nel += len(ref)
        ok = np.array_equal(ref, got)
        mc_ok = np.array_equal(formats.mc_as_sorted_pairs(formats.read_mm_count(os.path.join(d, "ref-L%d-MC-%02d-of-%02d.dat" % (lv, c, T)))), formats.mc_as_sorted_pairs(ix.top_mc))
        if not (ok and mc_ok):
            bad += 1
            n = min(len(ref), len(got))
            neq = np.flatnonzero((ref["x"][:n] != got["x"][:n]) | (ref["y"][:n] != got["y"][:n]))
            i = int(neq[0]) if len(neq) else n
            r = int(ref["y"][min(i, len(ref) - 1)] >> np.uint64(32))
            print("chunk %d DIFFERS (lists equal %s, counts equal %s): first at element %d, read %d (len %d)" % (c, ok, mc_ok, i, r, int(rlen[r])), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.save(os.path.join(ROOT, "gpurun_out", "l2diff_c4_read_%d.npy" % r), seq[int(roff[r]):int(roff[r]) + int(rlen[r])].cpu().numpy())
        else:
            print("chunk %d: %d elements, identical (count table too)" % (c, len(ref)), flush=True)
    print("CHUNKS %d, ELEMENTS %d, reads taken run by run %d, DIFFERENCES %d" % (T, nel, nsecond, bad))
finally:
    shutil.rmtree(d, ignore_errors=True)
