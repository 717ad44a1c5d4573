#!/usr/bin/env python3
"""Index-file parity hunt at scale: the final-level shimmer FILE of a workload from (a) the reference binary (oracle/_ref/shmr_index,
N processes over N chunks), (b) bin/native/shmr_index (the drop-in, one process per chunk) and (c) the resident API (ResidentDB.index),
compared byte for byte; on a difference the first differing read is dumped (gpurun_out/l2diff_read.npy) with the oracle's list for it.
  python tools/l2diff.py [workload=c4s] [chunks=8] [levels=2]"""
import os, subprocess, sys, tempfile, shutil, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as U
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB
import concurrent.futures as cf

wl = sys.argv[1] if len(sys.argv) > 1 else "c4s"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lv = int(sys.argv[3]) if len(sys.argv) > 3 else 2
db = simreads.make_workload_torch(wl) if wl in simreads.TORCH_WORKLOADS else simreads.make_workload(wl)
d = tempfile.mkdtemp(prefix="l2diff_", dir="/dev/shm")
try:
    pre = os.path.join(d, "sd")
    formats.write_seqdb(pre, db)
    t = time.time()
    with cf.ThreadPoolExecutor(N) as ex:
        list(ex.map(lambda c: U.ref_run("shmr_index", "-p", pre, "-t", N, "-c", c, "-m", 0, "-l", lv, "-o", os.path.join(d, "ref")), range(1, N + 1)))
    print("reference index: %.1f s" % (time.time() - t), flush=True)
    exe = os.path.join(ROOT, "bin", "native", "shmr_index")
    for c in range(1, N + 1):
        subprocess.run([exe, "-p", pre, "-t", str(N), "-c", str(c), "-m", "0", "-l", str(lv), "-o", os.path.join(d, "nat")], check=True)
    rdb = ResidentDB(db, 0)
    bad = 0
    for c in range(1, N + 1):
        ref = formats.read_mmlist(os.path.join(d, "ref-L%d-%02d-of-%02d.dat" % (lv, c, N)))
        nat = formats.read_mmlist(os.path.join(d, "nat-L%d-%02d-of-%02d.dat" % (lv, c, N)))
        res = rdb.index(total_chunk=N, mychunk=c, levels=lv).top
        same_file = open(os.path.join(d, "ref-L%d-%02d-of-%02d.dat" % (lv, c, N)), "rb").read() == open(os.path.join(d, "nat-L%d-%02d-of-%02d.dat" % (lv, c, N)), "rb").read()
        print("chunk %d: ref %d  native %d (file identical: %s)  resident %d (equal ref: %s)" % (c, len(ref), len(nat), same_file, len(res), np.array_equal(ref, res)), flush=True)
        for name, got in (("native", nat), ("resident", res)):
            if np.array_equal(ref, got):
                continue
            bad += 1
            n = min(len(ref), len(got))
            neq = np.flatnonzero((ref["x"][:n] != got["x"][:n]) | (ref["y"][:n] != got["y"][:n]))
            i = int(neq[0]) if len(neq) else n
            r = int((ref["y"][min(i, len(ref) - 1)]) >> np.uint64(32))
            print("  %s differs first at element %d (read %d); ref %s  got %s" % (name, i, r, ref[max(0, i - 1):i + 3], got[max(0, i - 1):i + 3]))
            rr = ref[(ref["y"] >> np.uint64(32)) == np.uint64(r)]
            gg = got[(got["y"] >> np.uint64(32)) == np.uint64(r)]
            rb = np.asarray(db.seqdb[int(db.roff[r]):int(db.roff[r]) + int(db.rlen[r])])
            want = U.orc_sketch_seqdb(rb, 80, 16, r)
            l0 = want
            for _ in range(lv):
                want = U.orc_reduce(want, 6)
            print("  read %d (len %d): reference %d elements, %s %d, oracle %d; oracle == reference: %s; oracle L0 %d" % (r, len(rb), len(rr), name, len(gg), len(want), np.array_equal(want, rr), len(l0)))
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.save(os.path.join(ROOT, "gpurun_out", "l2diff_read_%s_%d.npy" % (name, c)), rb)
            if bad >= 4:
                break
    print("DIFFERENCES:", bad)
finally:
    shutil.rmtree(d, ignore_errors=True)
