#!/usr/bin/env python3
"""The stages either side of the hot path at Gbase scale against the reference binaries:
  f2  cat ovlp.* | shmr_dedup : the records of ALL 8 overlap chunks of the 9-Gbase repeat-seeded set (36.9 M records) -> text, byte for byte
  f1  shmr_mkseqdb            : the 4.5-Gbase set as FASTA (300 k records, 4.6 GB) -> .seqdb / .idx, byte for byte
  python tools/scale_aux.py [dedup] [mkseqdb]"""
import os, subprocess, sys, tempfile, shutil, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_util as U
from peregrine_amd import formats, simreads
from peregrine_amd.shimmer import ResidentDB, shmr_dedup, shmr_mkseqdb

what = set(sys.argv[1:]) or {"dedup", "mkseqdb"}
d = tempfile.mkdtemp(prefix="pgx_aux_", dir="/dev/shm")
try:
    if "dedup" in what:
        db = simreads.make_workload_torch("c4s")
        rdb = ResidentDB(db, 0)
        N = 8
        parts = [rdb.index(total_chunk=N, mychunk=c) for c in range(1, N + 1)]
        mm = np.concatenate([p.top for p in parts]); mc = np.concatenate([p.top_mc for p in parts])
        paths = []
        for c in range(1, N + 1):
            ov, _ = rdb.overlap(mm, mc, total_chunk=N, mychunk=c)
            paths.append(os.path.join(d, "ovlp.%02d" % c))
            np.asarray(ov).tofile(paths[-1])
        rdb.close()
        t = time.time()
        text, nu = shmr_dedup(paths)
        t_gpu = time.time() - t
        t = time.time()
        with open(os.path.join(d, "ref.ovl"), "wb") as f:
            cat = subprocess.Popen(["cat", *paths], stdout=subprocess.PIPE)
            subprocess.run([os.path.join(U.REF_DIR, "shmr_dedup")], stdin=cat.stdout, stdout=f, check=True)
            cat.wait()
        t_ref = time.time() - t
        ref = open(os.path.join(d, "ref.ovl"), "rb").read()
        print("dedup: %d records in, %d unique pairs, %d bytes of text; GPU %.1f s, reference %.1f s; IDENTICAL: %s" % (sum(os.path.getsize(p) for p in paths) // 64, nu, len(ref), t_gpu, t_ref, ref == text), flush=True)
        del db, mm, mc, text, ref
    if "mkseqdb" in what:
        db = simreads.make_workload_torch("c3")
        lut = np.full(16, ord("N"), np.uint8); lut[[1, 2, 4, 8]] = [ord(c) for c in "ACGT"]
        fa = os.path.join(d, "reads.fa")
        with open(fa, "wb") as f:
            for r in range(db.n_reads):
                o, n = int(db.roff[r]), int(db.rlen[r])
                f.write(b">read%07d\n" % r); f.write(lut[db.seqdb[o:o + n] & 0x0F].tobytes()); f.write(b"\n")
        open(os.path.join(d, "lst"), "w").write(fa + "\n")
        t = time.time(); U.ref_run("shmr_mkseqdb", "-p", os.path.join(d, "ref"), "-d", os.path.join(d, "lst")); t_ref = time.time() - t
        t = time.time(); st = shmr_mkseqdb(os.path.join(d, "lst"), os.path.join(d, "gpu")); t_gpu = time.time() - t
        same = all(open(os.path.join(d, "ref" + e), "rb").read() == open(os.path.join(d, "gpu" + e), "rb").read() for e in (".seqdb", ".idx"))
        same_as_sim = open(os.path.join(d, "gpu.seqdb"), "rb").read() == np.asarray(db.seqdb).tobytes()
        print("mkseqdb: %s; GPU %.1f s, reference %.1f s; .seqdb and .idx IDENTICAL to the reference's: %s; .seqdb identical to the simulator's encoding: %s" % (st, t_gpu, t_ref, same, same_as_sim), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
