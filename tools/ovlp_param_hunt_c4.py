#!/usr/bin/env python3
"""Overlap-stream parity at FULL size under NON-default stage parameters: BASELINE configs[3] (93 Gbases), one overlap chunk of T per
parameter set, the GPU's ovlp_t stream against oracle/_ref/shmr_overlap with the same flags on the same files, field by field.
(The suite pins the default parameters at this size; this hunts the dispatch corners: bestn 1 / 8, narrow and wide bands -- other V-ring
sizes --, small and large ovlp_upper, a low multiplicity cut-off, mc_lower 1.)
  python tools/ovlp_param_hunt_c4.py [T=384] [genome_mb=3100 (0: full)] [levels=2] [first n sets]
(levels = 1: BASELINE configs[4]'s dense L1 shimmers at full size -- 776 M of them; two sets are plenty there: each reference process
loads all of them)"""
import os, sys, tempfile, shutil, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import concurrent.futures as cf
import torch
import oracle_util as U
from peregrine_amd import _lib, formats, simreads
from peregrine_amd.parallel import GpuEngine
from peregrine_amd.shimmer import ResidentDB

T = int(sys.argv[1]) if len(sys.argv) > 1 else 384
gmb = (float(sys.argv[2]) or None) if len(sys.argv) > 2 else None
LV = int(sys.argv[3]) if len(sys.argv) > 3 else 2
NSETS = int(sys.argv[4]) if len(sys.argv) > 4 else 99
SETS = [   # (chunk, reference flags, library keywords)
    (5, ["-b", 1], dict(bestn=1)),
    (11, ["-b", 8], dict(bestn=8)),
    (17, ["-w", 30], dict(align_bandwidth=30)),
    (23, ["-w", 130], dict(align_bandwidth=130)),
    (29, ["-n", 24], dict(ovlp_upper=24)),
    (35, ["-n", 128, "-M", 400], dict(ovlp_upper=128, mc_upper=400)),
    (41, ["-M", 60], dict(mc_upper=60)),
    (47, ["-m", 1, "-b", 2, "-w", 60], dict(mc_lower=1, bestn=2, align_bandwidth=60)),
]
SETS = SETS[:NSETS]
seq, total, rlen = simreads.make_workload_resident("c4", genome_mb=gmb)
rid = np.arange(len(rlen), dtype=np.uint32)
roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
rdb = ResidentDB.adopt_device(seq, total, rid, rlen, roff, 0)
eng = GpuEngine(rdb, torch.device("cuda", 0))
d = tempfile.mkdtemp(prefix="pgx_hunt_", dir="/dev/shm")
try:
    pre = os.path.join(d, "sd")
    simreads.write_seqdb_from_device(pre, seq, total, rid, rlen, roff)
    N = 8
    tops, mcs = [], []
    for c in range(1, N + 1):
        p = rdb.index(total_chunk=N, mychunk=c, levels=LV)
        formats.write_mmlist(os.path.join(d, "ix-L%d-%02d-of-%02d.dat" % (LV, c, N)), p.top)
        formats.write_mm_count(os.path.join(d, "ix-L%d-MC-%02d-of-%02d.dat" % (LV, c, N)), p.top_mc)
        tops.append(torch.from_numpy(p.top.view(np.uint8)).cuda()); mcs.append(torch.from_numpy(p.top_mc.view(np.uint8)).cuda())
    mm, mc = torch.cat(tops), torch.cat(mcs)
    del tops, mcs
    t = time.time()
    with cf.ThreadPoolExecutor(len(SETS)) as ex:
        list(ex.map(lambda s: U.ref_run("shmr_overlap", "-p", pre, "-l", os.path.join(d, "ix-L%d" % LV), "-t", T, "-c", s[0], *s[1], "-o", os.path.join(d, "ref.%d" % s[0])), SETS))
    print("reference: %d overlap chunks of %d in %.0f s" % (len(SETS), T, time.time() - t), flush=True)
    _lib.stream_wait()
    bad = 0
    for c, flags, kw in SETS:
        want = formats.read_ovlp(os.path.join(d, "ref.%d" % c))
        ov, st = rdb.overlap_dev(mm.data_ptr(), mm.numel() // 16, mc.data_ptr(), mc.numel() // 16, total_chunk=T, mychunk=c, **kw)
        ok = formats.ovlp_fields_equal(np.asarray(ov), want)
        bad += 0 if ok else 1
        print("chunk %d of %d, flags %s: reference %d records, GPU %d (device replay %d, sweeps %d) -> %s" % (c, T, " ".join(map(str, flags)), len(want), len(ov), st["device_replay"], st["rounds"], "EQUAL" if ok else "DIFFERENT"), flush=True)
    print("PARAMETER SETS %d, DIFFERENCES %d" % (len(SETS), bad))
finally:
    shutil.rmtree(d, ignore_errors=True)
