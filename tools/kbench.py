#!/usr/bin/env python3
"""Kernel-level throughput probe (not the graded bench): sketch kernel on an R-times replicated read set, align
kernel on the E. coli-size overlap run.  usage: tools/kbench.py [R] [both|sketch|align]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peregrine_amd import _lib, simreads
from peregrine_amd.formats import SeqDB
from peregrine_amd.shimmer import ResidentDB

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
what = sys.argv[2] if len(sys.argv) > 2 else "both"
base = simreads.make_workload("ecoli")
if what in ("both", "sketch"):
    rlen = np.tile(base.rlen, R)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    db = SeqDB(np.tile(base.seqdb, R), np.arange(len(rlen), dtype=np.uint32), rlen, roff)
    rdb = ResidentDB(db, 0)
    for it in range(3):
        _lib.timing_reset()
        t0 = time.perf_counter(); ix = rdb.index(); t1 = time.perf_counter()
        ms, n, units = _lib.timing("sketch")
        print(f"[sketch x{R}] bases {units} kernel {ms:.3f} ms -> {units/ms/1e6:.1f} Gbases/s = {1.04*units/ms/1e6/8000*100:.2f}% HBM; "
              f"index stage wall {1e3*(t1-t0):.1f} ms; reduce {_lib.timing('reduce')[0]:.2f} ms count {_lib.timing('count')[0]:.2f} ms "
              f"gather {_lib.timing('sketch_gather')[0]:.2f} ms literal reads {ix.reads_literal} L2 {len(ix.top)}", flush=True)
    rdb.close()
if what in ("both", "align"):
    rdb = ResidentDB(base, 0)
    ix = rdb.index()
    for it in range(3):
        _lib.timing_reset()
        t0 = time.perf_counter(); ov, st = rdb.overlap(ix.top, ix.top_mc); t1 = time.perf_counter()
        ms, n, units = _lib.timing("align")
        print(f"[overlap] records {len(ov)} wall {1e3*(t1-t0):.1f} ms host {st['host_ms']:.1f} gpu {st['gpu_ms']:.1f} rounds {st['rounds']} "
              f"align kernel {ms:.2f} ms / {units} aln = {units/ms/1e3:.2f} M aln/s (needed {st['n_align_needed']})", flush=True)
