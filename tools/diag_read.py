#!/usr/bin/env python3
"""One read through every index path against the oracle (which list level first differs, under which kernel selection)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import oracle_util as U
    from peregrine_amd.formats import SeqDB
    from peregrine_amd.shimmer import ResidentDB
    rb = np.load(sys.argv[1])
    pad = [rb] * 3     # the same read three times (rids 0..2)
    rlen = np.array([len(e) for e in pad], np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    db = SeqDB(np.concatenate(pad), np.arange(len(pad), dtype=np.uint32), rlen, roff, None)
    rdb = ResidentDB(db, 0)
    l0 = np.concatenate([U.orc_sketch_seqdb(e, 80, 16, i) for i, e in enumerate(pad)])
    l1 = U.orc_reduce(l0, 6); l2 = U.orc_reduce(l1, 6)
    a = rdb.index(want_l0=True)
    print("  general path: L0 equal", np.array_equal(a.l0, l0), " L2 equal", np.array_equal(a.top, l2), "literal", a.reads_literal)
    b = rdb.index(levels=1); c = rdb.index(levels=2)
    print("  fused path: L1 equal", np.array_equal(b.top, l1), " L2 equal", np.array_equal(c.top, l2), "second-path reads", c.reads_literal)
    if not np.array_equal(b.top, l1):
        n = min(len(b.top), len(l1)); i = int(np.flatnonzero((b.top['x'][:n] != l1['x'][:n]) | (b.top['y'][:n] != l1['y'][:n]))[0])
        pos = lambda a: ((a['y'] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.int64)
        print("   L1 first differs at", i, "of", len(l1), len(b.top), " oracle pos", pos(l1)[i - 2:i + 6], " got pos", pos(b.top)[i - 2:i + 6])
        p = int(pos(l1)[i]); j = int(np.searchsorted(pos(l0[:len(l0) // 3]), p))
        print("   L0 index of that position in the read:", j, " L0 positions around", pos(l0)[j - 8:j + 4])
    sys.exit(0)
for env in ({}, {"PGX_SKETCH": "wave"}, {"PGX_SKETCH": "fuse"}, {"PGX_TRACE": "1"}):
    print("env", env, flush=True)
    r = subprocess.run([sys.executable, __file__, sys.argv[1], "child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout, end="")
    if "PGX_TRACE" in env or r.returncode:
        print("\n".join(l for l in r.stderr.splitlines() if "index:" in l or "Error" in l or "error" in l)[:3000])
