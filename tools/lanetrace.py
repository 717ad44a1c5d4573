#!/usr/bin/env python3
"""debug: one key through the lane kernel built with -DPGX_LANE_TRACE; prints the device trace beside the host's view of the data"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from peregrine_amd import _lib, simreads
from peregrine_amd.shimmer import ResidentDB
import oracle_util as U
db = simreads.make_workload("small")
rdb = ResidentDB(db, 0)
ix = rdb.index()
ov, st = rdb.overlap(ix.top, ix.top_mc)
i = int(sys.argv[1]) if len(sys.argv) > 1 else 0
key = np.zeros(1, _lib.ALIGN_KEY_DTYPE)
o = ov[i]
key["rid0"] = o["y0"] >> np.uint64(32); key["rid1"] = o["y1"] >> np.uint64(32)
key["q_off"] = ((int(o["y0"]) & 0xFFFFFFFF) >> 1) - ((int(o["y1"]) & 0xFFFFFFFF) >> 1); key["dir0"] = o["strand0"]; key["dir1"] = o["strand1"]
a, b = int(key["rid0"][0]), int(key["rid1"][0])
q = db.seqdb[int(db.roff[a]) + int(key["q_off"][0]):int(db.roff[a]) + int(db.rlen[a])]
t = db.seqdb[int(db.roff[b]):int(db.roff[b]) + int(db.rlen[b])]
code = {1: 0, 2: 1, 4: 2, 8: 3}
def pack(seq, sh, n=16):
    v = 0
    for j in range(n): v |= code.get((int(seq[j]) >> sh) & 15, 0) << (2 * j)
    return v
print("key", key[0], "qg", int(db.roff[a]) + int(key["q_off"][0]), "tg", int(db.roff[b]))
print("host: first 16 bases packed q %08x t %08x" % (pack(q, 4 if key["dir0"][0] else 0), pack(t, 4 if key["dir1"][0] else 0)))
print("oracle", U.orc_ovlp_match(q, int(key["dir0"][0]), t, int(key["dir1"][0]), 100))
os.environ["PGX_ALIGN_LANE_MIN"] = "0"; os.environ["PGX_TRACE"] = "1"
print("lane  ", rdb.align(key, 100)[0])
