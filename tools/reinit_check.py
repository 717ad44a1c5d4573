import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from peregrine_amd import _lib, simreads, formats
from peregrine_amd.shimmer import ResidentDB
db = simreads.make_workload("small")
outs = []
for it in range(3):
    rdb = ResidentDB(db, 0)
    ix, ov, st = rdb.index_overlap()
    ix2 = rdb.index()
    ov2, st2 = rdb.overlap(ix2.top, ix2.top_mc)
    assert formats.ovlp_fields_equal(ov, ov2)
    outs.append(ov.copy())
    rdb.close()
    _lib.shutdown()
    _lib._inited = None
assert all(formats.ovlp_fields_equal(outs[0], o) for o in outs)
print("re-init ok", len(outs[0]))
