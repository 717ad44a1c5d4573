"""Loaders for the committed golden fixtures (outputs of the real reference; see tests/golden/make_golden.py)."""
import os

import numpy as np

from peregrine_amd.formats import SeqDB

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def tiny_db(z=None) -> SeqDB:
    z = z if z is not None else load("tiny_stage.npz")
    rlen = z["rlen"].astype(np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    names = [ln.split()[1] for ln in z["idx_text"].tobytes().decode().strip().split("\n")]
    return SeqDB(z["seqdb"].copy(), np.arange(len(rlen), dtype=np.uint32), rlen, roff, names)


def sketch_cases():
    z = load("sketch_cases.npz")
    blob = z["blob"].tobytes()
    for i in range(len(z["w"])):
        yield (i, blob[z["soff"][i]:z["soff"][i + 1]], int(z["w"][i]), int(z["k"][i]),
               z["out"][z["ooff"][i]:z["ooff"][i + 1]])


def reduce_cases():
    z = load("reduce_cases.npz")
    for i in range(len(z["rs"])):
        yield (i, z["inp"][z["ioff"][i]:z["ioff"][i + 1]], int(z["rs"][i]), z["out"][z["ooff"][i]:z["ooff"][i + 1]])


def match_cases():
    z = load("match_cases.npz")
    for i in range(len(z["band"])):
        yield (i, z["q"][z["qoff"][i]:z["qoff"][i + 1]], int(z["qs"][i]), z["t"][z["toff"][i]:z["toff"][i + 1]],
               int(z["ts"][i]), int(z["band"][i]), tuple(int(v) for v in z["out"][i]))


OVERLAP_RUNS = {  # name: (index chunks, level, overlap chunks, kwargs)  -- mirrors make_golden.py
    "ov_i1_t1": (1, 2, 1, {}),
    "ov_i2_t1": (2, 2, 1, {}),
    "ov_i2_t2": (2, 2, 2, {}),
    "ov_i2_t3": (2, 2, 3, {}),
    "ov_l1_t1": (2, 1, 1, {}),
    "ov_par_t2": (2, 2, 2, dict(bestn=2, mc_upper=30, band=60, ovlp_upper=40, mc_lower=2)),
}


def query_fixture():
    """rows f3/f4: the tiny set's two-chunk level-2 index (inputs) + query_cases.npz (the reference's answers)"""
    from peregrine_amd import formats
    q, t = load("query_cases.npz"), load("tiny_stage.npz")
    mmers = np.concatenate([t["ix2l2_L2_1"], t["ix2l2_L2_2"]])
    pairs = np.concatenate([t["ix2l2_L2MC_1"], t["ix2l2_L2MC_2"]])
    mc = np.zeros(len(pairs), formats.MC_DTYPE)
    mc["mer"], mc["count"] = pairs[:, 0], pairs[:, 1]
    return q, mmers, mc, t["rlen"].astype(np.uint32)
