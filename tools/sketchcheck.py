#!/usr/bin/env python3
"""L0 of the fused index path (k_sketch_blk + the general kernel for the reads it flags) against the CPU oracle, read by read:
index(levels=1, reduction=1) is the identity reduce, so the list that comes back IS the L0 sketch.  Adversarial reads: planted
reverse-palindromic 16-mers ("drops") at chosen places (first window, tile borders, close pairs, read end), short reads, N,
homopolymers, tandem repeats.   usage: tools/sketchcheck.py [n_random_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_util as U
from peregrine_amd import _lib
from peregrine_amd.formats import SeqDB
from peregrine_amd.shimmer import ResidentDB

rng = np.random.default_rng(11)
COMP = np.array([3, 2, 1, 0], np.uint8)


def pal16():
    h = rng.integers(0, 4, 8, dtype=np.uint8)
    return np.concatenate([h, COMP[h[::-1]]])


def encode(codes):
    n = len(codes)
    return ((np.uint8(1) << codes) | ((np.uint8(8) >> codes[::-1]) << np.uint8(4))).astype(np.uint8)


reads, tags = [], []
def add(codes, tag):
    reads.append(np.asarray(codes, np.uint8)); tags.append(tag)

nrand = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for i in range(nrand):
    add(rng.integers(0, 4, int(rng.integers(400, 20000)), dtype=np.uint8), "random")
for pos in [0, 5, 60, 94, 110, 150, 199, 200, 201, 300, 990, 1000, 1008, 1009, 1010, 1020, 1023, 1024, 1025, 1040, 1100, 2040, 2048, 2050, 5000]:
    for ln in (6000, 6001, 6007, 6015):
        c = rng.integers(0, 4, ln, dtype=np.uint8); c[pos:pos + 16] = pal16(); add(c, f"pal@{pos}")
for gap in [1, 2, 8, 16, 17, 40, 79, 80, 81, 100, 200, 500, 1000, 1200, 1300, 1500, 3000]:
    c = rng.integers(0, 4, 9000, dtype=np.uint8); c[3000:3016] = pal16(); c[3000 + gap:3016 + gap] = pal16(); add(c, f"pal2 gap {gap}")
for back in [16, 17, 20, 50, 79, 80, 81, 95, 96, 97, 120, 300]:
    for ln in (4096 + 7, 5000, 5120 - 15):
        c = rng.integers(0, 4, ln, dtype=np.uint8); c[ln - back:ln - back + 16] = pal16()[:min(16, back)]; add(c, f"pal@end-{back}")
for ln in [16, 17, 50, 94, 95, 96, 100, 111, 200, 294, 295, 296, 300, 1000, 1008, 1009, 1023, 1024, 1025, 1040, 1103, 1104, 1105, 2047, 2048, 2049]:
    add(rng.integers(0, 4, ln, dtype=np.uint8), f"len {ln}")
for _ in range(10):
    c = rng.integers(0, 4, 8000, dtype=np.uint8); s0 = int(rng.integers(100, 7000)); c[s0:s0 + int(rng.integers(20, 600))] = rng.integers(0, 4); add(c, "homopolymer")
for _ in range(10):
    c = rng.integers(0, 4, 8000, dtype=np.uint8); s0 = int(rng.integers(100, 6000)); per = int(rng.integers(2, 30)); n = int(rng.integers(100, 1500))
    c[s0:s0 + n] = np.resize(rng.integers(0, 4, per, dtype=np.uint8), n); add(c, "tandem")
for _ in range(6):
    c = rng.integers(0, 4, 3000, dtype=np.uint8); c[::2] = 0; c[1::2] = 3; add(c, "(AT)n")     # every 16-mer is its own reverse complement

enc = [encode(c) for c in reads]
for j in (3, 40):   # ambiguous bases
    e = encode(rng.integers(0, 4, 5000, dtype=np.uint8)); e[1234] = 0; enc.append(e); reads.append(None); tags.append("N")
rlen = np.array([len(e) for e in enc], np.uint32)
roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
db = SeqDB(np.concatenate(enc), np.arange(len(enc), dtype=np.uint32), rlen, roff, None)
rdb = ResidentDB(db, 0)
os.environ["PGX_TRACE"] = "1"
ix = rdb.index(levels=1, reduction=1)
os.environ.pop("PGX_TRACE")
rid = (ix.top["y"] >> np.uint64(32)).astype(np.int64)
starts = np.searchsorted(rid, np.arange(len(enc) + 1))
bad = {}
t0 = time.time()
for r in range(len(enc)):
    want = U.orc_sketch_seqdb(enc[r], 80, 16, r)
    got = ix.top[starts[r]:starts[r + 1]]
    if not np.array_equal(got, want):
        bad.setdefault(tags[r], []).append(r)
        if sum(len(v) for v in bad.values()) <= 12:
            n = min(len(got), len(want))
            d = next((i for i in range(n) if got[i] != want[i]), n)
            gp = [int((g["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)) for g in got[max(0, d - 1):d + 3]]
            wp = [int((g["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)) for g in want[max(0, d - 1):d + 3]]
            print(f"MISMATCH read {r} [{tags[r]}] len {rlen[r]} lead {int(roff[r]) & 15}: got {len(got)} want {len(want)}, first diff at #{d}: got pos {gp} want pos {wp}")
print("reads", len(enc), "mismatching by tag:", {k: len(v) for k, v in bad.items()}, f"(oracle {time.time() - t0:.1f}s)")
# the default two-level path too (levels=2, reduction=6) on the same set
ix2 = rdb.index()
l2 = np.concatenate([U.orc_reduce(U.orc_reduce(U.orc_sketch_seqdb(enc[r], 80, 16, r), 6), 6) for r in range(len(enc))])
print("L2 equal:", bool(np.array_equal(ix2.top, l2)), len(ix2.top), len(l2))
sys.exit(1 if bad or not np.array_equal(ix2.top, l2) else 0)
