#!/usr/bin/env python3
"""Golden vectors for row f1 (shmr_mkseqdb): small FASTA/FASTQ inputs and the REAL reference binary's outputs.
Run in the build container: python tests/golden/make_golden_mkseqdb.py"""
import gzip
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_util as U  # noqa: E402


def main():
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rnd(n):
        return acgt[rng.integers(0, 4, n)].tobytes()

    files = {}
    files["a.fa"] = b">r1 some comment here\n" + rnd(300) + b"\n>r2\n" + rnd(17) + b"\n" + rnd(60) + b"\n\n" + rnd(5) + b"\n>r3\tx\nACGTNNacgtnRYKM\n"
    fq = b""
    for i in range(4):
        s = rnd(int(rng.integers(20, 200)))
        q = bytes(rng.integers(33, 74, len(s)).astype(np.uint8))
        fq += b"@q%d desc\n" % i + s + b"\n+\n" + q + b"\n"
    s = rnd(50)
    fq += b"@q_multi\n" + s[:20] + b"\n" + s[20:] + b"\n+q_multi\n" + b"I" * 20 + b"\n" + b"@" * 30 + b"\n"   # '@' inside quality
    files["b.fq"] = fq
    files["c.fa.gz"] = gzip.compress(b">g1\r\n" + rnd(40) + b"\r\n" + rnd(33) + b"\r\n>g2\r\nAC\r\n>empty\n>g3\n" + rnd(1000) + b"\n")
    files["d.fa"] = b"junk before first header\n>d1\n" + rnd(64) + b"\n>d2 trailing record without newline\n" + rnd(31)
    tmp = tempfile.mkdtemp()
    order = ["a.fa", "b.fq", "c.fa.gz", "d.fa"]
    for k in order:
        open(os.path.join(tmp, k), "wb").write(files[k])
    open(os.path.join(tmp, "seq.lst"), "w").write("\n".join(os.path.join(tmp, k) for k in order) + "\n")
    U.ref_run("shmr_mkseqdb", "-p", os.path.join(tmp, "ref"), "-d", os.path.join(tmp, "seq.lst"))
    store = {"order": np.array(order)}
    for k in order:
        store["file_" + k] = np.frombuffer(files[k], np.uint8)
    store["seqdb"] = np.fromfile(os.path.join(tmp, "ref.seqdb"), np.uint8)
    store["idx"] = np.frombuffer(open(os.path.join(tmp, "ref.idx"), "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "mkseqdb_cases.npz"), **store)
    print(open(os.path.join(tmp, "ref.idx")).read())
    print(len(store["seqdb"]))


if __name__ == "__main__":
    main()
