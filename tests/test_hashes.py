"""The stream hashes bench.py reports and the pins they are compared with (CPU; SURVEY 8c/d, VERDICT r4 task 3):
 * formats.stream_checksum -- the numpy statement of pgx_overlap_stats.stream_checksum (k_emit adds it up on the GPU;
   tests/test_gpu_parity.py compares the two): position-sensitive, field-sensitive, blind to the padding bytes;
 * formats.masked_stream_sha256 -- SHA-256 with the padding bytes (27, 60..63) zeroed, from an array or a file;
 * tests/golden/c4_stream_pins.json -- well-formed: the 8 chunks of full-size configs[3], their records adding up to the job's count."""
import hashlib
import json
import os

import numpy as np

from peregrine_amd import formats

HERE = os.path.dirname(os.path.abspath(__file__))


def _records(n, seed):
    rng = np.random.default_rng(seed)
    ov = np.zeros(n, formats.OVLP_DTYPE)
    for f in formats.OVLP_FIELDS:
        if ov[f].dtype.kind == "i":
            ov[f] = rng.integers(-5, 1 << 30, n)
        elif ov[f].dtype.itemsize == 1:
            ov[f] = rng.integers(0, 3, n)
        else:
            ov[f] = rng.integers(0, 1 << 62, n).astype(ov[f].dtype)
    return ov


def _scalar_checksum(ov):
    M = (1 << 64) - 1

    def mix(h):
        h ^= h >> 33
        h = (h * 0xff51afd7ed558ccd) & M
        h ^= h >> 33
        h = (h * 0xc4ceb9fe1a85ec53) & M
        return h ^ (h >> 33)
    tot = 0
    for i, o in enumerate(ov):
        h = mix((int(o["y0"]) + 0x9E3779B97F4A7C15 * (i + 1)) & M)
        h = mix(h ^ int(o["y1"]))
        h = mix(h ^ (int(o["rl0"]) | int(o["rl1"]) << 32))
        h = mix(h ^ (int(o["strand0"]) | int(o["strand1"]) << 8 | int(o["ovlp_type"]) << 16))
        for a, b in (("m_size", "dist"), ("q_bgn", "q_end"), ("t_bgn", "t_end"), ("t_m_end", "q_m_end")):
            h = mix(h ^ ((int(o[a]) & 0xFFFFFFFF) | (int(o[b]) & 0xFFFFFFFF) << 32))
        tot = (tot + h) & M
    return tot


def test_stream_checksum_is_the_scalar_statement_and_sees_order_and_fields():
    ov = _records(700, 1)
    want = _scalar_checksum(ov)
    assert formats.stream_checksum(ov) == want and formats.stream_checksum(ov, block=97) == want
    assert formats.stream_checksum(ov[:0]) == 0
    swapped = ov.copy()
    swapped[[3, 4]] = swapped[[4, 3]]
    assert formats.stream_checksum(swapped) != want                     # order
    for f in formats.OVLP_FIELDS:                                        # every field
        ch = ov.copy()
        ch[f][11] ^= 1
        assert formats.stream_checksum(ch) != want, f
    pad = ov.copy()
    pad["pad0"][:] = 7
    pad["pad1"][:] = 0xDEADBEEF
    assert formats.stream_checksum(pad) == want                          # not the padding


def test_masked_sha256_of_arrays_and_files(tmp_path):
    ov = _records(5000, 2)
    clean = ov.copy()
    clean["pad0"][:] = 0
    clean["pad1"][:] = 0
    want = hashlib.sha256(clean.tobytes()).hexdigest()
    dirty = ov.copy()
    dirty["pad0"][:] = 0x5A
    dirty["pad1"][:] = 0x12345678                                        # what the reference's stack leaves there
    assert formats.masked_stream_sha256(dirty) == want and formats.masked_stream_sha256(dirty, block=333) == want
    p = tmp_path / "ov.dat"
    dirty.tofile(p)
    assert formats.masked_stream_sha256(str(p)) == want
    other = dirty.copy()
    other["q_end"][4999] += 1
    assert formats.masked_stream_sha256(other) != want
    assert formats.masked_stream_sha256(ov[:0]) == hashlib.sha256(b"").hexdigest()


def test_full_size_pins_are_well_formed():
    pins = json.load(open(os.path.join(HERE, "golden", "c4_stream_pins.json")))["c4"]
    assert pins["chunks"] == 8 and pins["levels"] == 2 and pins["mc_upper"] == 240 and pins["genome_mb"] is None
    assert pins["reads"] == 6_200_080 and pins["seqdb_bytes"] == 93_310_610_680
    assert [s["chunk"] for s in pins["streams"]] == ["%d of 8" % c for c in range(1, 9)]
    assert sum(s["records"] for s in pins["streams"]) == 366_003_067     # records_per_step of the graded line
    assert all(len(s["masked_sha256"]) == 64 and int(s["masked_sha256"], 16) >= 0 for s in pins["streams"])
    assert len(set(s["masked_sha256"] for s in pins["streams"])) == 8 and len(pins["seqdb_sha256"]) == 64 and len(pins["read_set_hash"]) == 34


def test_read_set_hash_covers_every_byte():
    """bench.device_read_set_hash (what the ranks of a multi-GPU job compare, and what ties a run to the pins): any single byte changes it, the
    blocked form equals the one-shot form, trailing bytes beyond a multiple of 8 count too (runs on a CPU tensor here, on the device in bench.py)"""
    import torch
    import bench
    rng = np.random.default_rng(3)
    n = (1 << 20) + 5
    a = torch.from_numpy(rng.integers(0, 256, n + 1024, dtype=np.uint8))
    h = bench.device_read_set_hash(a, n)
    assert len(h) == 34 and h == bench.device_read_set_hash(a.clone(), n)
    for pos in (0, 7, 8, 123457, n - 6, n - 1):
        b = a.clone()
        b[pos] ^= 1
        assert bench.device_read_set_hash(b, n) != h, pos
    b = a.clone()
    b[n + 3] ^= 0xFF                                    # beyond `total`: the zero tail is not part of the read set
    assert bench.device_read_set_hash(b, n) == h
    c = a.clone()
    c[[10, 18]] = c[[18, 10]]                           # the same bytes in another place
    assert (c[10] == a[10]) or bench.device_read_set_hash(c, n) != h
