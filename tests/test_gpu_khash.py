"""klib-khash's slot layout as the overlap stage computes it for its OUTER table (pgx_khash.h: DistinctSlotTable -- one packed word
per slot, a skip count per home slot instead of re-walking probe chains, the kick-out rehash of khash.h:258-284) against the
oracle's literal put-by-put emulation (oracle/: otab_put, restating /root/reference/src/khash.h:232-336).  (The inner tables are
replayed on the device, pgx_visit.hip: tests/test_gpu_parity.py::test_device_visit_order_equals_oracle_and_host_visit.)"""
import ctypes as C

import numpy as np
import pytest

import oracle_util as U
from peregrine_amd import _lib

pytestmark = pytest.mark.gpu


def _order(lib, keys, touch):
    got = np.zeros(len(keys), np.uint64)
    rc = lib.pgx_khash_slot_order_ex(keys.ctypes.data_as(C.c_void_p), len(keys), touch, got.ctypes.data_as(C.c_void_p))
    return rc, got


def test_khash_layout_equals_the_literal_replay():
    lib = _lib.load()
    rng = np.random.default_rng(21)
    cases = [("one", np.array([5], np.uint64)), ("random64", rng.integers(0, 1 << 63, 300_000, dtype=np.uint64))]
    # keys shaped like the real ones: (32-bit minimizer hash) << 8 | span.  Minimizers are the SMALLEST hashes of their windows, so
    # the hash's top bits are zero and khash's integer hash leaves the span in the low index bits: long probe chains.
    # (the reference's keys have span = k = 16 throughout: 1 / 256 of the slots are the home of ALL keys -- the case the skip counts exist for)
    for n, hbits, spans in ((200_000, 24, [16]), (2, 20, [16]), (3, 20, [16, 17]), (4, 20, [16]), (7, 20, [16]), (13, 20, [20, 30]), (1000, 16, [16, 17, 19]),
                            (50_000, 22, [16, 16, 16, 17, 19]), (400_000, 23, list(range(28, 90))), (1_500_000, 23, list(range(28, 120)))):
        h = rng.choice(1 << hbits, n, replace=False).astype(np.uint64)
        span = rng.choice(np.array(spans, np.uint64), n)
        cases.append((f"shimmer-like {n}", (h << np.uint64(8)) | span))
    # every power-of-two boundary of the load factor: a resize exactly at / just before / just after the last key
    for nb in (4, 8, 16, 32, 1024, 65536):
        up = int(nb * 0.77 + 0.5)
        for n in (up - 1, up, up + 1):
            cases.append((f"boundary {n}", rng.integers(0, 1 << 40, n, dtype=np.uint64)))
    # all keys with (nearly) the same home: a probe chain as long as the table
    cases.append(("degenerate chain", np.arange(1, 60_001, dtype=np.uint64) << np.uint64(32)))
    for name, keys in cases:
        keys = np.unique(keys)
        keys = keys[rng.permutation(len(keys))]      # distinct, arbitrary insertion order
        for touch in (0, 1):
            want = U.orc_khash_order(np.concatenate([keys, keys[:1]]) if touch else keys)   # (a put of a present key: only the load check)
            rc, got = _order(lib, keys, touch)
            assert rc == 0 and np.array_equal(got, want), (name, len(keys), touch)
