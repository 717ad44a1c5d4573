"""GPU: klib-khash's slot layout computed on the device (pgx_khash_dev.hip) -- priority insertion between the resizes, a fixed
point over the eviction order inside a resize -- against the oracle's literal put-by-put emulation (oracle/: otab_put, restating
/root/reference/src/khash.h:232-336) and against the host form the overlap stage uses for small tables."""
import ctypes as C

import numpy as np
import pytest

import oracle_util as U
from peregrine_amd import _lib

pytestmark = pytest.mark.gpu


def _order(lib, keys, touch, dev):
    got = np.zeros(len(keys), np.uint64)
    rc = lib.pgx_khash_slot_order_ex(keys.ctypes.data_as(C.c_void_p), len(keys), touch, dev, got.ctypes.data_as(C.c_void_p))
    return rc, got


def test_device_khash_layout_equals_the_literal_replay():
    lib = _lib.load()
    _lib.init()
    rng = np.random.default_rng(21)
    cases = [("one", np.array([5], np.uint64)), ("random64", rng.integers(0, 1 << 63, 300_000, dtype=np.uint64))]
    # keys shaped like the real ones: (32-bit minimizer hash) << 8 | span.  Minimizers are the SMALLEST hashes of their windows, so
    # the hash's top bits are zero and khash's integer hash leaves the span in the low index bits: long probe chains.
    # (the reference's keys have span = k = 16 throughout: 1 / 256 of the slots are the home of ALL keys -- the case the host form's skip
    #  counts exist for; the device form walks those chains, which is why the stage does not use it by default)
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
    for name, keys in cases:
        keys = np.unique(keys)
        keys = keys[rng.permutation(len(keys))]      # distinct, arbitrary insertion order
        for touch in (0, 1):
            want = U.orc_khash_order(np.concatenate([keys, keys[:1]]) if touch else keys)   # (a put of a present key: only the load check)
            rc, got = _order(lib, keys, touch, 1)
            assert rc == 0 and np.array_equal(got, want), (name, len(keys), touch, "device")
            rc, got = _order(lib, keys, touch, 0)
            assert rc == 0 and np.array_equal(got, want), (name, len(keys), touch, "host")


def test_device_khash_gives_up_on_a_degenerate_chain():
    """all keys with the same home: a probe chain as long as the table -- the device form runs out of its probe budget and says so
    (PGX_ESTATE: the stage then takes the host form, whose skip counts make such chains cheap); the host form's answer is checked"""
    lib = _lib.load()
    _lib.init()
    n = 60_000
    keys = (np.arange(1, n + 1, dtype=np.uint64) << np.uint64(34))     # khash's hash of k << 34 is k << 1 ^ k << 45 (low 32 bits): spread ...
    keys = (np.arange(1, n + 1, dtype=np.uint64) << np.uint64(32))     # ... this one: (k >> 1) ^ 0 in the low bits -> collisions in pairs only
    rc, got = _order(lib, keys, 0, 1)
    assert rc in (0, -5)   # PGX_ESTATE
    if rc == 0:
        assert np.array_equal(got, U.orc_khash_order(keys))
    rc, got = _order(lib, keys, 0, 0)
    assert rc == 0 and np.array_equal(got, U.orc_khash_order(keys))
