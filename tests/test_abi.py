"""CPU-side checks of the drop-in boundary: libpgx.so loads, exports every symbol include/pgx.h declares, and
fails loudly (no CPU fallback) when asked to compute without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from peregrine_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(pgx_[a-z0-9_]+)\s*\(", hdr))
    names |= set(re.findall(r"\b(decode_biseq|encode_biseq|mm_sketch|mm_reduce|ovlp_match|free_ovlp_match|read_mmlist|write_mmlist|"
                            r"build_shimmer_map4py|get_shimmers_for_read|get_mmer_count|get_shimmer_hits)\s*\(", hdr))
    assert names == set(_lib.EXPORTS), names ^ set(_lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_struct_sizes_match_the_formats():
    assert C.sizeof(_lib.IndexParams) == 28 and C.sizeof(_lib.OverlapParams) == 28
    assert _lib.MATCH_DTYPE.itemsize == 32 and _lib.ALIGN_KEY_DTYPE.itemsize == 16


def test_codec_is_the_reference_format(lib):
    s = b"ACGTNacgtn"
    enc = np.zeros(len(s), np.uint8)
    lib.encode_biseq(enc.ctypes.data_as(C.c_void_p), C.c_char_p(s), C.c_size_t(len(s)))
    assert enc.tolist() == [0x01, 0x12, 0x24, 0x48, 0x80, 0x01, 0x12, 0x24, 0x48, 0x80]
    fwd = C.create_string_buffer(len(s)); rev = C.create_string_buffer(len(s))
    lib.decode_biseq(enc.ctypes.data_as(C.c_void_p), fwd, C.c_size_t(len(s)), C.c_uint8(0))
    lib.decode_biseq(enc.ctypes.data_as(C.c_void_p), rev, C.c_size_t(len(s)), C.c_uint8(1))
    assert fwd.raw == b"ACGTNACGTN"
    assert rev.raw == b"NACGTNACGT"  # reverse complement of the forward strand


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.pgx_init(0) != 0
    assert b"HIP" in lib.pgx_last_error() or b"device" in lib.pgx_last_error()
    with pytest.raises(_lib.PgxError):
        from peregrine_amd.shimmer import mm_count
        _lib._inited = None
        mm_count(np.zeros(4, dtype=[("x", "<u8"), ("y", "<u8")]))
