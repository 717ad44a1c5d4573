"""ctypes binding of libpgx.so (the C-ABI declared in include/pgx.h).  No CPU fallback: if the HIP library is
missing or no GPU is visible, every compute entry point raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .formats import MC_DTYPE, MM_DTYPE, OVLP_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PGX_LIB", os.path.join(_HERE, "libpgx.so"))   # (PGX_LIB: an instrumented build, e.g. -DPGX_ALIGN_STATS)

MATCH_DTYPE = np.dtype([(f, "<i4") for f in ("m_size", "dist", "q_bgn", "q_end", "t_bgn", "t_end", "t_m_end", "q_m_end")])
ALIGN_KEY_DTYPE = np.dtype([("rid0", "<u4"), ("rid1", "<u4"), ("q_off", "<u4"), ("dir0", "u1"), ("dir1", "u1"), ("pad", "u1", 2)])
assert MATCH_DTYPE.itemsize == 32 and ALIGN_KEY_DTYPE.itemsize == 16


class PgxError(RuntimeError):
    pass


class IndexParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("total_chunk", "mychunk", "levels", "reduction", "window", "kmer", "want_l0")]


class IndexResult(C.Structure):
    _fields_ = [("l0", C.c_void_p), ("n_l0", C.c_size_t), ("l0_mc", C.c_void_p), ("n_l0_mc", C.c_size_t),
                ("top", C.c_void_p), ("n_top", C.c_size_t), ("top_mc", C.c_void_p), ("n_top_mc", C.c_size_t),
                ("bases", C.c_uint64), ("reads", C.c_uint32), ("reads_literal", C.c_uint32), ("gpu_ms", C.c_double)]


class OverlapParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("total_chunk", "mychunk", "bestn", "mc_lower", "mc_upper", "align_bandwidth", "ovlp_upper")]


class MapParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("total_chunk", "mychunk", "mc_lower", "mc_upper")]


class KVec(C.Structure):  # klib kvec {size_t n, m; T *a} (mm128_v, mp256_v)
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class PyMmer(C.Structure):  # py_mmer_t (src/shimmer.h:132-138)
    _fields_ = [("mmers", C.POINTER(KVec)), ("mmer0_map", C.c_void_p), ("rlmap", C.c_void_p), ("mcmap", C.c_void_p),
                ("ridmm", C.c_void_p)]


MP256_DTYPE = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("y0", "<u8"), ("y1", "<u8"), ("direction", "u1"), ("pad", "u1", 7)])


class OverlapStats(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_pair_records", C.c_uint64), ("n_buckets", C.c_uint64),
                ("n_align_needed", C.c_uint64), ("n_align_gpu", C.c_uint64), ("n_seen_skip", C.c_uint64),
                ("rounds", C.c_uint32), ("gpu_ms", C.c_double), ("host_ms", C.c_double),
                ("n_evaluations", C.c_uint64), ("device_replay", C.c_uint32), ("device_visit", C.c_uint32),
                ("replay_attempts", C.c_uint32), ("reserved0", C.c_uint32), ("stream_checksum", C.c_uint64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/pgx.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "pgx_init", "pgx_shutdown", "pgx_last_error", "pgx_device_count", "pgx_version", "pgx_free",
    "pgx_timing_get", "pgx_timing_reset", "pgx_mem_ledger", "pgx_results_async", "pgx_results_wait",
    "pgx_seqdb_upload", "pgx_seqdb_load", "pgx_seqdb_free", "pgx_seqdb_bases", "pgx_seqdb_reads", "pgx_seqdb_release_bytes", "pgx_seqdb_has_bytes",
    "pgx_index_resident", "pgx_index_result_free", "pgx_index_chunk",
    "pgx_overlap_resident", "pgx_overlap_chunk", "pgx_index_chunk_db", "pgx_overlap_chunk_db", "pgx_overlap_chunk_db_begin", "pgx_output_finish", "pgx_index_overlap_resident", "pgx_mkseqdb", "pgx_dedup",
    "pgx_sketch_batch", "pgx_reduce_batch", "pgx_count_batch", "pgx_align_batch",
    "decode_biseq", "encode_biseq", "mm_sketch", "mm_reduce", "ovlp_match", "free_ovlp_match", "read_mmlist", "write_mmlist",
    "pgx_map", "pgx_map_chunk", "pgx_khash_slot_order", "pgx_khash_slot_order_ex",
    "pgx_seqdb_upload_dev", "pgx_index_resident_dev", "pgx_pairs_prepare_dev", "pgx_pairs_scatter_dev", "pgx_overlap_records_dev",
    "pgx_overlap_resident_dev", "pgx_copy_dev", "pgx_seqdb_adopt_dev", "pgx_stream_wait", "pgx_stream_signal",
    "build_shimmer_map4py", "get_shimmers_for_read", "get_mmer_count", "get_shimmer_hits", "pgx_shimmer_map_free",
]

_lib = None


def load():
    """Load libpgx.so (building it is __graft_entry__.build()'s job).  Raises if absent: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PgxError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so.7; if libpgx were loaded first
        # it would bind /opt/rocm's copy, a later `import torch` would load the bundled one next to it, and the second
        # runtime finds no device.  With torch imported first both resolve to the same copy (the loader matches SONAMEs).
        # Without torch installed nothing changes.
        # (PGX_NO_TORCH=1: the caller promises that torch is never imported in this process -- the bin/ drop-ins do, and
        # save its ~2 s import)
        if os.environ.get("PGX_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        lib = C.CDLL(LIB_PATH)
        lib.pgx_last_error.restype = C.c_char_p
        lib.pgx_version.restype = C.c_char_p
        lib.pgx_free.argtypes = [C.c_void_p]
        lib.pgx_seqdb_free.argtypes = [C.c_void_p]
        lib.pgx_seqdb_bases.restype = C.c_uint64
        lib.pgx_seqdb_bases.argtypes = [C.c_void_p]
        lib.pgx_seqdb_reads.restype = C.c_uint32
        lib.pgx_seqdb_reads.argtypes = [C.c_void_p]
        lib.pgx_seqdb_release_bytes.argtypes = [C.c_void_p]
        lib.pgx_seqdb_has_bytes.argtypes = [C.c_void_p]
        lib.pgx_seqdb_upload.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pgx_seqdb_load.argtypes = [C.c_char_p, C.c_void_p]
        lib.pgx_index_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_index_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]
        lib.pgx_overlap_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_overlap_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]
        lib.pgx_index_overlap_resident.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        lib.pgx_seqdb_upload_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pgx_index_resident_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_pairs_prepare_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        lib.pgx_pairs_scatter_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        lib.pgx_overlap_records_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_overlap_resident_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_copy_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.pgx_seqdb_adopt_dev.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pgx_stream_wait.argtypes = [C.c_void_p]
        lib.pgx_stream_signal.argtypes = [C.c_void_p]
        lib.pgx_mkseqdb.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p]
        lib.pgx_dedup.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_sketch_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.pgx_reduce_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.pgx_count_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.pgx_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.pgx_timing_get.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_khash_slot_order.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        lib.pgx_khash_slot_order_ex.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.pgx_map.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pgx_map_chunk.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.build_shimmer_map4py.restype = None
        lib.build_shimmer_map4py.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.get_shimmers_for_read.restype = None
        lib.get_shimmers_for_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lib.get_mmer_count.restype = C.c_uint32
        lib.get_mmer_count.argtypes = [C.c_void_p, C.c_uint64]
        lib.get_shimmer_hits.restype = None
        lib.get_shimmer_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        lib.pgx_shimmer_map_free.restype = None
        lib.pgx_shimmer_map_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


PGX_ESTATE = -5   # include/pgx.h


def check(rc: int, what: str = "pgx"):
    if rc != 0:
        raise PgxError(f"{what} failed (code {rc}): {load().pgx_last_error().decode(errors='replace')}")


_inited = None


def init(device: int | None = None):
    """Select the GPU: the argument, else PGX_DEVICE, else LOCAL_RANK, else 0 -- the order the native drop-ins use
    (csrc/pgx_cli.c).  One context per process: asking for a different device later raises.  Raises when no GPU is visible."""
    global _inited
    if device is None:
        device = int(os.environ.get("PGX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    device = int(device)
    if _inited is None:
        check(load().pgx_init(device), "pgx_init")
        _inited = device
    elif _inited != device:
        raise PgxError(f"libpgx is one context per process and already runs on device {_inited}; device {device} was asked for")
    return device


def shutdown():
    """release the library's device state; the next init() may choose another device"""
    global _inited
    load().pgx_shutdown()
    _inited = None


class _DevMem:
    """device memory owned by libpgx, exposed through the CUDA array interface (zero-copy torch.as_tensor)"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr: int, nbytes: int, device):
    """a torch uint8 tensor over `nbytes` of library-owned device memory at `ptr`: a zero-copy view where torch accepts the
    CUDA array interface, else a device-to-device copy.  The view is valid as long as the library keeps the memory."""
    import torch
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    try:
        t = torch.as_tensor(_DevMem(ptr, nbytes), device=device)
        if t.data_ptr() == ptr and t.numel() == nbytes:
            return t
    except Exception:
        pass
    t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    check(load().pgx_copy_dev(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), nbytes), "pgx_copy_dev")
    return t


def stream_wait(torch_stream=None):
    """the library's stream waits (on the device, the host does not) for what has been enqueued on `torch_stream` so far --
    e.g. the collective that produced a buffer the next library call reads.  None: torch's current stream."""
    import torch
    s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
    check(load().pgx_stream_wait(C.c_void_p(s.cuda_stream)), "pgx_stream_wait")


def stream_signal(torch_stream=None):
    """`torch_stream` (None: torch's current stream) waits for what the library has enqueued so far"""
    import torch
    s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
    check(load().pgx_stream_signal(C.c_void_p(s.cuda_stream)), "pgx_stream_signal")


def take(ptr, n: int, dtype: np.dtype) -> np.ndarray:
    """Wrap a library-owned host array as a numpy array WITHOUT copying; pgx_free runs when the array is collected."""
    import weakref
    n = int(n)
    if not ptr:
        return np.zeros(0, dtype)
    if n == 0:
        load().pgx_free(C.c_void_p(ptr))
        return np.zeros(0, dtype)
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr)
    out = np.frombuffer(buf, dtype=dtype)
    weakref.finalize(buf, load().pgx_free, C.c_void_p(ptr))
    return out


def results_async(on: bool) -> bool:
    """overlap records on their way to the host while the next chunk already runs (include/pgx.h: pgx_results_async); returns the
    previous setting.  With it on, call results_wait() before reading a returned record array."""
    return bool(load().pgx_results_async(1 if on else 0))


def results_wait():
    check(load().pgx_results_wait(), "pgx_results_wait")


def timing(name: str):
    ms, launches, units = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    load().pgx_timing_get(name.encode(), C.byref(ms), C.byref(launches), C.byref(units))
    return float(ms.value), int(launches.value), int(units.value)


def timing_reset():
    load().pgx_timing_reset()


def mem_ledger(reset_peak: bool = False) -> dict:
    """the library's device memory by owner at its peak (pgx_mem_ledger)"""
    import json
    lib = load()
    lib.pgx_mem_ledger.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    n = lib.pgx_mem_ledger(None, 0, 0)
    buf = C.create_string_buffer(n + 64)
    lib.pgx_mem_ledger(buf, n + 64, 1 if reset_peak else 0)
    return json.loads(buf.value.decode())
