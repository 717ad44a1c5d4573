"""On-disk formats of the SHIMMER index/overlap path (numpy views, no compute).

Formats restated from the reference (never copied):
  * seqdb / idx          -- /root/reference/src/shmr_mkseqdb.c:99-121, shmr_utils.c:44-62
  * mmlist  (L0/L1/L2)   -- shmr_utils.c:98-123   : u64 n, then n x {u64 x, u64 y}
  * mm_count (MC)        -- shmr_utils.c:178-203  : u64 n, then n x {u64 mer, u32 count, 4 B padding}
  * ovlp_t stream        -- shimmer.h:104-110, shmr_overlap.c:163-173 : headerless 64-byte records
"""
from __future__ import annotations

import glob as _glob
import os
from dataclasses import dataclass

import numpy as np

MM_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8")])
MC_DTYPE = np.dtype([("mer", "<u8"), ("count", "<u4"), ("pad", "<u4")])
MATCH_FIELDS = ("m_size", "dist", "q_bgn", "q_end", "t_bgn", "t_end", "t_m_end", "q_m_end")
OVLP_DTYPE = np.dtype(
    [("y0", "<u8"), ("y1", "<u8"), ("rl0", "<u4"), ("rl1", "<u4"),
     ("strand0", "u1"), ("strand1", "u1"), ("ovlp_type", "u1"), ("pad0", "u1")]
    + [(f, "<i4") for f in MATCH_FIELDS]
    + [("pad1", "<u4")]
)
assert MM_DTYPE.itemsize == 16 and MC_DTYPE.itemsize == 16 and OVLP_DTYPE.itemsize == 64

OVLP_FIELDS = tuple(n for n in OVLP_DTYPE.names if not n.startswith("pad"))


@dataclass
class SeqDB:
    """A read database: `seqdb` is the concatenated 1-byte/base two-strand encoding."""
    seqdb: np.ndarray      # uint8
    rid: np.ndarray        # uint32, idx-file order
    rlen: np.ndarray       # uint32
    roff: np.ndarray       # uint64 byte offset into seqdb
    names: list | None = None

    @property
    def n_reads(self) -> int:
        return int(self.rid.shape[0])

    @property
    def n_bases(self) -> int:
        return int(self.rlen.sum(dtype=np.uint64))

    def by_rid(self):
        """(rlen, roff) arrays indexed by rid (size max_rid+1)."""
        m = int(self.rid.max()) + 1 if self.n_reads else 0
        rl = np.zeros(m, np.uint32)
        ro = np.zeros(m, np.uint64)
        rl[self.rid] = self.rlen
        ro[self.rid] = self.roff
        return rl, ro


def write_seqdb(prefix: str, db: SeqDB) -> None:
    db.seqdb.tofile(prefix + ".seqdb")
    names = db.names or [f"r{int(r):09d}" for r in db.rid]
    with open(prefix + ".idx", "w") as f:
        for r, nm, ln, off in zip(db.rid, names, db.rlen, db.roff):
            f.write("%09d %s %u %u\n" % (int(r), nm, int(ln), int(off)))


def read_idx(path: str):
    rid, names, rlen, roff = [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) < 4:
                continue
            rid.append(int(p[0])); names.append(p[1]); rlen.append(int(p[2])); roff.append(int(p[3]))
    return (np.asarray(rid, np.uint32), names, np.asarray(rlen, np.uint32), np.asarray(roff, np.uint64))


def read_seqdb(prefix: str, mmap: bool = True) -> SeqDB:
    rid, names, rlen, roff = read_idx(prefix + ".idx")
    path = prefix + ".seqdb"
    if os.path.getsize(path) == 0:
        data = np.zeros(0, np.uint8)
    else:
        data = np.memmap(path, dtype=np.uint8, mode="r") if mmap else np.fromfile(path, dtype=np.uint8)
    return SeqDB(data, rid, rlen, roff, names)


def _read_counted(path: str, dtype: np.dtype) -> np.ndarray:
    with open(path, "rb") as f:
        n = int(np.frombuffer(f.read(8), "<u8")[0])
        return np.fromfile(f, dtype=dtype, count=n)


def _write_counted(path: str, arr: np.ndarray, dtype: np.dtype) -> None:
    arr = np.ascontiguousarray(arr, dtype=dtype)
    with open(path, "wb") as f:
        f.write(np.uint64(arr.shape[0]).tobytes())
        arr.tofile(f)


def read_mmlist(path: str) -> np.ndarray:
    return _read_counted(path, MM_DTYPE)


def write_mmlist(path: str, arr: np.ndarray) -> None:
    _write_counted(path, arr, MM_DTYPE)


def read_mm_count(path: str) -> np.ndarray:
    return _read_counted(path, MC_DTYPE)


def write_mm_count(path: str, arr: np.ndarray) -> None:
    _write_counted(path, arr, MC_DTYPE)


def read_ovlp(path: str) -> np.ndarray:
    return np.fromfile(path, dtype=OVLP_DTYPE)


def ovlp_fields_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """Field-exact, order-exact comparison that masks the padding bytes 27 and 60..63."""
    if a.shape != b.shape:
        return False
    return all(np.array_equal(a[f], b[f]) for f in OVLP_FIELDS)


def _mix64(h: np.ndarray) -> np.ndarray:
    h = h ^ (h >> np.uint64(33))
    h = h * np.uint64(0xff51afd7ed558ccd)
    h = h ^ (h >> np.uint64(33))
    h = h * np.uint64(0xc4ceb9fe1a85ec53)
    return h ^ (h >> np.uint64(33))


def stream_checksum(ov: np.ndarray, block: int = 1 << 22) -> int:
    """pgx_overlap_stats.stream_checksum (pgx_internal.h::record_checksum) in numpy: the sum over the records of a 64-bit mix of every
    field -- padding bytes excluded -- and the record's position in the stream (order-sensitive; arithmetic modulo 2^64)."""
    total = np.uint64(0)
    u = lambda a: a.astype(np.uint32).astype(np.uint64)   # noqa: E731  (two's complement of the int32 fields)
    with np.errstate(over="ignore"):
        for s in range(0, len(ov), block):
            o = ov[s:s + block]
            pos = np.arange(s, s + len(o), dtype=np.uint64)
            h = _mix64(o["y0"] + np.uint64(0x9E3779B97F4A7C15) * (pos + np.uint64(1)))
            h = _mix64(h ^ o["y1"])
            h = _mix64(h ^ (o["rl0"].astype(np.uint64) | (o["rl1"].astype(np.uint64) << np.uint64(32))))
            h = _mix64(h ^ (o["strand0"].astype(np.uint64) | (o["strand1"].astype(np.uint64) << np.uint64(8)) | (o["ovlp_type"].astype(np.uint64) << np.uint64(16))))
            for a, b in (("m_size", "dist"), ("q_bgn", "q_end"), ("t_bgn", "t_end"), ("t_m_end", "q_m_end")):
                h = _mix64(h ^ (u(o[a]) | (u(o[b]) << np.uint64(32))))
            total = total + h.sum(dtype=np.uint64)
    return int(total)


def masked_stream_sha256(ov, block: int = 1 << 20) -> str:
    """SHA-256 of an ovlp_t stream with the padding bytes (27 and 60..63 of every 64-byte record: the reference writes whatever its
    stack held there) zeroed.  ov: a record array or a file path.  `cat stream | zero those bytes | sha256sum`."""
    import hashlib
    h = hashlib.sha256()
    if isinstance(ov, (str, bytes)):
        ov = np.memmap(ov, dtype=np.uint8, mode="r")
    raw = np.asarray(ov).view(np.uint8).reshape(-1, 64)
    for s in range(0, len(raw), block):
        b = np.array(raw[s:s + block], copy=True)
        b[:, 27] = 0
        b[:, 60:64] = 0
        h.update(b.tobytes() if not b.flags.c_contiguous else memoryview(b.reshape(-1)))
    return h.hexdigest()


def mc_as_sorted_pairs(arr: np.ndarray) -> np.ndarray:
    """MC parity is the multiset of (mer, count) -- slot order and padding carry no meaning."""
    out = np.stack([arr["mer"].astype(np.uint64), arr["count"].astype(np.uint64)], axis=1)
    return out[np.lexsort((out[:, 1], out[:, 0]))]


def shimmer_files(shimmer_prefix: str):
    """Name-sorted index-chunk files, as the reference's wordexp globs (shmr_overlap.c:355-384)."""
    mm = sorted(_glob.glob(_glob.escape(shimmer_prefix) + "-[0-9]*-of-[0-9]*.dat"))
    mc = sorted(_glob.glob(_glob.escape(shimmer_prefix) + "-MC-[0-9]*-of-[0-9]*.dat"))
    return mm, mc
