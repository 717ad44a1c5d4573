"""Multi-GPU form of the chunked workflow (one process per GPU, torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference couples its chunks only through files: every overlap chunk globs and reads ALL index chunks' shimmer and
count files (/root/reference/src/shmr_overlap.c:359-384), rebuilds the whole pair map and keeps the records whose first
key it owns, `(x >> 8) % N == c % N` (/root/reference/src/shmr_utils.c:337,362).  Here rank r runs index chunk r+1 and
overlap chunk r+1, and the coupling is the path's one exchange step, on device buffers over xGMI:

  1. count tables  : all-gather of every chunk's (hash, count) table, exact sizes; each rank aggregates them
                     (aggregate_mm_count, shmr_utils.c:162-176) for the multiplicity filter of build_map
  2. scan start    : all-gather of one integer per rank -- the position of the first shimmer with
                     lower <= count < upper in its list; the global scan of build_map starts at the first such element
                     of the concatenated list (shmr_utils.c:311-320)
  3. pair records  : every rank builds the forward / reverse records of ITS reads' adjacent kept shimmers and routes each
                     to the owner chunk of its first key: one all-to-all(v) of 32-byte records (pgx_pair_rec).  The
                     receiver sees them in source-rank order, scan order inside a source = the insertion order
                     build_map has over the concatenated lists, which the result order depends on.

The protocol is written against an `engine` (the stages); `GpuEngine` calls libpgx on device pointers.  The tests drive the
same functions with a numpy engine under gloo (tests/test_parallel_gloo.py) and with GpuEngine under gloo on one GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 32  # sizeof(pgx_pair_rec)


def _world(world):
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return world


def _comm_device(t: torch.Tensor) -> torch.device:
    """tensors travel on their own device with RCCL; the gloo debug backend needs host tensors"""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return t.device


def allgather_ints(vals, world: int | None = None, device=None) -> list[list[int]]:
    """all-gather a few integers per rank; returns out[r] = rank r's list"""
    world = _world(world)
    if world == 1:
        return [list(map(int, vals))]
    cdev = torch.device("cpu") if dist.get_backend() == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    mine = torch.tensor(list(vals), dtype=torch.int64, device=cdev)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(out, mine)
    return [[int(v) for v in row] for row in out.view(world, -1).tolist()]


def allgather_cat(t: torch.Tensor, world: int | None = None) -> tuple[torch.Tensor, list[int]]:
    """All-gather 1-D byte tensors of rank-dependent lengths and return (concatenation in rank order, per-rank lengths).
    Exact sizes on the wire are not possible with a single all-gather, so the payload is padded to the longest piece and
    compacted on the receiving device; no host hop with RCCL."""
    world = _world(world)
    assert t.dtype == torch.uint8 and t.dim() == 1
    if world == 1:
        return t, [t.numel()]
    home = t.device
    cdev = _comm_device(t)
    sizes = [s[0] for s in allgather_ints([t.numel()], world, device=home if home.type == "cuda" else None)]
    cap = max(max(sizes), 1)
    pad = torch.zeros(cap, dtype=torch.uint8, device=cdev)
    pad[: t.numel()] = t.to(cdev)
    buf = torch.empty(world * cap, dtype=torch.uint8, device=cdev)
    dist.all_gather_into_tensor(buf, pad)
    rows = buf.view(world, cap)
    cat = torch.cat([rows[r, : sizes[r]] for r in range(world)]) if sum(sizes) else torch.empty(0, dtype=torch.uint8, device=cdev)
    return cat.to(home), sizes


def alltoallv_bytes(send: torch.Tensor, send_bytes: list[int], world: int | None = None) -> tuple[torch.Tensor, list[int]]:
    """all-to-all(v) of a byte tensor laid out destination-major; returns (received bytes source-major, bytes per source)"""
    world = _world(world)
    assert send.dtype == torch.uint8 and send.dim() == 1 and len(send_bytes) == world and sum(send_bytes) == send.numel()
    if world == 1:
        return send, list(send_bytes)
    home = send.device
    cdev = _comm_device(send)
    recv_bytes = [row[0] for row in _alltoall_ints(send_bytes, world, home)]
    src = send.to(cdev)
    out = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=cdev)
    dist.all_to_all_single(out, src, output_split_sizes=recv_bytes, input_split_sizes=list(send_bytes))
    return out.to(home), recv_bytes


def _alltoall_ints(vals, world, home):
    """rank r sends vals[d] to rank d; returns [[v_from_rank0], [v_from_rank1], ...]"""
    cdev = torch.device("cpu") if dist.get_backend() == "gloo" else home
    mine = torch.tensor(list(vals), dtype=torch.int64, device=cdev)
    out = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(out, mine)
    return [[int(v)] for v in out.tolist()]


def scan_start(firsts: list[int], rank: int) -> int:
    """The list position rank `rank` starts its part of build_map's scan at, from every rank's first-strict index (-1: none).
    The scan of the concatenated list starts at the first element with lower <= count < upper (shmr_utils.c:311-320): ranks
    before the one that holds it contribute nothing (-1), that rank starts there, later ranks start at 0."""
    g = next((r for r, f in enumerate(firsts) if f >= 0), None)
    if g is None or rank < g:
        return -1
    return firsts[rank] if rank == g else 0


def exchange_overlap(engine, rank: int, world: int, top: torch.Tensor, mc: torch.Tensor, **params):
    """The exchange step + the overlap stage of chunk rank+1 of `world`.  top / mc: this rank's final-level list and count
    table as byte tensors (device tensors with GpuEngine).  Returns what engine.overlap_records returns, plus a dict of sizes."""
    counts_all, _ = allgather_cat(mc, world)                                        # (1)
    first = engine.pairs_prepare(top, counts_all, params.get("mc_lower", 2), params.get("mc_upper", 240))
    firsts = [f[0] for f in allgather_ints([first], world, device=top.device if top.device.type == "cuda" else None)]   # (2)
    send, counts = engine.pairs_scatter(world, scan_start(firsts, rank))
    recv, recv_bytes = alltoallv_bytes(send, [int(c) * REC_BYTES for c in counts], world)   # (3)
    info = {"sent_records": int(sum(counts)), "received_records": recv.numel() // REC_BYTES,
            "received_per_source": [b // REC_BYTES for b in recv_bytes], "count_entries_all": counts_all.numel() // 16}
    return engine.overlap_records(recv, world, rank + 1, **params), info


class GpuEngine:
    """the stages on the GPU through libpgx's device-pointer entry points (include/pgx.h: pgx_*_dev)"""

    def __init__(self, rdb, device: torch.device):
        self.rdb = rdb
        self.device = device

    def index(self, world: int, chunk: int, levels: int = 2):
        """returns (IndexOut, list bytes, count-table bytes): zero-copy views of library-owned device memory, valid until the
        next index call"""
        from . import _lib
        ix, d_top, n_top, d_mc, n_mc = self.rdb.index_dev(total_chunk=world, mychunk=chunk, levels=levels)
        return ix, _lib.dev_tensor(d_top, n_top * 16, self.device), _lib.dev_tensor(d_mc, n_mc * 16, self.device)

    def pairs_prepare(self, top: torch.Tensor, counts_all: torch.Tensor, lower: int, upper: int) -> int:
        torch.cuda.current_stream(self.device).synchronize()   # the collective that produced counts_all ran on torch's stream
        self._keep = (top, counts_all)                         # the library reads them again in pairs_scatter
        return self.rdb.pairs_prepare_dev(top.data_ptr(), top.numel() // 16, counts_all.data_ptr(), counts_all.numel() // 16,
                                          lower, upper)

    def pairs_scatter(self, world: int, start: int):
        from . import _lib
        d_send, counts = self.rdb.pairs_scatter_dev(world, start)
        n = int(counts.sum())
        return _lib.dev_tensor(d_send, n * REC_BYTES, self.device), [int(c) for c in counts]

    def overlap_records(self, recv: torch.Tensor, world: int, chunk: int, **params):
        torch.cuda.current_stream(self.device).synchronize()
        params = {k: v for k, v in params.items() if k in ("bestn", "mc_lower", "mc_upper", "align_bandwidth", "ovlp_upper")}
        return self.rdb.overlap_records_dev(recv.data_ptr(), recv.numel() // REC_BYTES, total_chunk=world, mychunk=chunk, **params)


# ---- kept from round 1 (used by the setup of a multi-rank job and by tests) -------------------------------------------------
def allgather_records(t: torch.Tensor, world: int | None = None) -> list[torch.Tensor]:
    """All-gather 1-D tensors of different lengths; returns the per-rank pieces in rank (= chunk) order."""
    world = _world(world)
    if world == 1:
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    pad = torch.zeros(cap, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    bufs = [torch.empty(cap, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:s] for b, s in zip(bufs, sizes)]


def allgather_many(ts: list[torch.Tensor], world: int | None = None) -> list[list[torch.Tensor]]:
    """All-gather SEVERAL 1-D byte tensors of rank-dependent lengths with two collectives in total (one for the lengths,
    one for the concatenated, padded payload).  Returns out[i][r] = rank r's i-th tensor."""
    world = _world(world)
    if world == 1:
        return [[t] for t in ts]
    assert all(t.dtype == torch.uint8 and t.dim() == 1 for t in ts)
    dev = ts[0].device
    n = torch.tensor([t.numel() for t in ts], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [[int(v) for v in s.tolist()] for s in sizes]
    cap = max(max(sum(s) for s in sizes), 1)
    pad = torch.zeros(cap, dtype=torch.uint8, device=dev)
    pad[: int(n.sum().item())] = torch.cat(ts) if len(ts) > 1 else ts[0]
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = [[] for _ in ts]
    for r in range(world):
        o = 0
        for i, sz in enumerate(sizes[r]):
            out[i].append(bufs[r][o:o + sz])
            o += sz
    return out


def chunk_of_rank(rank: int, world: int) -> int:
    """Rank r runs index chunk r+1 and overlap chunk r+1 of `world` (chunks are 1-based in the reference CLIs)."""
    return rank + 1


def reads_of_chunk(rid, chunk: int, total: int):
    """Read ownership of an index chunk: rid % total == chunk % total (/root/reference/src/shmr_index.c:157)."""
    rid = np.asarray(rid)
    return rid[rid % total == chunk % total]
