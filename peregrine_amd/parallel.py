"""Multi-GPU form of the chunked workflow (one process per GPU, torch.distributed; backend "nccl" IS RCCL on ROCm).

The reference couples its chunks only through files: every overlap chunk globs and reads ALL index chunks' shimmer and
count files (/root/reference/src/shmr_overlap.c:359-384), rebuilds the whole pair map and keeps the records whose first
key it owns, `(x >> 8) % N == c % N` (/root/reference/src/shmr_utils.c:337,362).  Here rank r runs index chunk r+1 and
overlap chunk r+1, and the coupling is the path's one exchange step, on device buffers over xGMI:

  1. count tables  : all-gather of every chunk's (hash, count) table, EXACT sizes, every piece written straight to its place in
                     the concatenation; each rank aggregates them (aggregate_mm_count, shmr_utils.c:162-176) for the
                     multiplicity filter of build_map
  2. scan start    : the global scan of build_map starts at the first shimmer with lower <= count < upper of the concatenated
                     list (shmr_utils.c:311-320): ranks before the one that holds it contribute nothing, that rank starts there,
                     later ranks start at 0.  Nearly always rank 0 holds it, so every rank builds its records on that
                     assumption (rank 0 from its own first such element, the others from 0) and ONE all-gather of
                     [first, records per destination] per rank both confirms it and tells every rank the whole send matrix;
                     a rank whose assumption was wrong rebuilds and the counts are gathered once more.
  3. pair records  : every rank routes the forward / reverse records of ITS reads' adjacent kept shimmers to the owner chunk
                     of their first key: one all-to-all(v) of 32-byte records (pgx_pair_rec), exact sizes.  The receiver
                     sees them in source-rank order, scan order inside a source = the insertion order build_map has over
                     the concatenated lists, which the result order depends on.

Per step: two integer all-gathers + the two payload collectives.  Library and torch / RCCL streams are ordered with events
(pgx_stream_wait), the host is only stopped where it needs a value.

The protocol is written against an `engine` (the stages); `GpuEngine` calls libpgx on device pointers.  The tests drive the
same functions with a numpy engine under gloo (tests/test_parallel_gloo.py), with GpuEngine under gloo on one GPU, and with
GpuEngine over RCCL as a one-rank job (PGX_FORCE_EXCHANGE=1: every collective is really issued).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 32  # sizeof(pgx_pair_rec)


def forced() -> bool:
    """PGX_FORCE_EXCHANGE=1: take the multi-rank code path -- process group, collectives on device views -- also in a
    one-rank job (RCCL accepts a one-rank communicator): the way to execute the RCCL path on a single GPU"""
    return os.environ.get("PGX_FORCE_EXCHANGE") == "1" and dist.is_available() and dist.is_initialized()


def _world(world):
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return world


def _collective(world: int) -> bool:
    return world > 1 or forced()


def _comm_device(t: torch.Tensor) -> torch.device:
    """tensors travel on their own device with RCCL; the gloo debug backend needs host tensors"""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return t.device


def allgather_ints(vals, world: int | None = None, device=None) -> list[list[int]]:
    """all-gather a few integers per rank (one collective, one host read-back); returns out[r] = rank r's list"""
    world = _world(world)
    if not _collective(world):
        return [list(map(int, vals))]
    cdev = torch.device("cpu") if dist.get_backend() == "gloo" else (device or torch.device("cuda", torch.cuda.current_device()))
    mine = torch.tensor(list(vals), dtype=torch.int64, device=cdev)
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(out, mine)
    return [[int(v) for v in row] for row in out.view(world, -1).tolist()]


def allgather_exact(t: torch.Tensor, sizes: list[int], out: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather 1-D byte tensors of KNOWN rank-dependent lengths: rank r's piece is received at its final offset of one
    buffer (`out`, or a new tensor of sum(sizes) bytes) -- nothing is padded to the longest piece and nothing is compacted
    afterwards.  RCCL: one grouped collective over views of the buffer (c10d turns an all_gather over unequal views into
    grouped broadcasts); gloo (host tensors, debugging / CPU tests): a broadcast per source."""
    rank, world = dist.get_rank(), len(sizes)
    assert t.dtype == torch.uint8 and t.dim() == 1 and t.numel() == sizes[rank]
    total = int(sum(sizes))
    home = t.device
    cdev = _comm_device(t)
    if out is None:
        out = torch.empty(total, dtype=torch.uint8, device=cdev)
    else:
        assert out.device == cdev and out.numel() >= total
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    views = [out[int(offs[r]):int(offs[r + 1])] for r in range(world)]
    if dist.get_backend() == "gloo":
        views[rank].copy_(t.to(cdev))
        for r in range(world):
            if sizes[r]:
                dist.broadcast(views[r], src=r)
    elif len(set(sizes)) == 1 and os.environ.get("PGX_FORCE_UNEVEN") != "1":   # (the knob lets a one-rank test take the branch below)
        dist.all_gather_into_tensor(out[:total], t)
    else:
        dist.all_gather(views, t)
    return out if out.device == home else out.to(home)


def allgather_cat(t: torch.Tensor, world: int | None = None) -> tuple[torch.Tensor, list[int]]:
    """All-gather 1-D byte tensors of rank-dependent lengths: (concatenation in rank order, per-rank lengths).  One integer
    all-gather for the lengths, then allgather_exact."""
    world = _world(world)
    assert t.dtype == torch.uint8 and t.dim() == 1
    if not _collective(world):
        return t, [t.numel()]
    sizes = [s[0] for s in allgather_ints([t.numel()], world, device=t.device if t.device.type == "cuda" else None)]
    return allgather_exact(t, sizes), sizes


def alltoallv_bytes(send: torch.Tensor, send_bytes: list[int], world: int | None = None,
                    recv_bytes: list[int] | None = None) -> tuple[torch.Tensor, list[int]]:
    """all-to-all(v) of a byte tensor laid out destination-major; returns (received bytes source-major, bytes per source).
    recv_bytes: what every source sends here, when the caller already knows it (exchange_overlap does); else one more small
    collective asks for it."""
    world = _world(world)
    assert send.dtype == torch.uint8 and send.dim() == 1 and len(send_bytes) == world and sum(send_bytes) == send.numel()
    if not _collective(world):
        return send, list(send_bytes)
    home = send.device
    cdev = _comm_device(send)
    if recv_bytes is None:
        mine = torch.tensor(list(send_bytes), dtype=torch.int64, device=cdev)
        got = torch.empty(world, dtype=torch.int64, device=cdev)
        dist.all_to_all_single(got, mine)
        recv_bytes = [int(v) for v in got.tolist()]
    src = send.to(cdev)
    out = torch.empty(sum(recv_bytes), dtype=torch.uint8, device=cdev)
    dist.all_to_all_single(out, src, output_split_sizes=list(recv_bytes), input_split_sizes=list(send_bytes))
    return out.to(home), list(recv_bytes)


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
    idev = top.device if top.device.type == "cuda" else None
    counts_all, _ = allgather_cat(mc, world)                                        # (1): sizes + payload
    first = engine.pairs_prepare(top, counts_all, params.get("mc_lower", 2), params.get("mc_upper", 240))
    guess = first if rank == 0 else 0                                               # (2): rank 0 holds the scan start (nearly always)
    send, counts = engine.pairs_scatter(world, guess)
    table = allgather_ints([first] + [int(c) for c in counts], world, device=idev)
    firsts = [row[0] for row in table]
    matrix = [row[1:] for row in table]
    redo = [scan_start(firsts, r) != (firsts[0] if r == 0 else 0) for r in range(world)]
    if any(redo):                                                                   # some rank before the first holder assumed wrongly
        if redo[rank]:
            send, counts = engine.pairs_scatter(world, scan_start(firsts, rank))
        matrix = allgather_ints([int(c) for c in counts], world, device=idev)
    recv_bytes = [int(matrix[s][rank]) * REC_BYTES for s in range(world)]
    recv, recv_bytes = alltoallv_bytes(send, [int(c) * REC_BYTES for c in counts], world, recv_bytes)   # (3)
    info = {"sent_records": int(sum(counts)), "received_records": recv.numel() // REC_BYTES,
            "received_per_source": [b // REC_BYTES for b in recv_bytes], "count_entries_all": counts_all.numel() // 16,
            "scan_start_redone": bool(any(redo))}
    return engine.overlap_records(recv, world, rank + 1, **params), info


def gather_seqdb(mine_seq: np.ndarray, mine_rlen: np.ndarray, world: int, device: torch.device):
    """Setup of a multi-rank job: the job's read set = the union of the ranks' sets in rank order, replicated in every GPU's
    HBM.  Every rank's bytes are received straight into their final place in ONE buffer of total + 1024 bytes
    that the library then adopts without a copy (pgx_seqdb_adopt_dev): one copy of the seqdb per GPU.
    Returns (device buffer, total bytes, rlen of all reads)."""
    nb = allgather_ints([int(mine_seq.size), int(mine_rlen.size)], world, device=device)
    seq_sizes, len_sizes = [r[0] for r in nb], [r[1] * 4 for r in nb]
    total = int(sum(seq_sizes))
    gloo = dist.get_backend() == "gloo"
    buf = torch.empty(total + 1024, dtype=torch.uint8, device=torch.device("cpu") if gloo else device)
    piece = torch.from_numpy(mine_seq)
    allgather_exact(piece if gloo else piece.to(device), seq_sizes, out=buf)
    if gloo:
        buf = buf.to(device)
    lens = torch.from_numpy(np.ascontiguousarray(mine_rlen, np.uint32).view(np.uint8))
    len_all = allgather_exact(lens if gloo else lens.to(device), len_sizes)
    rlen = len_all.cpu().numpy().view(np.uint32).copy()
    return buf, total, rlen


class GpuEngine:
    """the stages on the GPU through libpgx's device-pointer entry points (include/pgx.h: pgx_*_dev)"""

    def __init__(self, rdb, device: torch.device):
        self.rdb = rdb
        self.device = device

    def index(self, world: int, chunk: int, levels: int = 2):
        """returns (IndexOut, list bytes, count-table bytes): zero-copy views of library-owned device memory, valid until the
        next index call (the library refuses a scatter whose prepared list was rewritten in between, PGX_ESTATE)"""
        from . import _lib
        ix, d_top, n_top, d_mc, n_mc = self.rdb.index_dev(total_chunk=world, mychunk=chunk, levels=levels)
        return ix, _lib.dev_tensor(d_top, n_top * 16, self.device), _lib.dev_tensor(d_mc, n_mc * 16, self.device)

    def pairs_prepare(self, top: torch.Tensor, counts_all: torch.Tensor, lower: int, upper: int) -> int:
        from . import _lib
        _lib.stream_wait()                  # the collective that produced counts_all was enqueued on torch's stream: an event, no host stop
        self._keep = (top, counts_all)      # the library reads them again in pairs_scatter
        return self.rdb.pairs_prepare_dev(top.data_ptr(), top.numel() // 16, counts_all.data_ptr(), counts_all.numel() // 16,
                                          lower, upper)

    def pairs_scatter(self, world: int, start: int):
        from . import _lib
        d_send, counts = self.rdb.pairs_scatter_dev(world, start)   # (returns after the library's stream has produced the counts)
        n = int(counts.sum())
        return _lib.dev_tensor(d_send, n * REC_BYTES, self.device), [int(c) for c in counts]

    def overlap_records(self, recv: torch.Tensor, world: int, chunk: int, **params):
        from . import _lib
        _lib.stream_wait()
        params = {k: v for k, v in params.items() if k in ("bestn", "mc_lower", "mc_upper", "align_bandwidth", "ovlp_upper")}
        return self.rdb.overlap_records_dev(recv.data_ptr(), recv.numel() // REC_BYTES, total_chunk=world, mychunk=chunk, **params)


def reads_of_chunk(rid, chunk: int, total: int):
    """Read ownership of an index chunk: rid % total == chunk % total (/root/reference/src/shmr_index.c:157)."""
    rid = np.asarray(rid)
    return rid[rid % total == chunk % total]
