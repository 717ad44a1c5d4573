"""Multi-GPU exchange for the chunked workflow (one process per GPU, torch.distributed; backend "nccl" is RCCL).

The reference couples chunks only through files: every overlap chunk globs and reads ALL index chunks' shimmer and MC
files (/root/reference/src/shmr_overlap.c:355-384).  Here that step is one variable-length all-gather over xGMI:
rank r contributes the list of index chunk r+1 and every rank receives the concatenation in chunk order (the order the
reference's name-sorted glob yields), which is the insertion order build_map depends on.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def allgather_records(t: torch.Tensor, world: int | None = None) -> list[torch.Tensor]:
    """All-gather 1-D tensors of different lengths; returns the per-rank pieces in rank (= chunk) order."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
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
    dist.all_gather(bufs, pad)   # one bucketed collective; sizes are a few MB..GB per rank
    return [b[:s] for b, s in zip(bufs, sizes)]


def allgather_many(ts: list[torch.Tensor], world: int | None = None) -> list[list[torch.Tensor]]:
    """All-gather SEVERAL 1-D byte tensors of rank-dependent lengths with two collectives in total (one for the lengths,
    one for the concatenated, padded payload) instead of two per tensor.  Returns out[i][r] = rank r's i-th tensor."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
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
    import numpy as np
    rid = np.asarray(rid)
    return rid[rid % total == chunk % total]
