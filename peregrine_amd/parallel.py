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


def chunk_of_rank(rank: int, world: int) -> int:
    """Rank r runs index chunk r+1 and overlap chunk r+1 of `world` (chunks are 1-based in the reference CLIs)."""
    return rank + 1


def reads_of_chunk(rid, chunk: int, total: int):
    """Read ownership of an index chunk: rid % total == chunk % total (/root/reference/src/shmr_index.c:157)."""
    import numpy as np
    rid = np.asarray(rid)
    return rid[rid % total == chunk % total]
