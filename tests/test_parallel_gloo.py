"""Multi-rank exchange on CPU (gloo, world_size 2): the variable-length all-gather that replaces the reference's
"every overlap chunk globs every index chunk's files" step (src/shmr_overlap.c:355-384) must hand every rank the
chunks' lists concatenated in chunk (= rank) order -- the insertion order build_map depends on."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peregrine_amd.formats import MM_DTYPE
    from peregrine_amd.parallel import allgather_many, allgather_records, chunk_of_rank
    rng = np.random.default_rng(100 + rank)
    n = 1000 + 37 * rank                        # ranks contribute different lengths (rank 1 may even be empty below)
    mine = np.zeros(n if rank != 1 or world < 3 else 0, MM_DTYPE)
    mine["x"] = rng.integers(0, 2**40, len(mine), dtype=np.uint64)
    mine["y"] = (np.uint64(rank) << np.uint64(32)) | np.arange(len(mine), dtype=np.uint64)
    parts = allgather_records(torch.from_numpy(mine.view(np.uint8).copy()), world)
    got = np.concatenate([p.numpy().view(MM_DTYPE) for p in parts])
    np.save(os.path.join(out_dir, f"got{rank}.npy"), got)
    np.save(os.path.join(out_dir, f"mine{rank}.npy"), mine)
    # the two-collective form used between the stages: shimmer list + a second payload of another length per rank
    extra = rng.integers(0, 256, 16 * (5 - rank), dtype=np.uint8)
    two = allgather_many([torch.from_numpy(mine.view(np.uint8).copy()), torch.from_numpy(extra.copy())], world)
    assert np.array_equal(np.concatenate([p.numpy().view(MM_DTYPE) for p in two[0]]), got)
    np.save(os.path.join(out_dir, f"extra{rank}.npy"), extra)
    np.save(os.path.join(out_dir, f"gotx{rank}.npy"), np.concatenate([p.numpy() for p in two[1]]))
    assert chunk_of_rank(rank, world) == rank + 1
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_allgather_records_orders_by_chunk(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = np.concatenate([np.load(tmp_path / f"mine{r}.npy") for r in range(world)])
    for r in range(world):
        got = np.load(tmp_path / f"got{r}.npy")
        assert np.array_equal(got, want), f"rank {r}"
        assert np.array_equal(np.load(tmp_path / f"gotx{r}.npy"), np.concatenate([np.load(tmp_path / f"extra{q}.npy") for q in range(world)]))
    # chunk ownership of reads follows the reference: rid % N == chunk % N with 1-based chunks (shmr_index.c:157)
    from peregrine_amd.parallel import reads_of_chunk
    rid = np.arange(20, dtype=np.uint32)
    owned = [reads_of_chunk(rid, c, world) for c in range(1, world + 1)]
    assert sorted(np.concatenate(owned).tolist()) == rid.tolist()
    assert reads_of_chunk(rid, world, world).tolist() == [r for r in range(20) if r % world == 0]
