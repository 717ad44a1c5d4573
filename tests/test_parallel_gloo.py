"""Multi-rank exchange on CPU (gloo, world_size 2, 3 and 8 -- the one-chunk-per-rank form an 8-GPU node runs).

The reference's "every overlap chunk globs every index chunk's files" step (src/shmr_overlap.c:359-384) is, in the multi-GPU
form, the exchange of peregrine_amd/parallel.py: count tables all-gathered, pair records routed to their owner chunk.  Here
every rank runs its index chunk (the CPU oracle stands in for the GPU stage), the REAL protocol code (exchange_overlap) with a
numpy engine standing in for libpgx's record builder, and the records each rank receives must be, element for element and in
order, the insertion sequence of the reference's build_map over the concatenated lists for that chunk (oracle: orc_pair_records;
the oracle's overlap stage over the same sequence is pinned to the reference binaries by tests/test_oracle_vs_ref.py).
The GPU twin (tests/test_gpu_parallel.py) runs the same flow with the HIP kernels and compares ovlp_t streams with the reference."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _index_chunk_oracle(db, world, chunk):
    """the index stage of one chunk on the CPU oracle: reads with rid % N == chunk % N in idx order (shmr_index.c:157)"""
    import oracle_util as U
    from peregrine_amd.formats import MM_DTYPE
    parts = []
    for r, n, o in zip(db.rid, db.rlen, db.roff):
        if int(r) % world != chunk % world:
            continue
        l0 = U.orc_sketch_seqdb(db.seqdb[int(o):int(o) + int(n)], 80, 16, int(r))
        parts.append(U.orc_reduce(U.orc_reduce(l0, 6), 6))
    top = np.concatenate(parts) if parts else np.zeros(0, MM_DTYPE)
    return top, U.orc_count(top)


def _pipeline_worker(rank, world, port, out_dir, lower, upper):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from np_engine import NumpyEngine
    from peregrine_amd import simreads
    from peregrine_amd.parallel import exchange_overlap
    db = simreads.make_workload("tiny")
    rl, _ = db.by_rid()
    top, mc = _index_chunk_oracle(db, world, rank + 1)
    if upper == "empty0":  # rank 0 holds no shimmers at all and rank 1's first element sits AT the upper bound: the scan starts
        if rank == 0:      # inside rank 1's list, so rank 1's assumption "rank 0 holds the scan start" is wrong and it rebuilds
            top, mc = top[:0], mc[:0]
        upper = int(np.load(os.path.join(out_dir, "upper.npy")))
    if upper == "first":   # make the very first element of the concatenated list sit exactly AT the upper bound (strict rule)
        upper = int(np.load(os.path.join(out_dir, "upper.npy")))
    eng = NumpyEngine(rl)
    got, info = exchange_overlap(eng, rank, world, torch.from_numpy(top.view(np.uint8).copy()), torch.from_numpy(mc.view(np.uint8).copy()),
                                 mc_lower=lower, mc_upper=upper)
    np.save(os.path.join(out_dir, f"recs{rank}.npy"), got)
    np.save(os.path.join(out_dir, f"top{rank}.npy"), top)
    np.save(os.path.join(out_dir, f"mc{rank}.npy"), mc)
    assert info["received_records"] == len(got) and sum(info["received_per_source"]) == len(got)
    np.save(os.path.join(out_dir, f"redone{rank}.npy"), np.int64(info["scan_start_redone"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,lower,upper", [(2, 2, 240), (3, 2, 240), (2, 1, 30), (3, 2, "first"), (3, 2, "empty0"), (8, 2, 240), (8, 2, "empty0")])
def test_two_rank_pipeline(tmp_path, world, lower, upper):
    sys.path.insert(0, HERE)
    import oracle_util as U
    from peregrine_amd import simreads
    db = simreads.make_workload("tiny")
    rl, _ = db.by_rid()
    if upper in ("first", "empty0"):
        tops = [_index_chunk_oracle(db, world, c) for c in range(1, world + 1)]
        if upper == "empty0":
            tops[0] = (tops[0][0][:0], tops[0][1][:0])
        mc_all = np.concatenate([t[1] for t in tops])
        first = tops[0 if upper == "first" else 1][0][0]
        tot = int(mc_all["count"][mc_all["mer"] == (first["x"] >> np.uint64(8))].sum())
        np.save(tmp_path / "upper.npy", np.int64(tot))
        upper_v = tot
    else:
        upper_v = upper
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path), lower, upper), nprocs=world, join=True)
    mm = np.concatenate([np.load(tmp_path / f"top{r}.npy") for r in range(world)])      # chunk order = the reference's glob order
    mc = np.concatenate([np.load(tmp_path / f"mc{r}.npy") for r in range(world)])
    total = 0
    for r in range(world):
        want = U.orc_pair_records(mm, mc, rl, mychunk=r + 1, total=world, mc_lower=lower, mc_upper=upper_v)
        got = np.load(tmp_path / f"recs{r}.npy")
        assert len(got) == len(want), f"rank {r}: {len(got)} records received, build_map inserts {len(want)}"
        for f in ("key0", "key1", "y0", "npos", "dir"):
            assert np.array_equal(got[f], want[f]), f"rank {r}: field {f} differs"
        total += len(want)
    assert total > (1000 if upper not in ("first", "empty0") else 300)
    redone = [int(np.load(tmp_path / f"redone{r}.npy")) for r in range(world)]
    assert redone == [1 if upper == "empty0" else 0] * world, "the speculative scan start must hold except when rank 0 holds no anchor"
    if upper == "first":   # the strict rule really was in play: with the inclusive rule the first element would have anchored
        want_incl = U.orc_pair_records(mm, mc, rl, mychunk=1, total=1, mc_lower=lower, mc_upper=upper_v + 1)
        want_strict = U.orc_pair_records(mm, mc, rl, mychunk=1, total=1, mc_lower=lower, mc_upper=upper_v)
        assert len(want_incl) != len(want_strict) or not np.array_equal(want_incl["y0"], want_strict["y0"])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peregrine_amd.formats import MM_DTYPE
    from peregrine_amd.parallel import allgather_cat, allgather_exact, allgather_ints, alltoallv_bytes, scan_start
    rng = np.random.default_rng(100 + rank)
    n = 1000 + 37 * rank                        # ranks contribute different lengths (rank 1 is empty in the 3-rank run)
    mine = np.zeros(n if rank != 1 or world < 3 else 0, MM_DTYPE)
    mine["x"] = rng.integers(0, 2**40, len(mine), dtype=np.uint64)
    mine["y"] = (np.uint64(rank) << np.uint64(32)) | np.arange(len(mine), dtype=np.uint64)
    cat, sizes = allgather_cat(torch.from_numpy(mine.view(np.uint8).copy()), world)
    np.save(os.path.join(out_dir, f"got{rank}.npy"), cat.numpy().view(MM_DTYPE))
    np.save(os.path.join(out_dir, f"mine{rank}.npy"), mine)
    assert sizes[rank] == mine.nbytes
    ints = allgather_ints([rank * 7, -1 if rank else 5], world)
    assert ints == [[r * 7, -1 if r else 5] for r in range(world)]
    # all-to-all(v): rank r sends (r + 1) * (d + 1) bytes of value 10 r + d to rank d
    send = torch.cat([torch.full(((rank + 1) * (d + 1),), 10 * rank + d, dtype=torch.uint8) for d in range(world)])
    recv, rb = alltoallv_bytes(send, [(rank + 1) * (d + 1) for d in range(world)], world)
    want = torch.cat([torch.full(((s + 1) * (rank + 1),), 10 * s + rank, dtype=torch.uint8) for s in range(world)])
    assert torch.equal(recv, want) and rb == [(s + 1) * (rank + 1) for s in range(world)]
    # exact-size gather into a caller's buffer (what the seqdb replication uses): pieces land at their final offsets
    piece = torch.full((5 + 3 * rank,), rank + 1, dtype=torch.uint8)
    sizes2 = [5 + 3 * r for r in range(world)]
    buf = torch.zeros(sum(sizes2) + 7, dtype=torch.uint8)
    allgather_exact(piece, sizes2, out=buf)
    assert torch.equal(buf[:sum(sizes2)], torch.cat([torch.full((5 + 3 * r,), r + 1, dtype=torch.uint8) for r in range(world)])) and int(buf[-7:].sum()) == 0
    recv2, rb2 = alltoallv_bytes(send, [(rank + 1) * (d + 1) for d in range(world)], world, recv_bytes=[(s + 1) * (rank + 1) for s in range(world)])
    assert torch.equal(recv2, want) and rb2 == rb
    assert scan_start([-1, 4, 9][:world] + [0] * (world - 3), rank) == ([-1, 4][rank] if rank < 2 else 0)   # ranks behind the first holder start at 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_collectives_order_by_chunk(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = np.concatenate([np.load(tmp_path / f"mine{r}.npy") for r in range(world)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"got{r}.npy"), want), f"rank {r}"
    # chunk ownership of reads follows the reference: rid % N == chunk % N with 1-based chunks (shmr_index.c:157)
    from peregrine_amd.parallel import reads_of_chunk
    rid = np.arange(20, dtype=np.uint32)
    owned = [reads_of_chunk(rid, c, world) for c in range(1, world + 1)]
    assert sorted(np.concatenate(owned).tolist()) == rid.tolist()
    assert reads_of_chunk(rid, world, world).tolist() == [r for r in range(20) if r % world == 0]
