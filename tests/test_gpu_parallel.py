"""The multi-GPU exchange with the HIP kernels, on one GPU:
 * one process plays all ranks in turn (index chunk -> prepare -> scatter per chunk, records regrouped per destination as the
   all-to-all would, overlap stage per chunk) and every chunk's ovlp_t stream must equal the reference's `-t N -c c`;
 * two processes under gloo (PGX's debug backend: RCCL refuses two ranks on one GPU) run the real protocol code
   (peregrine_amd.parallel.exchange_overlap with GpuEngine) end to end."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

import oracle_util as U
from peregrine_amd import formats, simreads
from peregrine_amd.parallel import REC_BYTES, GpuEngine, scan_start
from peregrine_amd.shimmer import ResidentDB

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def _reference_chunks(db, tmp, N, levels=2, extra=()):
    """the reference (or, without the prebuilt binaries, the oracle) on files: N index chunks, N overlap chunks"""
    pre = os.path.join(tmp, "sd")
    formats.write_seqdb(pre, db)
    outs = []
    for c in range(1, N + 1):
        if U.have_ref():
            U.ref_run("shmr_index", "-p", pre, "-t", N, "-c", c, "-m", 0, "-l", levels, "-o", os.path.join(tmp, "ix"))
        else:
            U.orc_index_chunk(pre, os.path.join(tmp, "ix"), N, c, levels, 6, 0, 80, 16)
    for c in range(1, N + 1):
        o = os.path.join(tmp, "ov.%02d" % c)
        if U.have_ref():
            U.ref_run("shmr_overlap", "-p", pre, "-l", os.path.join(tmp, "ix-L%d" % levels), "-t", N, "-c", c, "-o", o, *extra)
        else:
            kw = dict(zip(extra[0::2], extra[1::2]))
            U.orc_overlap_chunk(pre, os.path.join(tmp, "ix-L%d" % levels), o, N, c, 4, int(kw.get("-m", 2)), int(kw.get("-M", 240)))
        outs.append(formats.read_ovlp(o))
    return outs


@pytest.mark.parametrize("N,lower,upper", [(2, 2, 240), (3, 2, 240), (3, 1, 12), (8, 2, 240)])
def test_scatter_and_records_path_equal_reference(tmp_path, N, lower, upper):
    db = simreads.make_workload("small")
    want = _reference_chunks(db, str(tmp_path), N, extra=("-m", lower, "-M", upper))
    dev = torch.device("cuda", 0)
    rdb = ResidentDB(db, 0)
    eng = GpuEngine(rdb, dev)
    tops, mcs = [], []
    for c in range(1, N + 1):
        _, top, mc = eng.index(N, c)
        tops.append(top.clone()), mcs.append(mc.clone())
    counts_all = torch.cat(mcs)
    firsts = []
    for r in range(N):
        firsts.append(eng.pairs_prepare(tops[r], counts_all, lower, upper))
    sends, counts = [], []
    for r in range(N):
        eng.pairs_prepare(tops[r], counts_all, lower, upper)
        s, cnt = eng.pairs_scatter(N, scan_start(firsts, r))
        sends.append(s.clone()), counts.append(cnt)
    for d in range(N):   # what rank d receives: source-major
        parts = []
        for r in range(N):
            o = sum(counts[r][:d]) * REC_BYTES
            parts.append(sends[r][o:o + counts[r][d] * REC_BYTES])
        recv = torch.cat(parts)
        (ov, st), = [eng.overlap_records(recv, N, d + 1, mc_lower=lower, mc_upper=upper)]
        assert len(want[d]) > (500 if N < 8 else 100) and formats.ovlp_fields_equal(ov, want[d]), f"chunk {d + 1} of {N}"
        # and the same chunk through the all-gather form (device lists): pgx_overlap_resident_dev
        allmm = torch.cat(tops)
        torch.cuda.synchronize()
        ov2, _ = rdb.overlap_dev(allmm.data_ptr(), allmm.numel() // 16, counts_all.data_ptr(), counts_all.numel() // 16, total_chunk=N,
                                 mychunk=d + 1, mc_lower=lower, mc_upper=upper)
        assert formats.ovlp_fields_equal(ov2, want[d])
    rdb.close()


def test_seqdb_from_device_pointer():
    db = simreads.make_workload("tiny")
    t = torch.from_numpy(db.seqdb).to("cuda:0")
    torch.cuda.synchronize()
    a = ResidentDB.from_device(t.data_ptr(), t.numel(), db.rid, db.rlen, db.roff, 0)
    b = ResidentDB(db, 0)
    ia, ib = a.index(), b.index()
    assert np.array_equal(ia.top, ib.top) and len(ia.top) > 100
    a.close(), b.close()


def test_shimmer_list_of_another_seqdb_is_rejected():
    """rids / positions that do not fit the loaded seqdb must fail with PGX_EARG, not read out of bounds"""
    from peregrine_amd import _lib
    big = simreads.make_workload("small")
    small = simreads.make_workload("tiny")
    rb, rs = ResidentDB(big, 0), ResidentDB(small, 0)
    ix = rb.index()
    with pytest.raises(_lib.PgxError, match="do not fit the read database"):
        rs.overlap(ix.top, ix.top_mc)
    ix2 = rs.index()
    bad = ix2.top.copy()
    bad["y"][5] = (bad["y"][5] & np.uint64(0xFFFFFFFF00000001)) | np.uint64(0x7FFFFFFE)   # a position far beyond the read
    with pytest.raises(_lib.PgxError, match="do not fit the read database"):
        rs.overlap(bad, ix2.top_mc)
    ov, _ = rs.overlap(ix2.top, ix2.top_mc)   # the context is still usable
    assert len(ov) > 100
    rb.close(), rs.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from peregrine_amd.parallel import exchange_overlap
    db = simreads.make_workload("small")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rdb = ResidentDB(db, 0)
    eng = GpuEngine(rdb, dev)
    _, top, mc = eng.index(world, rank + 1)
    (ov, st), info = exchange_overlap(eng, rank, world, top, mc)
    np.save(os.path.join(out_dir, f"ov{rank}.npy"), ov)
    assert st["n_records"] == len(ov) and info["received_records"] > 0
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pipeline_on_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    db = simreads.make_workload("small")
    want = _reference_chunks(db, str(tmp_path), 2)
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"ov{r}.npy")
        assert formats.ovlp_fields_equal(got, want[r]), f"rank {r} (overlap chunk {r + 1} of 2)"


def _rccl_one_rank(rank, port, out_dir):
    """a ONE-rank job over RCCL with PGX_FORCE_EXCHANGE=1: every collective of the multi-GPU path is really issued, on device
    views (the nccl code path had never executed)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", PGX_FORCE_EXCHANGE="1")
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from peregrine_amd import parallel
    assert parallel.forced() and dist.get_backend() == "nccl"
    db = simreads.make_workload("small")
    seq_all, total, rlen = parallel.gather_seqdb(db.seqdb, db.rlen, 1, dev)        # all-gather into the final buffer, adopted without a copy
    assert total == db.seqdb.size and np.array_equal(rlen, db.rlen) and seq_all.is_cuda and seq_all.numel() == total + 1024
    rdb = ResidentDB.adopt_device(seq_all, total, db.rid, db.rlen, db.roff, 0)
    eng = GpuEngine(rdb, dev)
    for it in range(2):   # twice: the second step re-uses every workspace the first one left behind
        os.environ["PGX_FORCE_UNEVEN"] = str(it)   # second pass: the all_gather over views of unequal pieces (grouped broadcasts)
        _, top, mc = eng.index(1, 1)
        assert top.is_cuda and mc.is_cuda
        (ov, st), info = parallel.exchange_overlap(eng, 0, 1, top, mc)
        assert info["received_records"] == info["sent_records"] > 0 and not info["scan_start_redone"]
    np.save(os.path.join(out_dir, "ov_rccl.npy"), ov)
    # a view of the index workspace that an index call rewrote must be refused, not silently read
    from peregrine_amd import _lib
    _, top, mc = eng.index(1, 1)
    eng.pairs_prepare(top, mc, 2, 240)
    eng.index(1, 1)
    try:
        eng.pairs_scatter(1, 0)
        stale = "accepted"
    except _lib.PgxError as e:
        stale = str(e)
    assert "rewrote the shimmer list" in stale, stale
    rdb.close()
    ref = ResidentDB(db, 0)                                                        # the single-chunk path of the same process
    _, ov1, _ = ref.index_overlap()
    np.save(os.path.join(out_dir, "ov_direct.npy"), ov1)
    ref.close()
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_exchange_as_one_rank_job(tmp_path):
    import torch.multiprocessing as mp
    db = simreads.make_workload("small")
    want = _reference_chunks(db, str(tmp_path), 1)[0]
    mp.spawn(_rccl_one_rank, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    got, direct = np.load(tmp_path / "ov_rccl.npy"), np.load(tmp_path / "ov_direct.npy")
    assert len(want) > 1000 and formats.ovlp_fields_equal(got, want), "records through the RCCL exchange differ from the reference's"
    assert formats.ovlp_fields_equal(direct, want)


@pytest.mark.skipif(not U.have_ref(), reason="needs the prebuilt reference binaries (oracle/_ref)")
@pytest.mark.parametrize("ranks,chunks", [(1, 4), (2, 2), (2, 4)])
def test_bench_strong_scaling_form_of_configs3_against_the_reference(ranks, chunks):
    """bench.py's c4 family (BASELINE configs[3] AS STATED: one read set on every rank, the job's chunks dealt to the ranks, strong
    scaling) on a 12 Mb genome of the same recipe: one rank with 4 chunks; two ranks sharing the GPU under gloo with 2 chunks (one
    chunk per rank: count all-gather + pair-record all-to-all) and with 4 (two per rank: lists all-gathered round by round).
    (Round 5 tried four and eight ranks on the one GPU as the dress rehearsal of an 8-GPU node: the hardware scheduler time-slices the
    processes and the read simulator alone did not finish in five minutes -- (4, 8) passed in 313 s, (8, 8) not within 900.  The world-8 form is
    covered by test_scatter_and_records_path_equal_reference[8-...] here -- every kernel of the one-chunk-per-rank path, eight chunks, one
    process playing the ranks in turn -- and by tests/test_parallel_gloo.py at world 8 -- the real protocol code, eight processes, CPU.)  Every rank
    compares the ovlp_t stream of each of its chunks with oracle/_ref/shmr_overlap -t C -c c on files (--check-ref)."""
    import json
    import subprocess
    cmd = [sys.executable]
    env = dict(os.environ)
    if ranks > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
        env["PGX_BENCH_BACKEND"] = "gloo"
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--workload", "c4", "--genome-mb", "12", "--chunks", str(chunks), "--check-ref",
            "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    chk = line["check_vs_reference"]
    assert chk["all_equal"] is True and len(chk["chunks"]) == chunks and all(c["records"] > 50_000 for c in chk["chunks"]), chk
    assert line["scaling"] == "strong" and line["n_gpus"] == ranks and line["records_per_step"] == sum(c["records"] for c in chk["chunks"])
    assert line["world_size"] == ranks and len(line["per_rank"]) == ranks and all(r["hbm_bytes_in_use"] > 0 for r in line["per_rank"])
    assert line["read_set_hash_equal_on_all_ranks"] is True
    if ranks > 1 and chunks == ranks:   # the all-to-all form: every record sent is received, and they are the records the overlap stages consumed
        sent, recv = sum(r["sent_records"] for r in line["per_rank"]), sum(r["received_records"] for r in line["per_rank"])
        assert sent == recv > 0


def test_shutdown_forgets_device_state():
    """pgx_shutdown() must reset the plan caches / held buffers, or a pgx_seqdb that outlives a shutdown + init runs the
    sketch kernels on freed descriptors.  Runs in its own process (one context per process)."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from peregrine_amd import _lib, simreads\n"
        "from peregrine_amd.shimmer import ResidentDB\n"
        "db = simreads.make_workload('tiny')\n"
        "a = ResidentDB(db, 0); t1 = a.index().top.copy(); a.index(); a.close()\n"
        "_lib.shutdown()\n"
        "b = ResidentDB(db, 0); t2 = b.index().top.copy(); t3 = b.index().top.copy(); b.close()\n"
        "assert len(t1) > 100 and np.array_equal(t1, t2) and np.array_equal(t1, t3)\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
