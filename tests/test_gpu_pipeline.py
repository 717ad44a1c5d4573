"""The reference's own pipeline test (test/ecoli_K12/run_test.sh, test/genome_mapping/run_test.sh), step for step, with the
drop-in executables of bin/ (Python) and bin/native/ (one C binary against the C-ABI) next to the REAL reference binaries (oracle/_ref, prebuilt): FASTA files -> shmr_mkseqdb ->
shmr_index (several chunks) -> shmr_overlap (several chunks) -> cat | shmr_dedup -> shmr_map reads->contigs and
contigs->contigs.  Every output file must be byte-identical (MC files: same multiset of (mer, count))."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import oracle_util as U
from peregrine_amd import formats, simreads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mine(tool, *args, cwd=None, stdin=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bin", tool), *map(str, args)], cwd=cwd, check=True, input=stdin,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout


def _native(tool, *args, cwd=None, stdin=None):
    """the native multi-call binary (peregrine_amd/csrc/pgx_cli.c), through its per-tool link when the link survived the copy"""
    exe = os.path.join(ROOT, "bin", "native", tool)
    cmd = [exe] if os.path.exists(exe) else [os.path.join(ROOT, "bin", "native", "pgx_cli"), tool]
    return subprocess.run([*cmd, *map(str, args)], cwd=cwd, check=True, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout


def _ref(tool, *args, cwd=None, stdin=None):
    return subprocess.run([os.path.join(U.REF_DIR, tool), *map(str, args)], cwd=cwd, check=True, input=stdin,
                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout


@pytest.mark.skipif(not (U.have_ref() and os.path.exists(os.path.join(U.REF_DIR, "shmr_map"))), reason="needs the prebuilt reference binaries (oracle/_ref)")
def test_pipeline_like_the_reference_test_scripts(tmp_path):
    g = simreads.make_genome(300_000, 21, repeat_families=2, repeat_len=3000, repeat_copies=4, divergence=0.02, tandem=2)
    db = simreads.simulate_reads(g, coverage=14.0, seed=3, mean_len=7000, sd_len=1500, err=0.01, n_files=1)
    # reads in three FASTA files, contigs = two pieces of the genome (one reverse-complemented) in a fourth
    third = -(-db.n_reads // 3)
    paths = []
    lut = np.full(16, ord("N"), np.uint8)
    lut[[1, 2, 4, 8]] = [ord(c) for c in "ACGT"]
    for i in range(3):
        p = tmp_path / f"reads_{i}.fa"
        with open(p, "wb") as f:
            for r in range(i * third, min(db.n_reads, (i + 1) * third)):
                o, n = int(db.roff[r]), int(db.rlen[r])
                f.write(b">read%06d\n" % r + lut[db.seqdb[o:o + n] & 0x0F].tobytes() + b"\n")
        paths.append(str(p))
    (tmp_path / "seq_dataset.lst").write_text("\n".join(paths) + "\n")
    acgt = np.frombuffer(b"ACGT", np.uint8)
    with open(tmp_path / "ctg.fa", "wb") as f:
        f.write(b">ctg0\n" + acgt[g[:180_000]].tobytes() + b"\n>ctg1\n" + acgt[(3 - g[150_000:])[::-1]].tobytes() + b"\n")
    (tmp_path / "ctg.lst").write_text(str(tmp_path / "ctg.fa") + "\n")

    out = {}
    for who, run in (("ref", _ref), ("mine", _mine), ("native", _native)):
        d = tmp_path / who
        (d / "index").mkdir(parents=True)
        (d / "ovlp").mkdir()
        ix = str(d / "index")
        run("shmr_mkseqdb", "-p", f"{ix}/seq_dataset", "-d", tmp_path / "seq_dataset.lst")
        run("shmr_mkseqdb", "-p", f"{ix}/p_ctg", "-d", tmp_path / "ctg.lst")
        for c in (1, 2, 3):
            run("shmr_index", "-p", f"{ix}/seq_dataset", "-r", 6, "-t", 3, "-c", c, "-o", f"{ix}/shmr")
        run("shmr_index", "-p", f"{ix}/p_ctg", "-r", 6, "-t", 1, "-c", 1, "-o", f"{ix}/p_ctg")
        for c in (1, 2):
            run("shmr_overlap", "-p", f"{ix}/seq_dataset", "-l", f"{ix}/shmr-L2", "-t", 2, "-c", f"{c:02d}", "-o", d / "ovlp" / f"ovlp.{c:02d}")
        cat = b"".join((d / "ovlp" / f"ovlp.{c:02d}").read_bytes() for c in (1, 2))
        out[who, "preads.ovl"] = run("shmr_dedup", stdin=cat)
        out[who, "read_map.txt"] = run("shmr_map", "-r", f"{ix}/p_ctg", "-m", f"{ix}/p_ctg-L2", "-p", f"{ix}/seq_dataset", "-l",
                                       f"{ix}/shmr-L2", "-t", 1, "-c", 1)
        out[who, "ref2ref.out"] = run("shmr_map", "-r", f"{ix}/p_ctg", "-m", f"{ix}/p_ctg-L2", "-p", f"{ix}/p_ctg", "-l", f"{ix}/p_ctg-L2",
                                      "-t", 1, "-c", 1)
    names = sorted(os.listdir(tmp_path / "ref" / "index")) + [os.path.join("..", "ovlp", f) for f in sorted(os.listdir(tmp_path / "ref" / "ovlp"))]
    for who in ("mine", "native"):   # the Python drop-ins of bin/ and the native multi-call binary of bin/native/
        assert sorted(os.listdir(tmp_path / who / "index")) == sorted(os.listdir(tmp_path / "ref" / "index"))
        for n in names:
            a = (tmp_path / "ref" / "index" / n).read_bytes()
            b = (tmp_path / who / "index" / n).read_bytes()
            if "-MC-" in n:   # khash slot order in the reference, sorted here: same (mer, count) multiset
                pa = formats.mc_as_sorted_pairs(formats.read_mm_count(str(tmp_path / "ref" / "index" / n)))
                pb = formats.mc_as_sorted_pairs(formats.read_mm_count(str(tmp_path / who / "index" / n)))
                assert np.array_equal(pa, pb), (who, n)
            else:
                assert a == b, (who, n)
        for k in ("preads.ovl", "read_map.txt", "ref2ref.out"):
            assert out["ref", k] == out[who, k], (who, k)
    assert out["ref", "preads.ovl"].count(b"\n") > 500 and out["ref", "read_map.txt"].count(b"\n") > 500
    # the same overlap chunks once more with the greedy walk forced onto the GPU (production picks it from 0.2 M pair records;
    # this set is smaller): the files must not change
    env = dict(os.environ, PGX_GPU_REPLAY="1")
    for c in (1, 2):
        exe = os.path.join(ROOT, "bin", "native", "pgx_cli")
        o = tmp_path / f"ovlp_dev.{c:02d}"
        subprocess.run([exe, "shmr_overlap", "-p", str(tmp_path / "native" / "index" / "seq_dataset"), "-l", str(tmp_path / "native" / "index" / "shmr-L2"),
                        "-t", "2", "-c", f"{c:02d}", "-o", str(o)], check=True, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert o.read_bytes() == (tmp_path / "ref" / "ovlp" / f"ovlp.{c:02d}").read_bytes(), c


@pytest.mark.skipif(not U.have_ref(), reason="needs the prebuilt reference binaries (oracle/_ref)")
@pytest.mark.parametrize("device_replay", ["default", "1"])
def test_served_mode_writes_the_same_files(tmp_path, device_replay):
    """`pgx_cli serve -p <prefix>` keeps the read database in HBM; the native shmr_index / shmr_overlap drop-ins attach to it through
    <prefix>.pgx.sock (round 4: a job's 8 + 8 chunk commands upload the seqdb once).  Files byte-identical to the reference's; a
    command for another prefix, a replaced .seqdb file and a dead server all fall back to the stand-alone path; records also go to a
    pipe (-o /dev/stdout: the sequential writer).  Round 6: the server keeps the lists its index commands wrote on the device (and reads files
    it holds no current copy of), sends the device replay's records straight from the device into the mapped output file (device_replay "1":
    PGX_GPU_REPLAY=1 in the server; "default": a set this small takes the host replay and the host-array writer) and answers a client when its
    file is complete while the next command may already run: two commands in flight, and lists that changed behind the server's back."""
    import signal
    import time
    g = simreads.make_genome(200_000, 23, repeat_families=1, repeat_len=3000, repeat_copies=4, divergence=0.02, tandem=2)
    db = simreads.simulate_reads(g, coverage=12.0, seed=5, mean_len=7000, sd_len=1500)
    pre = str(tmp_path / "sd")
    formats.write_seqdb(pre, db)
    other = str(tmp_path / "other")
    formats.write_seqdb(other, simreads.simulate_reads(simreads.make_genome(60_000, 24), coverage=8.0, seed=6, mean_len=5000, sd_len=500))
    cli = os.path.join(ROOT, "bin", "native", "pgx_cli")
    (tmp_path / "ref").mkdir(), (tmp_path / "srv").mkdir(), (tmp_path / "alone").mkdir()
    for c in (1, 2):
        _ref("shmr_index", "-p", pre, "-t", 2, "-c", c, "-m", 0, "-o", tmp_path / "ref" / "ix")
    for c in (1, 2, 3):
        _ref("shmr_overlap", "-p", pre, "-l", tmp_path / "ref" / "ix-L2", "-t", 3, "-c", c, "-o", tmp_path / "ref" / f"ov.{c}")
    env = dict(os.environ)
    if device_replay != "default":
        env["PGX_GPU_REPLAY"] = device_replay
    srv = subprocess.Popen([cli, "serve", "-p", pre], stderr=subprocess.PIPE, text=True, env=env)
    try:
        for _ in range(600):
            if os.path.exists(pre + ".pgx.sock") or srv.poll() is not None:
                break
            time.sleep(0.1)
        assert srv.poll() is None and os.path.exists(pre + ".pgx.sock"), srv.stderr.read() if srv.poll() is not None else "no socket"
        t0 = time.perf_counter()
        for c in (1, 2):   # relative output paths: resolved against the CLIENT's directory
            r = subprocess.run([cli, "shmr_index", "-p", pre, "-t", "2", "-c", str(c), "-m", "0", "-o", "ix"], cwd=tmp_path / "srv", check=True, capture_output=True, text=True)
            assert "resident: pgx_cli serve" in r.stderr
        for c in (1, 2, 3):
            subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", str(c), "-o", f"ov.{c}"], cwd=tmp_path / "srv", check=True, capture_output=True)
        served_s = time.perf_counter() - t0
        for c in (1, 2):
            assert (tmp_path / "srv" / f"ix-L2-{c:02d}-of-02.dat").read_bytes() == (tmp_path / "ref" / f"ix-L2-{c:02d}-of-02.dat").read_bytes()
        for c in (1, 2, 3):
            assert (tmp_path / "srv" / f"ov.{c}").read_bytes() == (tmp_path / "ref" / f"ov.{c}").read_bytes()
        # two overlap commands in flight (a scheduler with two job slots): the second one's stage runs while the first one's file is completed
        ps = [subprocess.Popen([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", str(c), "-o", f"par.{c}"], cwd=tmp_path / "srv") for c in (1, 2, 3)]
        assert [q.wait(timeout=120) for q in ps] == [0, 0, 0]
        for c in (1, 2, 3):
            assert (tmp_path / "srv" / f"par.{c}").read_bytes() == (tmp_path / "ref" / f"ov.{c}").read_bytes()
        # lists the server never wrote (the reference's own index files): read from the files
        subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", str(tmp_path / "ref" / "ix-L2"), "-t", "3", "-c", "3", "-o", "fromref.3"], cwd=tmp_path / "srv", check=True, capture_output=True)
        assert (tmp_path / "srv" / "fromref.3").read_bytes() == (tmp_path / "ref" / "ov.3").read_bytes()
        # a job's lists REPLACED behind the server's back (another chunking under the same prefix): neither the device copies of the index commands nor the
        # assembled lists may be used -- the streams must be those of the new files (here: T = 1 index, the reference's T = 1 run as the expectation)
        _ref("shmr_index", "-p", pre, "-t", 1, "-c", 1, "-m", 0, "-o", tmp_path / "ref" / "one")
        _ref("shmr_overlap", "-p", pre, "-l", tmp_path / "ref" / "one-L2", "-t", 3, "-c", 2, "-o", tmp_path / "ref" / "one.2")
        for f in (tmp_path / "srv").glob("ix-L2-*"):
            f.unlink()
        shutil.copy(tmp_path / "ref" / "one-L2-01-of-01.dat", tmp_path / "srv" / "ix-L2-01-of-01.dat")
        shutil.copy(tmp_path / "ref" / "one-L2-MC-01-of-01.dat", tmp_path / "srv" / "ix-L2-MC-01-of-01.dat")
        subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", "2", "-o", "one.2"], cwd=tmp_path / "srv", check=True, capture_output=True)
        assert (tmp_path / "srv" / "one.2").read_bytes() == (tmp_path / "ref" / "one.2").read_bytes()
        for c in (1, 2):   # ... and back to the job's own lists (re-indexed by the server: fresh device copies)
            (tmp_path / "srv" / "ix-L2-01-of-01.dat").unlink(missing_ok=True), (tmp_path / "srv" / "ix-L2-MC-01-of-01.dat").unlink(missing_ok=True)
            subprocess.run([cli, "shmr_index", "-p", pre, "-t", "2", "-c", str(c), "-m", "0", "-o", "ix"], cwd=tmp_path / "srv", check=True, capture_output=True)
        # -o /dev/stdout with a LIVE server: the server cannot open the client's descriptor -- the command runs stand-alone and the
        # records arrive on the client's stdout; a file under /dev/shm is a plain file and is served like any other
        r = subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", "2", "-o", "/dev/stdout"], cwd=tmp_path / "srv", check=True, capture_output=True)
        assert r.stdout == (tmp_path / "ref" / "ov.2").read_bytes()
        shm = None
        if os.path.isdir("/dev/shm"):
            import tempfile
            shm = tempfile.mkdtemp(prefix="pgx_srv_", dir="/dev/shm")
            subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", "1", "-o", os.path.join(shm, "ov.1")], cwd=tmp_path / "srv", check=True, capture_output=True)
            assert open(os.path.join(shm, "ov.1"), "rb").read() == (tmp_path / "ref" / "ov.1").read_bytes()
            shutil.rmtree(shm, ignore_errors=True)
        # errors come back with the exit status and the library's message (chunk 7 of 3)
        bad = subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", "ix-L2", "-t", "3", "-c", "7", "-o", "x"], cwd=tmp_path / "srv", capture_output=True, text=True)
        assert bad.returncode != 0 and "pgx_overlap_chunk_db failed" in bad.stderr
        # another prefix: no socket for it -> stand-alone
        r = subprocess.run([cli, "shmr_index", "-p", other, "-m", "0", "-o", str(tmp_path / "alone" / "o")], check=True, capture_output=True, text=True)
        assert "resident" not in r.stderr and os.path.exists(tmp_path / "alone" / "o-L2-01-of-01.dat")
        # the .seqdb file replaced behind the server's back (same bytes, new mtime): the server declines, the client runs alone
        os.utime(pre + ".seqdb", ns=(1, 1))
        r = subprocess.run([cli, "shmr_index", "-p", pre, "-t", "2", "-c", "1", "-m", "0", "-o", str(tmp_path / "alone" / "ix")], check=True, capture_output=True, text=True)
        assert "resident" not in r.stderr
        assert (tmp_path / "alone" / "ix-L2-01-of-02.dat").read_bytes() == (tmp_path / "ref" / "ix-L2-01-of-02.dat").read_bytes()
    finally:
        srv.send_signal(signal.SIGTERM)
        srv.wait(timeout=30)
    assert not os.path.exists(pre + ".pgx.sock")
    # a dead server's leftovers do not matter either; and the records through a pipe (no pwrite there)
    open(pre + ".pgx.sock", "w").close()
    r = subprocess.run([cli, "shmr_overlap", "-p", pre, "-l", str(tmp_path / "ref" / "ix-L2"), "-t", "3", "-c", "2", "-o", "/dev/stdout"], check=True, capture_output=True)
    assert r.stdout == (tmp_path / "ref" / "ov.2").read_bytes()
    print(f"served: 2 index + 3 overlap commands in {served_s:.2f} s")
