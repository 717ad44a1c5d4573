#!/usr/bin/env python3
"""Makes tests/golden/c4_stream_pins.json: the REFERENCE's overlap streams of BASELINE configs[3] at full size, pinned by hash.

Runs on the GPU box (the 93-Gbase read set is generated on the device by the seeded simulator, peregrine_amd.simreads; it does not
fit the build container), but everything hashed here is computed by the reference compiled in place (oracle/_ref, oracle/Makefile)
on the box's HOST cores:

  1. the read set -> <scratch>/sd.seqdb / sd.idx (the bytes the resident benchmark adopts); read_set_hash (bench.device_read_set_hash,
     every byte, on the device) and seqdb_sha256 (sha256sum of the file) identify the input;
  2. oracle/_ref/shmr_index  -p sd -t CH -c c -m 0 -l L   for c = 1..CH (CH processes side by side);
  3. oracle/_ref/shmr_overlap -p sd -l ix-L<L> -t CH -c c -M 240 for c = 1..CH (CH processes side by side, ~25-30 min at full size);
  4. per chunk: record count and the SHA-256 of the ovlp_t stream with the padding bytes (27 and 60..63 of every 64-byte record:
     src/shmr_overlap.c:163-173 writes whatever its stack held there) zeroed -- formats.masked_stream_sha256.

bench.py hashes the streams of its timed configuration the same way and reports `streams_match_pins`; tests/test_gpu_configs.py
asserts two of the chunks.  The GPU is only used in step 1 (and released before step 2).

usage: python tests/golden/make_c4_stream_pins.py [--workload c4] [--genome-mb MB] [--chunks CH] [--out gpurun_out/c4_stream_pins.json] [--scratch DIR]
(--genome-mb: a reduced set of the same recipe, to try the flow; the committed pins are the full-size ones)"""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def sha256_file(path, piece=1 << 28):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(piece)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--genome-mb", type=float, default=0)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "c4_stream_pins.json"))
    ap.add_argument("--scratch", default="/dev/shm")
    ap.add_argument("--keep-files", action="store_true", help="leave the scratch directory (seqdb, index and overlap files) in place and print its path")
    ap.add_argument("--only", default="", help="comma-separated overlap chunks to pin (default: all); the index stage always runs every chunk")
    ap.add_argument("--procs", type=int, default=0, help="reference overlap processes side by side (default: one per pinned chunk); a process at -l 1 holds "
                                                         "~13 GB of lists and tables at full size: price the host's memory first")
    a = ap.parse_args()
    import oracle_util as U
    from peregrine_amd import formats, simreads
    assert U.have_ref(), "oracle/_ref (the reference compiled in place) is not in this tree"
    sp = dict(levels=2, mc_upper=240, chunks=8)
    sp.update(simreads.STAGE_PARAMS.get(a.workload, {}))
    CH, lv = a.chunks or sp["chunks"], sp["levels"]
    t_start = time.perf_counter()
    d = tempfile.mkdtemp(prefix="pgx_pins_", dir=a.scratch)
    try:
        # ---- 1. the read set (GPU), its files, its hashes
        import torch
        import bench
        seq, total, rlen = simreads.make_workload_resident(a.workload, genome_mb=a.genome_mb or None)
        rid = np.arange(len(rlen), dtype=np.uint32)
        roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
        read_set_hash = bench.device_read_set_hash(seq, total)
        pre = os.path.join(d, "sd")
        simreads.write_seqdb_from_device(pre, seq, total, rid, rlen, roff)
        del seq
        torch.cuda.empty_cache()
        print(f"[pins +{time.perf_counter() - t_start:.0f} s] read set written: {len(rlen)} reads, {total} bytes; GPU released", flush=True)
        with cf.ThreadPoolExecutor(CH + 1) as ex:
            sha_job = ex.submit(sha256_file, pre + ".seqdb")
            # ---- 2. the reference's index chunks
            list(ex.map(lambda c: U.ref_run("shmr_index", "-p", pre, "-t", CH, "-c", c, "-m", 0, "-l", lv, "-o", os.path.join(d, "ix")), range(1, CH + 1)))
            print(f"[pins +{time.perf_counter() - t_start:.0f} s] reference index chunks done", flush=True)
            # ---- 3. the reference's overlap chunks
            t_index = time.perf_counter() - t_start
            t0 = time.perf_counter()
            chunks = [int(x) for x in a.only.split(",") if x] or list(range(1, CH + 1))
            procs = a.procs or len(chunks)
            mem_kb = {l.split(":")[0]: int(l.split()[1]) for l in open("/proc/meminfo") if l.split(":")[0] in ("MemTotal", "MemAvailable")}
            print(f"[pins] host memory: {mem_kb.get('MemTotal', 0) / 1e6:.0f} GB total, {mem_kb.get('MemAvailable', 0) / 1e6:.0f} GB available; {procs} overlap "
                  f"processes side by side over chunks {chunks}", flush=True)
            with cf.ThreadPoolExecutor(procs) as ex2:
                secs = list(ex2.map(lambda c: (time.perf_counter(), U.ref_run("shmr_overlap", "-p", pre, "-l", os.path.join(d, "ix-L%d" % lv), "-t", CH, "-c", c, "-M",
                                                                          sp["mc_upper"], "-o", os.path.join(d, "ov.%02d" % c)), time.perf_counter()), chunks))
            secs = [t1 - t0_ for t0_, _, t1 in secs]
            t_ovlp = time.perf_counter() - t0
            print(f"[pins +{time.perf_counter() - t_start:.0f} s] reference overlap chunks done ({t_ovlp:.0f} s, {procs} processes side by side)", flush=True)
            # ---- 4. the hashes
            shas = list(ex.map(lambda c: formats.masked_stream_sha256(os.path.join(d, "ov.%02d" % c)), chunks))
            seqdb_sha = sha_job.result()
        streams = [{"chunk": "%d of %d" % (c, CH), "records": os.path.getsize(os.path.join(d, "ov.%02d" % c)) // 64, "masked_sha256": shas[i], "reference_s": secs[i]}
                   for i, c in enumerate(chunks)]
        entry = {"workload": a.workload, "genome_mb": a.genome_mb or None, "chunks": CH, "levels": lv, "mc_upper": sp["mc_upper"],
                 "reads": int(len(rlen)), "seqdb_bytes": int(total), "read_set_hash": read_set_hash, "seqdb_sha256": seqdb_sha, "streams": streams,
                 "reference_overlap_leg_s": t_ovlp, "reference_overlap_processes": procs, "reference_index_and_files_s": t_index,
                 "host_mem_total_gb": mem_kb.get("MemTotal", 0) / 1e6, "host_cores": os.cpu_count(),
                 "made_by": "tests/golden/make_c4_stream_pins.py: oracle/_ref/shmr_index + shmr_overlap -t %d -c 1..%d -M %d on the files written from the "
                            "device-resident read set; masked = bytes 27 and 60..63 of every record zeroed" % (CH, CH, sp["mc_upper"])}
        key = a.workload if not a.genome_mb else "%s_%gMb" % (a.workload, a.genome_mb)
        out = {}
        if os.path.exists(a.out):
            try:
                out = json.load(open(a.out))
            except Exception:
                out = {}
        out[key] = entry
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
        print(json.dumps({k: v for k, v in entry.items() if k != "streams"}), flush=True)
        print(f"[pins +{time.perf_counter() - t_start:.0f} s] written {a.out}", flush=True)
        if a.keep_files:
            print("files kept in", d)
    finally:
        if not a.keep_files:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
