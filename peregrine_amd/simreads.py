"""Deterministic synthetic read sets (numpy PCG64) for parity tests and benchmarks.

The recipe follows the *spirit* of the reference's read simulators
(/root/reference/test/ecoli_K12/simulate_reads.py:10-45, py-utils/simread.py:8-41): reads of length
int(15000 + N(0,1500)) drawn from a genome extended by its first 40 kb, every base hit with probability
1 % by one of {A, C, G, T, deletion, base+A, base+C, base+G, base+T}, half of the reads reverse-complemented.
K12MG1655.fa / CHM13 are not obtainable offline, so genomes are uniform-random (optionally with planted
repeat families), SURVEY.md 8(d).  This is our own vectorised generator, not the reference's script.
"""
from __future__ import annotations

import numpy as np

from .formats import SeqDB

_COMP = np.array([3, 2, 1, 0], np.uint8)


def make_genome(length: int, seed: int, repeat_families: int = 0, repeat_len: int = 6000,
                repeat_copies: int = 0, divergence: float = 0.01, tandem: int = 0) -> np.ndarray:
    """Uniform-random genome of 2-bit codes; optional planted repeat families / tandem arrays / low complexity."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = rng.integers(0, 4, size=length, dtype=np.uint8)
    for _ in range(repeat_families):
        unit = rng.integers(0, 4, size=repeat_len, dtype=np.uint8)
        for _c in range(repeat_copies):
            s = int(rng.integers(0, length - repeat_len))
            cp = unit.copy()
            hit = rng.random(repeat_len) < divergence
            cp[hit] = rng.integers(0, 4, size=int(hit.sum()), dtype=np.uint8)
            g[s:s + repeat_len] = cp
    for _ in range(tandem):
        period = int(rng.integers(1, 40))
        n = int(rng.integers(200, 3000))
        unit = rng.integers(0, 4, size=period, dtype=np.uint8)
        s = int(rng.integers(0, length - n))
        g[s:s + n] = np.resize(unit, n)
    return g


def simulate_reads(genome: np.ndarray, n_reads: int | None = None, coverage: float | None = None,
                   seed: int = 42, mean_len: int = 15000, sd_len: int = 1500, err: float = 0.01,
                   wrap: int = 40000, n_files: int = 1, batch_reads: int = 2048, min_len: int = 200) -> SeqDB:
    """Simulate reads and return them already in seqdb encoding (1 byte/base, both strands)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    wrap = min(wrap, genome.shape[0])
    ext = np.concatenate([genome, genome[:wrap]])
    if n_reads is None:
        n_reads = int(coverage * ext.shape[0] / mean_len)
    per_file = -(-n_reads // n_files)
    out_chunks, rlen_all, names = [], [], []
    done = 0
    while done < n_reads:
        nb = min(batch_reads, n_reads - done)
        tl = np.maximum((mean_len + rng.normal(0.0, sd_len, nb)).astype(np.int64), min_len)
        tl = np.minimum(tl, ext.shape[0])
        st = (rng.random(nb) * (ext.shape[0] - tl + 1)).astype(np.int64)
        rc = rng.integers(0, 2, nb).astype(bool)
        tot = int(tl.sum())
        seg0 = np.concatenate([[0], np.cumsum(tl)[:-1]])
        src = np.repeat(st - seg0, tl) + np.arange(tot, dtype=np.int64)
        tmpl = ext[src]
        # error model: one draw per template base
        hit = rng.random(tot) < err
        kind = rng.integers(0, 9, tot, dtype=np.int8)
        kind[~hit] = -1
        sub = (kind >= 0) & (kind < 4)
        base = tmpl.copy()
        base[sub] = kind[sub].astype(np.uint8)
        emit = np.ones(tot, np.int64)
        emit[kind == 4] = 0
        ins = kind >= 5
        emit[ins] = 2
        olen = np.add.reduceat(emit, seg0) if tot else np.zeros(0, np.int64)
        ototal = int(emit.sum())
        opos = np.cumsum(emit) - emit                      # output offset of each template base
        raw = np.empty(ototal, np.uint8)
        keep = emit > 0
        raw[opos[keep]] = base[keep]
        raw[opos[ins] + 1] = (kind[ins] - 5).astype(np.uint8)
        # strand flip + seqdb encoding need the within-read mirrored index
        oseg0 = np.concatenate([[0], np.cumsum(olen)[:-1]])
        seg_start = np.repeat(oseg0, olen)
        seg_len = np.repeat(olen, olen)
        idx = np.arange(ototal, dtype=np.int64)
        mirror = 2 * seg_start + seg_len - 1 - idx
        rc_rep = np.repeat(rc, olen)
        codes = np.where(rc_rep, _COMP[raw[mirror]], raw)
        enc = (np.uint8(1) << codes) | ((np.uint8(8) >> codes[mirror]) << np.uint8(4))
        out_chunks.append(enc.astype(np.uint8))
        rlen_all.append(olen.astype(np.uint32))
        for i in range(nb):
            j = done + i
            names.append("%02d/%06d/0_%d" % (j // per_file, j % per_file, int(olen[i])))
        done += nb
    seqdb = np.concatenate(out_chunks) if out_chunks else np.zeros(0, np.uint8)
    rlen = np.concatenate(rlen_all) if rlen_all else np.zeros(0, np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    return SeqDB(seqdb, np.arange(n_reads, dtype=np.uint32), rlen, roff, names)


def make_genome_torch(length: int, seed: int, device: str = "cuda", repeat_families: int = 0, repeat_len: int = 6000,
                      repeat_copies: int = 0, divergence: float = 0.01, tandem: int = 0, homopolymers: int = 0):
    """Genome of 2-bit codes on the device: uniform-random background (the same stream simulate_reads_torch always drew)
    plus, optionally, the repeat content SURVEY.md 8(d) C4 asks for in place of CHM13: `repeat_families` families of
    `repeat_copies` copies of a `repeat_len` unit, each copy diverged independently; `tandem` arrays of a 1..39-base unit,
    200..3000 bases long; `homopolymers` runs of 30..400 bases.  Plants come from a second seeded generator, so a
    repeat-free genome is bit-identical to what earlier rounds benchmarked."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    g = torch.randint(0, 4, (length,), dtype=torch.uint8, device=dev, generator=gen)
    if not (repeat_families or tandem or homopolymers):
        return g
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    pg = torch.Generator(device=dev)
    pg.manual_seed(seed + 0x9E3779B9)
    for _ in range(repeat_families):
        unit = torch.randint(0, 4, (repeat_len,), dtype=torch.uint8, device=dev, generator=pg)
        starts = rng.integers(0, length - repeat_len, repeat_copies)
        for s in starts:
            hit = torch.rand(repeat_len, device=dev, generator=pg) < divergence
            sub = torch.randint(0, 4, (repeat_len,), dtype=torch.uint8, device=dev, generator=pg)
            g[int(s):int(s) + repeat_len] = torch.where(hit, sub, unit)
    for _ in range(tandem):
        period = int(rng.integers(1, 40))
        n = int(rng.integers(200, 3000))
        unit = torch.from_numpy(rng.integers(0, 4, period, dtype=np.uint8)).to(dev)
        s = int(rng.integers(0, length - n))
        g[s:s + n] = unit.repeat(n // period + 1)[:n]
    for _ in range(homopolymers):
        n = int(rng.integers(30, 400))
        s = int(rng.integers(0, length - n))
        g[s:s + n] = int(rng.integers(0, 4))
    return g


def simulate_reads_torch(genome_len: int, genome_seed: int, coverage: float, seed: int = 42, device: str = "cuda",
                         mean_len: int = 15000, sd_len: int = 1500, err: float = 0.01, wrap: int = 40000,
                         batch_reads: int = 2048, min_len: int = 200, **genome_kw) -> SeqDB:
    """The same recipe as simulate_reads, vectorised with torch so that multi-Gbase sets (BASELINE configs[2], 150 Mb x
    30x) are generated in seconds on the GPU.  Seeded torch generators: deterministic for a given device type, but NOT
    the same stream as the numpy generator (the E. coli-size bench set stays on the numpy path)."""
    import torch
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    genome = make_genome_torch(genome_len, genome_seed, device, **genome_kw)
    wrap = min(wrap, genome_len)
    ext = torch.cat([genome, genome[:wrap]])
    L = ext.numel()
    n_reads = int(coverage * L / mean_len)
    gen.manual_seed(seed)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
    chunks, lens = [], []
    done = 0
    while done < n_reads:
        nb = min(batch_reads, n_reads - done)
        tl = (mean_len + sd_len * torch.randn(nb, device=dev, generator=gen)).to(torch.int64).clamp_(min_len, L)
        st = (torch.rand(nb, device=dev, generator=gen, dtype=torch.float64) * (L - tl + 1).to(torch.float64)).to(torch.int64)
        rc = torch.randint(0, 2, (nb,), device=dev, generator=gen).bool()
        tot = int(tl.sum())
        seg0 = torch.cumsum(tl, 0) - tl
        src = torch.repeat_interleave(st - seg0, tl) + torch.arange(tot, device=dev)
        base = ext[src]
        hit = torch.rand(tot, device=dev, generator=gen) < err
        kind = torch.randint(0, 9, (tot,), device=dev, generator=gen, dtype=torch.int8)
        kind = torch.where(hit, kind, torch.full_like(kind, -1))
        sub = (kind >= 0) & (kind < 4)
        base = torch.where(sub, kind.to(torch.uint8), base)
        emit = torch.ones(tot, dtype=torch.int64, device=dev)
        emit[kind == 4] = 0
        ins = kind >= 5
        emit[ins] = 2
        cs = torch.cumsum(emit, 0)
        opos = cs - emit
        seg_end = seg0 + tl - 1
        olen = cs[seg_end] - opos[seg0]
        ototal = int(cs[-1]) if tot else 0
        raw = torch.empty(ototal, dtype=torch.uint8, device=dev)
        keep = emit > 0
        raw[opos[keep]] = base[keep]
        raw[opos[ins] + 1] = (kind[ins] - 5).to(torch.uint8)
        oseg0 = torch.cumsum(olen, 0) - olen
        seg_start = torch.repeat_interleave(oseg0, olen)
        seg_len = torch.repeat_interleave(olen, olen)
        idx = torch.arange(ototal, device=dev)
        mirror = 2 * seg_start + seg_len - 1 - idx
        rc_rep = torch.repeat_interleave(rc, olen)
        codes = torch.where(rc_rep, comp[raw[mirror].long()], raw)
        one = torch.ones_like(codes)
        enc = torch.bitwise_left_shift(one, codes) | torch.bitwise_left_shift(torch.bitwise_right_shift(one * 8, codes[mirror]), 4)
        chunks.append(enc.cpu().numpy())
        lens.append(olen.to(torch.int32).cpu().numpy().astype(np.uint32))
        done += nb
    seqdb = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    rlen = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
    return SeqDB(seqdb, np.arange(len(rlen), dtype=np.uint32), rlen, roff, None)


def simulate_reads_resident_torch(genome_len: int, genome_seed: int, coverage: float, seed: int = 42, device: str = "cuda",
                                  mean_len: int = 15000, sd_len: int = 1500, err: float = 0.01, wrap: int = 40000,
                                  batch_reads: int = 16384, min_len: int = 200, progress=None, **genome_kw):
    """The recipe of simulate_reads_torch for sets that do not fit the host comfortably (BASELINE configs[3] at full size:
    3.1 Gb x 30x = 93 Gbases): every batch is encoded on the GPU STRAIGHT INTO one device buffer, which the library then adopts
    without a copy (ResidentDB.adopt_device).  Returns (uint8 device tensor of >= n_bases + 1024 elements, n_bases, rlen).
    (Batches of 16,384 reads: its own seeded stream, not simulate_reads_torch's.)"""
    import torch
    dev = torch.device(device)
    genome = make_genome_torch(genome_len, genome_seed, device, **genome_kw)
    wrap = min(wrap, genome_len)
    ext = torch.cat([genome, genome[:wrap]])
    del genome
    EL = ext.numel()
    n_reads = int(coverage * EL / mean_len)
    cap = int(n_reads * (mean_len + 60) * 1.002 + 8 * sd_len * n_reads ** 0.5) + 2048   # (+ 8 sigma of the length sum: small sets)
    seq = torch.empty(cap, dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
    lens, done, fill = [], 0, 0
    while done < n_reads:
        nb = min(batch_reads, n_reads - done)
        tl = (mean_len + sd_len * torch.randn(nb, device=dev, generator=gen)).to(torch.int64).clamp_(min_len, EL)
        st = (torch.rand(nb, device=dev, generator=gen, dtype=torch.float64) * (EL - tl + 1).to(torch.float64)).to(torch.int64)
        rc = torch.randint(0, 2, (nb,), device=dev, generator=gen).bool()
        tot = int(tl.sum())
        seg0 = torch.cumsum(tl, 0) - tl
        src = torch.repeat_interleave(st - seg0, tl) + torch.arange(tot, device=dev)
        base = ext[src]
        del src
        hit = torch.rand(tot, device=dev, generator=gen) < err
        kind = torch.randint(0, 9, (tot,), device=dev, generator=gen, dtype=torch.int8)
        kind = torch.where(hit, kind, torch.full_like(kind, -1))
        del hit
        base = torch.where((kind >= 0) & (kind < 4), kind.to(torch.uint8), base)
        emit = torch.ones(tot, dtype=torch.int64, device=dev)
        emit[kind == 4] = 0
        ins = kind >= 5
        emit[ins] = 2
        cs = torch.cumsum(emit, 0)
        opos = cs - emit
        olen = cs[seg0 + tl - 1] - opos[seg0]
        ototal = int(cs[-1])
        raw = torch.empty(ototal, dtype=torch.uint8, device=dev)
        keep = emit > 0
        raw[opos[keep]] = base[keep]
        raw[opos[ins] + 1] = (kind[ins] - 5).to(torch.uint8)
        del base, kind, emit, cs, opos, keep, ins
        oseg0 = torch.cumsum(olen, 0) - olen
        seg_start = torch.repeat_interleave(oseg0, olen)
        seg_len = torch.repeat_interleave(olen, olen)
        mirror = 2 * seg_start + seg_len - 1 - torch.arange(ototal, device=dev)
        del seg_start, seg_len
        codes = torch.where(torch.repeat_interleave(rc, olen), comp[raw[mirror].long()], raw)
        del raw
        one = torch.ones_like(codes)
        assert fill + ototal + 1024 <= cap
        seq[fill:fill + ototal] = torch.bitwise_left_shift(one, codes) | torch.bitwise_left_shift(torch.bitwise_right_shift(one * 8, codes[mirror]), 4)
        del codes, one, mirror
        lens.append(olen.to(torch.int32).cpu().numpy().astype(np.uint32))
        fill += ototal
        done += nb
        if progress:
            progress(done, n_reads)
    del ext
    seq[fill:fill + 1024] = 0
    torch.cuda.empty_cache()
    rlen = np.concatenate(lens)
    return seq, int(fill), rlen


def write_seqdb_from_device(prefix: str, seq, nbytes: int, rid, rlen, roff, piece: int = 1 << 30) -> None:
    """the files the stage executables read (formats.write_seqdb), from a seqdb that lives in a device tensor"""
    with open(prefix + ".seqdb", "wb") as f:
        for o in range(0, nbytes, piece):
            seq[o:min(nbytes, o + piece)].cpu().numpy().tofile(f)
    with open(prefix + ".idx", "w") as f:
        f.writelines("%09d r%09d %u %u\n" % (int(r), int(r), int(n), int(o)) for r, n, o in zip(rid, rlen, roff))


def seqdb_to_fasta(db: SeqDB, path: str) -> None:
    """Write the forward strand of every read as FASTA (for feeding the real shmr_mkseqdb in oracle tests)."""
    lut = np.full(16, ord("N"), np.uint8)
    lut[[1, 2, 4, 8]] = [ord(c) for c in "ACGT"]
    with open(path, "wb") as f:
        for nm, ln, off in zip(db.names, db.rlen, db.roff):
            f.write(b">" + nm.encode() + b"\n")
            f.write(lut[db.seqdb[int(off):int(off) + int(ln)] & 0x0F].tobytes() + b"\n")


# ---- the named workloads of BASELINE.json / SURVEY.md 8(d) -------------------------------------------------
_REPEATS = dict(repeat_families=20, repeat_len=6000, repeat_copies=300, divergence=0.01, tandem=3000, homopolymers=3000)
WORKLOADS = {
    "tiny": dict(genome_len=50_000, genome_seed=7, n_reads=160, mean_len=5000, sd_len=500, wrap=0),
    "small": dict(genome_len=1_000_000, genome_seed=1002, coverage=16.0),
    "ecoli": dict(genome_len=4_639_675, genome_seed=1001, n_reads=4984, n_files=8),          # C1 / C2 (configs[0]/[1])
    "c3": dict(genome_len=150_000_000, genome_seed=1003, coverage=30.0),                      # C3 (configs[2])
    # C4 / C5 (configs[3]/[4]) scaled to one GPU: CHM13 is not obtainable offline, so a 300 Mb genome seeded with 20 families of
    # 300 copies of a 6 kb unit at 1 % divergence (36 Mb of interspersed repeats), 3,000 tandem arrays and 3,000 homopolymer
    # runs, 30x.  c5s is the same read set indexed with -l 1 (dense L1 shimmers) under mc_upper 240.
    "c4s": dict(genome_len=300_000_000, genome_seed=1004, coverage=30.0, **_REPEATS),
    "c5s": dict(genome_len=300_000_000, genome_seed=1004, coverage=30.0, **_REPEATS),
    # C4 (configs[3]) at FULL size: 3.1 Gb with the same repeat content per Mb (207 families x 300 copies of a 6 kb unit, 31 k tandem
    # arrays, 31 k homopolymer runs) x 30x = 6.2 M reads, 93 Gbases -- generated into a device buffer (RESIDENT_WORKLOADS)
    "c4": dict(genome_len=3_100_000_000, genome_seed=1004, coverage=30.0, repeat_families=207, repeat_len=6000, repeat_copies=300,
               divergence=0.01, tandem=31000, homopolymers=31000),
    # C5 (configs[4]) at full size: the SAME read set as c4, indexed with -l 1 (dense L1 shimmers, ~3.2x the L2 list) under mc_upper 240
    "c5": dict(genome_len=3_100_000_000, genome_seed=1004, coverage=30.0, repeat_families=207, repeat_len=6000, repeat_copies=300,
               divergence=0.01, tandem=31000, homopolymers=31000),
    # the same recipes on a slice small enough for the CPU oracle (parity tests)
    "c4t": dict(genome_len=20_000_000, genome_seed=1004, coverage=30.0, repeat_families=4, repeat_len=6000, repeat_copies=300,
                divergence=0.01, tandem=200, homopolymers=200),
}
# stage parameters that differ from the defaults (k=16 w=80 r=6 l=2, bestn 4, mc 2..240, aln_bw 100, ovlp_upper 120)
STAGE_PARAMS = {"c5s": dict(levels=1, mc_upper=240), "c4s": dict(levels=2, mc_upper=240), "c4": dict(levels=2, mc_upper=240, chunks=8),
                # (24 chunks on ONE GPU: an l = 1 chunk of 8 would need twice the HBM the seqdb and its packs leave for a chunk's tables.  With 16
                #  chunks one pass ran in 23.2 s with 285 of 288 GB in use -- profiles/r04k_bench_c5_full_size_1step_nocpu.json -- and a second
                #  run, with the reference's 24 processes beside it, lost the box: not a configuration to leave as a default)
                "c5": dict(levels=1, mc_upper=240, chunks=24)}
TORCH_WORKLOADS = ("c3", "c4s", "c5s", "c4t")   # generated on the GPU (multi-Gbase sets in seconds instead of tens of minutes)


RESIDENT_WORKLOADS = ("c4", "c5")                   # generated straight into ONE device buffer the library adopts (no host copy)


def make_workload_resident(name: str, device: str = "cuda", progress=None, genome_mb: float | None = None):
    """genome_mb: the same recipe on a smaller genome, the repeat content scaled with the size (parity / multi-rank checks)"""
    cfg = dict(WORKLOADS[name])
    if genome_mb:
        f = genome_mb * 1e6 / cfg["genome_len"]
        cfg["genome_len"] = int(genome_mb * 1e6)
        for k in ("repeat_families", "tandem", "homopolymers"):
            cfg[k] = max(1, round(cfg[k] * f))
    return simulate_reads_resident_torch(cfg.pop("genome_len"), cfg.pop("genome_seed"), cfg.pop("coverage"), seed=42, device=device,
                                         progress=progress, **cfg)


def make_workload(name: str) -> SeqDB:
    cfg = dict(WORKLOADS[name])
    g = make_genome(cfg.pop("genome_len"), cfg.pop("genome_seed"))
    return simulate_reads(g, seed=42, **cfg)


def make_workload_torch(name: str, rank: int = 0, device: str = "cuda") -> SeqDB:
    """rank r of a weak-scaling job simulates its own genome (seed + 7919 r) with its own read stream (42 + r)"""
    cfg = dict(WORKLOADS[name])
    return simulate_reads_torch(cfg.pop("genome_len"), cfg.pop("genome_seed") + 7919 * rank, cfg.pop("coverage"), seed=42 + rank,
                                device=device, **cfg)
