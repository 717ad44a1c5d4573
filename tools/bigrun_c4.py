#!/usr/bin/env python3
"""BASELINE configs[3] at FULL size on ONE MI355X (VERDICT r2 "missing" #1): a 3.1 Gb genome with the C4 repeat content x 30x of
15 kb reads = ~93 Gbases, generated on the GPU batch by batch STRAIGHT INTO one resident seqdb buffer (adopted by the library
without a copy), then 8 index chunks and 8 overlap chunks one after the other, as pg_run.py does with index_nchunk = ovlp_nchunk = 8
(/root/reference/py/scripts/pg_run.py:232-244,305-317; the reference's own 30x human claim: README.md:15-16,160-166).

  python tools/bigrun_c4.py [genome_Mb=3100] [coverage=30] [chunks=8] [--levels 2] [--out profiles/r03_bigrun_c4.json]

Checks (properties; the whole set is far beyond the CPU oracle): chunks partition the reads; sampled reads' shimmers equal the
oracle's; sampled ovlp_t records re-derive bit-exactly from the oracle's ovlp_match and satisfy the acceptance rule; every read
pair appears once per chunk.  Reports bases/s indexed, records/s, unique pairs/s and the HBM high-water mark."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from peregrine_amd import _lib, formats, simreads
from peregrine_amd.shimmer import ResidentDB

ap = argparse.ArgumentParser()
ap.add_argument("genome_mb", nargs="?", type=float, default=3100)
ap.add_argument("coverage", nargs="?", type=float, default=30)
ap.add_argument("chunks", nargs="?", type=int, default=8)
ap.add_argument("--levels", type=int, default=2)
ap.add_argument("--mc-upper", type=int, default=240)
ap.add_argument("--batch-reads", type=int, default=16384)
ap.add_argument("--samples", type=int, default=24)
ap.add_argument("--out", default=None)
a = ap.parse_args()
dev = torch.device("cuda", 0)
total_hbm = torch.cuda.mem_get_info()[1]
low_free = [torch.cuda.mem_get_info()[0]]
def mark():
    low_free[0] = min(low_free[0], torch.cuda.mem_get_info()[0])

# ---- the genome: the C4 recipe (interspersed 6 kb x 300-copy families ~ 12 % of the genome, tandem arrays, homopolymers),
# scaled from the 300 Mb of c4s to genome_mb
t0 = time.perf_counter()
L = int(a.genome_mb * 1e6)
scale = L / 300e6
rep = dict(repeat_families=max(1, round(20 * scale)), repeat_len=6000, repeat_copies=300, divergence=0.01, tandem=round(3000 * scale), homopolymers=round(3000 * scale))
genome = simreads.make_genome_torch(L, 1004, "cuda", **rep)
print(f"genome {L/1e9:.2f} Gb with {rep} in {time.perf_counter()-t0:.1f} s", flush=True)

# ---- the reads, into one device buffer ------------------------------------------------------------------------------------------
mean_len, sd_len, err, min_len, wrap = 15000, 1500, 0.01, 200, 40000
ext = torch.cat([genome, genome[:wrap]]); del genome
EL = ext.numel()
n_reads = int(a.coverage * EL / mean_len)
cap = int(n_reads * (mean_len + 60) * 1.002) + 1024
seq = torch.empty(cap, dtype=torch.uint8, device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(42)
comp = torch.tensor([3, 2, 1, 0], dtype=torch.uint8, device=dev)
lens, done, fill = [], 0, 0
t0 = time.perf_counter()
while done < n_reads:
    nb = min(a.batch_reads, n_reads - done)
    tl = (mean_len + sd_len * torch.randn(nb, device=dev, generator=gen)).to(torch.int64).clamp_(min_len, EL)
    st = (torch.rand(nb, device=dev, generator=gen, dtype=torch.float64) * (EL - tl + 1).to(torch.float64)).to(torch.int64)
    rc = torch.randint(0, 2, (nb,), device=dev, generator=gen).bool()
    tot = int(tl.sum())
    seg0 = torch.cumsum(tl, 0) - tl
    src = torch.repeat_interleave(st - seg0, tl) + torch.arange(tot, device=dev)
    base = ext[src]; del src
    hit = torch.rand(tot, device=dev, generator=gen) < err
    kind = torch.randint(0, 9, (tot,), device=dev, generator=gen, dtype=torch.int8)
    kind = torch.where(hit, kind, torch.full_like(kind, -1)); del hit
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
    mirror = 2 * seg_start + seg_len - 1 - torch.arange(ototal, device=dev); del seg_start, seg_len
    codes = torch.where(torch.repeat_interleave(rc, olen), comp[raw[mirror].long()], raw); del raw
    one = torch.ones_like(codes)
    assert fill + ototal + 1024 <= cap
    seq[fill:fill + ototal] = torch.bitwise_left_shift(one, codes) | torch.bitwise_left_shift(torch.bitwise_right_shift(one * 8, codes[mirror]), 4)
    del codes, one, mirror
    lens.append(olen.to(torch.int32).cpu().numpy().astype(np.uint32))
    fill += ototal; done += nb
    mark()
del ext
torch.cuda.empty_cache()
rlen = np.concatenate(lens)
roff = np.concatenate([[0], np.cumsum(rlen.astype(np.uint64))[:-1]]).astype(np.uint64)
rid = np.arange(len(rlen), dtype=np.uint32)
n_bases = int(rlen.sum(dtype=np.uint64))
print(f"simulated {len(rlen)} reads, {n_bases/1e9:.2f} Gbases into a {cap/1e9:.1f} GB device buffer in {time.perf_counter()-t0:.1f} s", flush=True)
torch.cuda.synchronize()
rdb = ResidentDB.adopt_device(seq, n_bases, rid, rlen, roff, 0)   # no copy: the library reads the buffer where it is
mark()

def read_bytes(r):
    o, n = int(roff[r]), int(rlen[r])
    return seq[o:o + n].cpu().numpy()

# ---- index: N chunks one after the other -------------------------------------------------------------------------------------------
import oracle_util as U
N = a.chunks
rng = np.random.default_rng(17)
parts, t_index, sketch_ms = [], 0.0, 0.0
for c in range(1, N + 1):
    _lib.timing_reset()
    t = time.perf_counter(); p = rdb.index(total_chunk=N, mychunk=c, levels=a.levels); dt = time.perf_counter() - t
    t_index += dt; sketch_ms += _lib.timing("sketch")[0]
    parts.append(p); mark()
    r_ = (p.top["y"] >> np.uint64(32)).astype(np.int64)
    assert np.all(r_ % N == c % N) and np.all(np.diff(r_) >= 0)
    starts = np.searchsorted(r_, np.arange(len(rlen) + 1))
    mine = np.flatnonzero(rid % N == c % N)
    for r in rng.choice(mine, 3, replace=False):
        want = U.orc_sketch_seqdb(read_bytes(r), 80, 16, int(r))
        for _ in range(a.levels): want = U.orc_reduce(want, 6)
        assert np.array_equal(p.top[starts[r]:starts[r + 1]], want), (c, int(r))
    print(f"index chunk {c}/{N}: {dt*1e3:.0f} ms wall, {p.bases/1e9:.2f} Gbases, {len(p.top)} shimmers, literal reads {p.reads_literal}", flush=True)
mm = np.concatenate([p.top for p in parts]); mc = np.concatenate([p.top_mc for p in parts]); del parts
print(f"index: {n_bases/t_index/1e9:.1f} Gbases/s wall over {N} chunks ({t_index:.2f} s; sketch kernels {sketch_ms:.0f} ms = {n_bases/sketch_ms/1e6:.0f} Gbases/s); {len(mm)} shimmers, {len(mc)} count entries", flush=True)

# ---- overlap: N chunks one after the other, lists handed over on the device ----------------------------------------------------------
d_mm = torch.from_numpy(mm.view(np.uint8)).to(dev); d_mc = torch.from_numpy(mc.view(np.uint8)).to(dev)
torch.cuda.synchronize()
t_ovlp, total, uniq_keys, stats, align_ms, align_n = 0.0, 0, [], [], 0.0, 0
for c in range(1, N + 1):
    _lib.timing_reset()
    t = time.perf_counter()
    ov, st = rdb.overlap_dev(d_mm.data_ptr(), len(mm), d_mc.data_ptr(), len(mc), total_chunk=N, mychunk=c, mc_upper=a.mc_upper)
    dt = time.perf_counter() - t
    t_ovlp += dt; total += len(ov); mark()
    am = sum(_lib.timing(k)[0] for k in ("align", "align1")); an = sum(_lib.timing(k)[2] for k in ("align", "align1"))
    align_ms += am; align_n += an
    r0 = ov["y0"] >> np.uint64(32); r1 = ov["y1"] >> np.uint64(32)
    pair = (np.minimum(r0, r1) << np.uint64(32)) | np.maximum(r0, r1)
    assert len(np.unique(pair)) == len(pair), "a read pair twice within one chunk"
    uniq_keys.append(pair)
    for i in rng.choice(len(ov), min(a.samples, len(ov)), replace=False):   # sampled records against the oracle's ovlp_match + the acceptance rule
        o = ov[i]
        p0 = ((int(o["y0"]) & 0xFFFFFFFF) >> 1) + 1; p1 = ((int(o["y1"]) & 0xFFFFFFFF) >> 1) + 1
        q = read_bytes(int(r0[i]))[p0 - p1:]; tt = read_bytes(int(r1[i]))
        m = U.orc_ovlp_match(q, int(o["strand0"]), tt, int(o["strand1"]), 100)
        assert m == tuple(int(o[f]) for f in formats.MATCH_FIELDS), (c, int(i))
        assert m[2] < 48 and m[4] < 48 and (abs(len(q) - m[3]) < 48 or abs(len(tt) - m[5]) < 48) and m[3] > 500 and m[5] > 500
    stats.append({k: st[k] for k in ("n_records", "n_pair_records", "n_buckets", "n_align_needed", "n_align_gpu", "rounds", "device_replay")} | {"wall_s": dt})
    print(f"overlap chunk {c}/{N}: {dt:.2f} s, {len(ov)} records ({len(ov)/dt/1e6:.2f} M rec/s), {st['n_pair_records']} pair records, {st['n_buckets']} buckets, "
          f"{st['rounds']} sweeps, device replay {st['device_replay']}; alignment kernels {am:.0f} ms / {an} = {an/max(am,1e-9)/1e3:.1f} M aln/s", flush=True)
    del ov
uniq = int(len(np.unique(np.concatenate(uniq_keys))))
peak = total_hbm - low_free[0]
res = {
    "what": f"BASELINE configs[3] at full size on ONE MI355X: {a.genome_mb:.0f} Mb genome (C4 repeat recipe) x {a.coverage:g}x, {N} index + {N} overlap chunks run one after the other",
    "reads": int(len(rlen)), "bases": n_bases, "levels": a.levels, "chunks": N,
    "index_s": t_index, "bases_per_s_indexed": n_bases / t_index, "sketch_kernel_gbases_per_s": n_bases / sketch_ms / 1e6,
    "overlap_s": t_ovlp, "records": total, "records_per_s_overlap_stage": total / t_ovlp, "unique_pairs": uniq,
    "overlaps_per_s_index_plus_overlap": total / (t_index + t_ovlp), "unique_pairs_per_s": uniq / (t_index + t_ovlp),
    "alignments": int(align_n), "alignment_kernels_ms": align_ms, "alignments_per_s": align_n / max(align_ms, 1e-9) * 1e3,
    "hbm_total_bytes": int(total_hbm), "hbm_peak_used_bytes": int(peak), "seqdb_bytes": int(cap),
    "per_chunk": stats,
    "checks": "chunk lists partition the reads; 3 sampled reads per index chunk equal the oracle's shimmers; %d sampled records per overlap chunk re-derive "
              "bit-exactly from the oracle's ovlp_match and satisfy the acceptance rule; every read pair once per chunk" % a.samples,
}
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
