"""A numpy stand-in for the GPU stages of the multi-GPU exchange (peregrine_amd/parallel.py), TEST INFRASTRUCTURE: it restates
what libpgx's pgx_pairs_prepare_dev / pgx_pairs_scatter_dev compute (the record generation of build_map,
/root/reference/src/shmr_utils.c:295-404, split per index chunk) so that the exchange protocol can run under gloo in a
container without a GPU.  The records a rank receives are checked against the oracle's own build_map (orc_pair_records)."""
import numpy as np
import torch

from oracle_util import PAIR_REC_DTYPE
from peregrine_amd.formats import MC_DTYPE, MM_DTYPE


def _bytes(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy())


class NumpyEngine:
    def __init__(self, rlen_by_rid):
        self.rlen = np.asarray(rlen_by_rid, np.uint32)

    def pairs_prepare(self, top: torch.Tensor, counts_all: torch.Tensor, lower: int, upper: int) -> int:
        self.mm = top.numpy().view(MM_DTYPE)
        mc = counts_all.numpy().view(MC_DTYPE)
        mer, inv = np.unique(mc["mer"], return_inverse=True)
        cnt = np.zeros(len(mer), np.uint64)
        np.add.at(cnt, inv, mc["count"].astype(np.uint64))            # aggregate_mm_count, shmr_utils.c:162-176
        h = self.mm["x"] >> np.uint64(8)
        pos = np.searchsorted(mer, h)
        assert np.all(pos < len(mer)) and np.all(mer[pos] == h), "hash missing from the count tables"
        c = cnt[pos]
        self.keep = (c >= lower) & (c <= upper)                        # :327 inclusive
        strict = np.flatnonzero((c >= lower) & (c < upper))            # :311-320 strict
        return int(strict[0]) if len(strict) else -1

    def pairs_scatter(self, world: int, start: int):
        mm = self.mm
        counts = [0] * world
        if start < 0 or start >= len(mm):
            return torch.empty(0, dtype=torch.uint8), counts
        K = np.flatnonzero(self.keep & (np.arange(len(mm)) >= start))
        a, b = mm[K[:-1]], mm[K[1:]]
        same = (a["y"] >> np.uint64(32)) == (b["y"] >> np.uint64(32))
        gap = ((b["y"] >> np.uint64(1)) & np.uint64(0xFFFFFFF)).astype(np.uint32) - ((a["y"] >> np.uint64(1)) & np.uint64(0xFFFFFFF)).astype(np.uint32)
        ok = same & (gap >= 100)
        a, b = a[ok], b[ok]

        def flip(y, x):                                                 # shmr_utils.c:376-396
            span = (x & np.uint64(0xFF)).astype(np.uint32)
            rid = (y >> np.uint64(32)).astype(np.int64)
            p = ((y & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.uint32) + np.uint32(1)
            rpos = self.rlen[rid] - p + span - np.uint32(1)
            return ((y & np.uint64(0xFFFFFFFF00000001)) | (rpos.astype(np.uint64) << np.uint64(1))) ^ np.uint64(1)

        rec = np.zeros(2 * len(a), PAIR_REC_DTYPE)                      # forward before reverse for the same adjacent pair
        rec["key0"][0::2], rec["key1"][0::2], rec["y0"][0::2], rec["dir"][0::2] = a["x"], b["x"], a["y"], 0
        rec["key0"][1::2], rec["key1"][1::2], rec["y0"][1::2], rec["dir"][1::2] = b["x"], a["x"], flip(b["y"], b["x"]), 1
        rec["npos"] = ~((rec["y0"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.uint32)
        v = ((rec["key0"] >> np.uint64(8)) % np.uint64(world)).astype(np.int64)     # owner: chunk c with c % T == v
        rank_of = (v - 1) % world                                                    # chunk c runs on rank c-1
        order = np.argsort(rank_of, kind="stable")
        rec = rec[order]
        counts = np.bincount(rank_of, minlength=world).tolist()
        return _bytes(rec), counts

    def overlap_records(self, recv: torch.Tensor, world: int, chunk: int, **params):
        return recv.numpy().view(PAIR_REC_DTYPE).copy()
