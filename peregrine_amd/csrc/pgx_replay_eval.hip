// pgx_replay_eval.hip -- the evaluation kernels of the device replay: shimmer_to_overlap (/root/reference/src/shmr_overlap.c:52-180) for the dirty
// buckets of a pass, in three shapes -- k_eval (16 lanes per bucket, the dense rounds), k_eval_rows (a wavefront per bucket, four rows per step, the
// sparse passes), k_eval_big (a workgroup per big bucket).  The fixed-point formulation and the schedule: pgx_replay.hip.
#include "pgx_replay.h"

namespace pgx {
namespace rp {

// ---- shimmer_to_overlap (shmr_overlap.c:52-180) for every dirty bucket in [lo, hi) -------------------------------------
// A group of GL lanes per bucket.  The rows (ai, descending) are sequential -- they communicate through the "contained"
// flags -- but the partners of one row are examined GL at a time, speculatively: lane l takes partner pbase + l, and the
// sequential semantics (stop once bestn overlaps are counted, or when the row's own read turns out contained) are then
// resolved with ballots.  What a lane beyond the stop did is harmless: a pair key, a reader registration (only ever costs a
// spurious re-evaluation), loads.  Buckets that hold a read twice can meet a pair twice within one evaluation: they run one
// partner at a time.  A single evaluation is a chain of dependent memory round trips, so the group form is what bounds
// the latency of a pass: rows x ~4 round trips instead of examinations x ~4.
// ---- shimmer_to_overlap (shmr_overlap.c:52-180) for every dirty bucket of the launch ------------------------------------
// A group of 16 lanes per bucket.  The rows (ai, descending) are sequential -- they communicate through the "contained"
// flags -- but the partners of one row are examined 16 at a time, speculatively: lane l takes partner pbase + l, and the
// sequential semantics (stop once bestn overlaps are counted, or when the row's own read turns out contained) are then
// resolved with ballots.  What a lane beyond the stop did is harmless: loads.  Buckets that hold a read twice can meet a
// pair twice within one evaluation: they run one partner at a time.  A single evaluation is a chain of dependent memory
// round trips, which is what bounds a sparse pass: the bucket's entries are staged in LDS once, a probe brings key and
// owner in one 16-byte load, and registrations / item stores are not waited for.
__global__ __launch_bounds__(256) void k_eval(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  __shared__ uint32_t s_rid[GPB][128], s_pos[GPB][128], s_rl[GPB][128];
  __shared__ uint8_t s_dir[GPB][128];
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1), gbase = lane & ~(GL - 1), gib = threadIdx.x / GL;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPW + (uint32_t)(lane / GL));
  const uint32_t j = (uint32_t)jj;
  bool alive = jj < hi && r.dirty[j] && !r.c->overflow;
  if (alive && (r.bflags[j] & F_BIG)) {   // a big bucket: left to k_eval_big, which runs behind this kernel from the list written here
    if (gl == 0) {
      const uint32_t at = atomicAdd(&r.c->nbig, 1u);
      if (at < LIST_CAP) r.blist[at] = j;   // (beyond the list: the bucket stays dirty and is listed again by the next pass)
    }
    alive = false;
  }
  {
    const uint64_t am = __ballot(alive);
    if (!am) return;
    if (lane == (int)__builtin_ctzll(am)) atomicAdd(&r.spread[(wave % SPREAD) * 8], (unsigned long long)(__popcll(am) / GL));
  }
  // reader-node arena: wave-uniform cursor; the wavefront that evaluates these buckets next time continues where this one stops
  const uint32_t wave_id = nlist ? r.wlist0 + wave : (uint32_t)(((uint64_t)lo + (uint64_t)wave * GPW) / GPW);
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;
  uint32_t icur = wc.z, iend = wc.w;   // item arena of this wavefront slot (multiples of 16), same idea
  uint32_t s0 = 0, n = 0;
  bool dup = false, first_eval = true;
  if (alive) {
    const uint32_t b = r.bid[j];
    s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    dup = (r.bflags[j] & F_DUP) != 0;
    first_eval = r.ever[j] == 0;
    for (uint32_t i = (uint32_t)gl; i < n; i += GL) {  // the bucket's entries -> LDS
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[gib][i] = rid, s_pos[gib][i] = (((uint32_t)y) >> 1) + 1, s_dir[gib][i] = r.dir[s0 + i], s_rl[gib][i] = r.rlen[rid];
    }
    if (gl == 0) {
      r.dirty[j] = 0;
      r.evaluated[j] = 1;
      r.ever[j] = 1;
      r.parity[j] ^= 1;
      r.ohead[j] = r.ihead[j];
    }
  }
  uint64_t clo = 0, chi = 0;  // "contained" flags of the bucket's entries (n <= 128)
  auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
  uint32_t head = NIL, num = 0, lookups = 0, skips = 0;
  uint32_t chunk = 0;  // base of the item chunk holding insertion ordinals [num & ~15, ...)
  bool any_guess = false, any_unfiled = false;
  int ai = (int)n - 1;  // (the first row opened is n - 2)
  bool row_open = false;
  uint32_t pbase = 0, got = 0, rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0;
  bool p_reg = false;  // this lane has a registration whose list position (p_idx) has not been looked at yet
  uint32_t p_idx = 0, p_slot = 0;
  auto resolve_pending = [&]() -> bool {  // false: the reader-node arena is exhausted
    if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
    const uint64_t rm = __ballot(p_reg);  // (what is left goes to the linked overflow)
    if (rm) {
      const uint32_t total = (uint32_t)__popcll(rm);
      if (rcur + total > rend) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if ((uint64_t)base + NCH > r.rn_cap) {
          atomicOr(&r.c->overflow, OV_NODES);
          return false;
        }
        rcur = base, rend = base + NCH;
      }
      if (p_reg) {
        const uint32_t node = rcur + lane_rank(rm);
        const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
        r.rn[node] = RNode{old, j};
      }
      rcur += total;
      p_reg = false;
    }
    return true;
  };
  for (;;) {
    if (alive && !row_open) {
      do --ai;
      while (ai >= 0 && cget((uint32_t)ai));
      if (ai < 0 || r.bestn == 0) {  // the bucket is done
        if (gl == 0) {
          r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
          r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
        }
        alive = false;
      } else {
        rid0 = s_rid[gib][ai], pos0 = s_pos[gib][ai], rlen0 = s_rl[gib][ai], dir0 = s_dir[gib][ai];
        got = 0, pbase = (uint32_t)ai + 1, row_open = true;
      }
    }
    if (!__ballot(alive)) {
      if (!resolve_pending()) return;
      if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
      break;
    }
    // ---- one batch of partners ----
    const uint32_t step = dup ? 1u : (uint32_t)GL;
    const uint32_t pi = pbase + (uint32_t)gl;
    bool valid = alive && (uint32_t)gl < step && pi < n && !cget(pi);
    uint32_t rid1 = 0, pos1 = 0;
    if (valid) {
      rid1 = s_rid[gib][pi], pos1 = s_pos[gib][pi];
      if (rid1 == rid0) valid = false;
    }
    uint32_t slot = NONE, v = 0;
    const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
    if (valid) slot = pair_find(r, pair, &v);
    const uint64_t vm = __ballot(valid);
    bool present = false, accepted = false, guessed = false;
    uint32_t ptype = 0, type = 0, mslot = NONE;
    if (valid) {
      present = v != 0 && own_bucket(v) < j;
      ptype = present ? own_type(v) : 0;
      if (!present && dup && slot != NONE)  // inserted earlier in THIS evaluation?
        for (uint32_t it = head; it != NIL; it = r.items[it - 1].next)
          if (r.items[it - 1].pslot == slot) {
            present = true, ptype = (r.items[it - 1].info >> 16) & 3;
            break;
          }
      if (!present) {
        const uint32_t rlen1 = s_rl[gib][pi], dir1 = s_dir[gib][pi];
        const uint32_t q_off = pos0 - pos1;
        if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
        uint32_t req = NONE;
        if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
        if (req < r.settled) {
          accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
        } else {
          accepted = true, guessed = true, type = T_OVERLAP;
          if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
        }
      }
    }
    if (!resolve_pending()) return;  // (the previous batch's registrations: their atomics have returned behind the loads above)
    // ---- the sequential semantics of the row over this batch, lowest partner first ----
    const uint64_t Vg = gbits(vm, gbase);
    const uint64_t P = gbits(__ballot(valid && present), gbase);
    const uint64_t PO = gbits(__ballot(valid && present && ptype == T_OVERLAP), gbase);
    const uint64_t A = gbits(__ballot(valid && !present && accepted), gbase);
    const uint64_t AO = gbits(__ballot(valid && !present && accepted && type == T_OVERLAP), gbase);
    const uint64_t AC = gbits(__ballot(valid && !present && accepted && type == T_CONTAINED), gbase);
    uint64_t AP = gbits(__ballot(valid && !present && accepted && type == T_CONTAINS), gbase);
    const uint64_t inc = PO | AO;
    int stop = GL;  // the last partner the sequential loop processes in this batch (GL: all of them, and the row goes on)
    if (alive && row_open) {
      const uint32_t need = r.bestn - got;  // >= 1
      if ((uint32_t)__popcll(inc) >= need) {
        uint64_t m = inc;
        for (uint32_t k = 1; k < need; ++k) m &= m - 1;
        stop = __builtin_ctzll(m);
      }
      if (AC) stop = min(stop, (int)__builtin_ctzll(AC));
    }
    const uint64_t proc = (2ULL << stop) - 1ULL;
    const uint64_t ins = A & proc;
    {  // the partners the sequential loop really examined register as readers of their pairs (the lists are only read by
       // k_update, after this kernel); a bucket listed by an earlier evaluation is not listed again.  The list position comes
       // from an atomic whose result is only looked at after the NEXT batch's loads have been issued (resolve_pending).
      bool reg = valid && ((proc >> gl) & 1);
      if (reg && slot == NONE) slot = pair_slot(r, pair);  // a pair the walk really examines gets its slot now
      if (reg && !first_eval) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(w);  // cnt, rhead, in[0], in[1]
        const uint32_t c = min(h1.x, NIN);
        if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
        for (uint32_t q = 2; q < c && reg; q += 8) {
          const uint4 a = *reinterpret_cast<const uint4 *>(w + 2 + q), b = *reinterpret_cast<const uint4 *>(w + 6 + q);
          const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k)
            if (q + k < c && x[k] == j + 1) reg = false;
        }
      }
      if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
    }
    // ---- the batch's insertions: the bucket's items fill 16-aligned chunks of 16 in insertion order (k_update walks a
    // bucket's list a chunk at a time, one lane per item) ----
    const bool my_ins = valid && !present && accepted && ((proc >> gl) & 1);
    const uint32_t cins = (uint32_t)__popcll(ins);
    const bool need_chunk = cins != 0 && (num == 0 || ((num + cins - 1) >> 4) != ((num - 1) >> 4));  // group-uniform
    uint32_t fresh = 0;
    {
      const uint64_t cm = __ballot(need_chunk && gl == 0);
      if (cm) {
        const uint32_t want = 16u * (uint32_t)__popcll(cm);
        if (icur + want > iend) {  // refill: one atomic on the shared counter per ICH items instead of one per chunk
          const uint32_t take = want > ICH ? want : ICH;
          uint32_t base = 0;
          if (lane == (int)__builtin_ctzll(cm)) base = atomicAdd(&r.c->item_top, take);
          base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(cm), 64);
          if ((uint64_t)base + take > r.item_cap) {
            atomicOr(&r.c->overflow, OV_ITEMS);
            return;
          }
          icur = base, iend = base + take;   // (what was left of the old piece, < want, is not used)
        }
        fresh = icur + 16u * (uint32_t)__popcll(cm & ((1ULL << gbase) - 1ULL));  // (gl == 0 lanes: one bit per group)
        icur += want;
      }
    }
    if (cins) {
      if (my_ins) {
        const uint32_t ord = num + (uint32_t)__popcll(ins & ((1ULL << gl) - 1ULL));  // insertion ordinal within the bucket
        const bool in_fresh = need_chunk && (num == 0 || (ord >> 4) != ((num - 1) >> 4));
        const uint32_t idx = (in_fresh ? fresh : chunk) + (ord & 15);
        uint32_t next;
        if (ord == 0) next = NIL;
        else if ((ord & 15) == 0) next = chunk + 16;  // the last item of the previous chunk, + 1
        else next = idx;                              // the item before this one, + 1
        uint32_t info = (uint32_t)ai | pi << 8 | type << 16;
        if (guessed) info |= I_GUESS;
        if (mslot == NONE) info |= I_UNFILED;
        r.items[idx] = Item{slot, info, mslot, next};
      }
      const uint32_t last = num + cins - 1;
      if (need_chunk && (num == 0 || (last >> 4) != ((num - 1) >> 4))) chunk = fresh;
      head = chunk + (last & 15) + 1;
      num += cins;
    }
    {
      const uint64_t g1 = gbits(__ballot(my_ins && guessed), gbase), g2 = gbits(__ballot(my_ins && mslot == NONE), gbase);
      any_guess |= g1 != 0, any_unfiled |= g2 != 0;
    }
    if (alive && row_open) {
      got += (uint32_t)__popcll(inc & proc);
      skips += (uint32_t)__popcll(P & proc);
      lookups += (uint32_t)__popcll(Vg & ~P & proc);
      AP &= proc;
      if (AP) {  // partners found contained: entry pbase + l
        if (pbase < 64) {
          clo |= AP << pbase;
          if (pbase) chi |= AP >> (64 - pbase);
        } else {
          chi |= AP << (pbase - 64);
        }
      }
      if (AC & proc) {
        if (ai < 64) clo |= 1ULL << ai;
        else chi |= 1ULL << (ai - 64);
      }
      if (stop < GL || pbase + step >= n) row_open = false;
      else pbase += step;
    }
  }
}

// ---- the same evaluation with FOUR ROWS PER STEP, for the sparse passes (a wavefront per bucket: lanes are plentiful there and
// a pass lasts as long as its longest bucket's chain of rows).  Lane l works row l / PW, partner l % PW; the rows are committed
// in order while each one is complete within its PW partners and no earlier row of the step set a contained flag (then the
// rows below are looked at again with the new flags); a row that needs more partners is continued alone, GLT per step.
template <int GLT, int PW>
__global__ __launch_bounds__(256) void k_eval_rows(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  constexpr uint32_t GPWT = 64 / GLT, GPBT = 256 / GLT;
  constexpr int SH = PW == 16 ? 4 : 2;  // log2(PW)
  constexpr uint64_t RM = (1ULL << PW) - 1ULL;  // one row's lanes
  __shared__ uint32_t s_rid[GPBT][128], s_pos[GPBT][128], s_rl[GPBT][128];
  __shared__ uint8_t s_dir[GPBT][128];
  const int lane = threadIdx.x & 63, gl = lane & (GLT - 1), gbase = lane & ~(GLT - 1), gib = threadIdx.x / GLT;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPWT + (uint32_t)(lane / GLT));
  const uint32_t j = (uint32_t)jj;
  bool alive = jj < hi && r.dirty[j] && !r.c->overflow;
  if (alive && (r.bflags[j] & F_BIG)) {   // a big bucket: left to k_eval_big, which runs behind this kernel from the list written here
    if (gl == 0) {
      const uint32_t at = atomicAdd(&r.c->nbig, 1u);
      if (at < LIST_CAP) r.blist[at] = j;   // (beyond the list: the bucket stays dirty and is listed again by the next pass)
    }
    alive = false;
  }
  {
    const uint64_t am = __ballot(alive);
    if (!am) return;
    if (lane == (int)__builtin_ctzll(am)) atomicAdd(&r.spread[(wave % SPREAD) * 8], (unsigned long long)(__popcll(am) / GLT));
  }
  // reader-node arena: wave-uniform cursor; the wavefront that evaluates these buckets next time continues where this one stops
  const uint32_t wave_id = nlist ? r.wlist0 + wave : (uint32_t)(((uint64_t)lo + (uint64_t)wave * GPWT) / GPWT);
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;
  uint32_t icur = wc.z, iend = wc.w;   // item arena of this wavefront slot (multiples of 16), same idea
  uint32_t s0 = 0, n = 0;
  bool dup = false, first_eval = true;
  if (alive) {
    const uint32_t b = r.bid[j];
    s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    dup = (r.bflags[j] & F_DUP) != 0;
    first_eval = r.ever[j] == 0;
    for (uint32_t i = (uint32_t)gl; i < n; i += GLT) {  // the bucket's entries -> LDS
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[gib][i] = rid, s_pos[gib][i] = (((uint32_t)y) >> 1) + 1, s_dir[gib][i] = r.dir[s0 + i], s_rl[gib][i] = r.rlen[rid];
    }
    if (gl == 0) {
      r.dirty[j] = 0;
      r.evaluated[j] = 1;
      r.ever[j] = 1;
      r.parity[j] ^= 1;
      r.ohead[j] = r.ihead[j];
    }
  }
  auto gbits = [&](uint64_t wave_mask, int) { return GLT == 64 ? wave_mask : ((wave_mask >> gbase) & ((1ULL << (GLT & 63)) - 1ULL)); };
  uint64_t clo = 0, chi = 0;  // "contained" flags of the bucket's entries (n <= 128)
  auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
  auto cset = [&](uint32_t i) {
    if (i < 64) clo |= 1ULL << i;
    else chi |= 1ULL << (i - 64);
  };
  uint32_t head = NIL, num = 0, lookups = 0, skips = 0;
  uint32_t chunk = 0;  // base of the item chunk holding insertion ordinals [num & ~15, ...)
  bool any_guess = false, any_unfiled = false;
  int done_to = (int)n - 1;  // rows >= done_to are finished (the first row is n - 2)
  bool row_open = false;     // a single row (cur_row) is in progress, sixteen partners per step from pbase
  int cur_row = 0, a0 = -1, a1 = -1, a2 = -1, a3 = -1, nrows = 0;
  uint32_t pbase = 0, got = 0;
  bool p_reg = false;  // this lane has a registration whose list position (p_idx) has not been looked at yet
  uint32_t p_idx = 0, p_slot = 0;
  auto resolve_pending = [&]() -> bool {  // false: the reader-node arena is exhausted
    if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
    const uint64_t rm = __ballot(p_reg);  // (what is left goes to the linked overflow)
    if (rm) {
      const uint32_t total = (uint32_t)__popcll(rm);
      if (rcur + total > rend) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
        base = (uint32_t)__shfl((int)base, 0, 64);
        if ((uint64_t)base + NCH > r.rn_cap) {
          atomicOr(&r.c->overflow, OV_NODES);
          return false;
        }
        rcur = base, rend = base + NCH;
      }
      if (p_reg) {
        const uint32_t node = rcur + lane_rank(rm);
        const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
        r.rn[node] = RNode{old, j};
      }
      rcur += total;
      p_reg = false;
    }
    return true;
  };
  for (;;) {
    if (alive && !row_open) {  // the next (up to four) rows that are not contained
      nrows = 0, a0 = a1 = a2 = a3 = -1;
      int x = done_to;
      while (nrows < 4) {
        do --x;
        while (x >= 0 && cget((uint32_t)x));
        if (x < 0) break;
        if (nrows == 0) a0 = x;
        else if (nrows == 1) a1 = x;
        else if (nrows == 2) a2 = x;
        else a3 = x;
        ++nrows;
      }
      if (nrows == 0 || r.bestn == 0) {  // the bucket is done
        if (gl == 0) {
          r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
          r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
        }
        alive = false;
      } else if (dup) {
        cur_row = a0, got = 0, pbase = (uint32_t)a0 + 1, row_open = true;
      }
    }
    if (!__ballot(alive)) {
      if (!resolve_pending()) return;
      if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
      break;
    }
    // ---- this step's (row, partner) of the lane ----
    const bool single = row_open;  // group-uniform
    const uint32_t step = dup ? 1u : (uint32_t)GLT;
    const int q = gl >> SH;
    const int myrow = single ? cur_row : (q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3);
    const uint32_t pi = single ? pbase + (uint32_t)gl : (uint32_t)(myrow + 1 + (gl & (PW - 1)));
    bool valid = alive && (single ? (uint32_t)gl < step : q < nrows) && pi < n && !cget(pi);
    uint32_t rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0, rid1 = 0, pos1 = 0;
    if (valid) {
      rid0 = s_rid[gib][myrow], pos0 = s_pos[gib][myrow], rlen0 = s_rl[gib][myrow], dir0 = s_dir[gib][myrow];
      rid1 = s_rid[gib][pi], pos1 = s_pos[gib][pi];
      if (rid1 == rid0) valid = false;
    }
    uint32_t slot = NONE, v = 0;
    const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
    if (valid) slot = pair_find(r, pair, &v);
    const uint64_t vm = __ballot(valid);
    bool present = false, accepted = false, guessed = false;
    uint32_t ptype = 0, type = 0, mslot = NONE;
    if (valid) {
      present = v != 0 && own_bucket(v) < j;
      ptype = present ? own_type(v) : 0;
      if (!present && dup && slot != NONE)  // inserted earlier in THIS evaluation?
        for (uint32_t it = head; it != NIL; it = r.items[it - 1].next)
          if (r.items[it - 1].pslot == slot) {
            present = true, ptype = (r.items[it - 1].info >> 16) & 3;
            break;
          }
      if (!present) {
        const uint32_t rlen1 = s_rl[gib][pi], dir1 = s_dir[gib][pi];
        const uint32_t q_off = pos0 - pos1;
        if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
        uint32_t req = NONE;
        if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
        if (req < r.settled) {
          accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
        } else {
          accepted = true, guessed = true, type = T_OVERLAP;
          if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
        }
      }
    }
    if (!resolve_pending()) return;  // (the previous step's registrations: their atomics have returned behind the loads above)
    // ---- the sequential semantics over this step, lowest lane first ----
    const uint64_t Vg = gbits(vm, gbase);
    const uint64_t P = gbits(__ballot(valid && present), gbase);
    const uint64_t PO = gbits(__ballot(valid && present && ptype == T_OVERLAP), gbase);
    const uint64_t A = gbits(__ballot(valid && !present && accepted), gbase);
    const uint64_t AO = gbits(__ballot(valid && !present && accepted && type == T_OVERLAP), gbase);
    const uint64_t AC = gbits(__ballot(valid && !present && accepted && type == T_CONTAINED), gbase);
    const uint64_t AP = gbits(__ballot(valid && !present && accepted && type == T_CONTAINS), gbase);
    const uint64_t inc = PO | AO;
    uint64_t proc = 0;  // the lanes the sequential walk really visits in this step
    if (alive && single) {
      int stop = GLT;  // the last partner the row processes in this step (GLT: all of them, and the row goes on)
      const uint32_t need = r.bestn - got;  // >= 1
      if ((uint32_t)__popcll(inc) >= need) {
        uint64_t m = inc;
        for (uint32_t k = 1; k < need; ++k) m &= m - 1;
        stop = __builtin_ctzll(m);
      }
      if (AC) stop = min(stop, (int)__builtin_ctzll(AC));
      proc = stop >= 63 ? ~0ULL : ((2ULL << stop) - 1ULL);
      got += (uint32_t)__popcll(inc & proc);
      for (uint64_t m = AP & proc; m; m &= m - 1) cset(pbase + (uint32_t)__builtin_ctzll(m));  // partners found contained
      if (AC & proc) cset((uint32_t)cur_row);
      if (stop < GLT || pbase + step >= n) row_open = false, done_to = cur_row;
      else pbase += step;
    } else if (alive) {
      int committed = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k >= nrows || committed != k) continue;
        const int row = k == 0 ? a0 : k == 1 ? a1 : k == 2 ? a2 : a3;
        // partners an EARLIER row of this step found contained (or that were such a row) are not examined by this row: their lanes are
        // dropped from its masks here (round 3 ended the step at the first row that changed a flag; a row's partners lie above it, so
        // the rows themselves are never flagged by an earlier row of the step)
        const uint32_t gone = (uint32_t)shr128(M128{clo, chi}, (uint32_t)row + 1).lo & (uint32_t)RM;
        const uint32_t inc_k = (uint32_t)((inc >> (PW * k)) & RM) & ~gone, ac_k = (uint32_t)((AC >> (PW * k)) & RM) & ~gone,
                       ap_k = (uint32_t)((AP >> (PW * k)) & RM) & ~gone;
        int stop = PW;
        if ((uint32_t)__popc(inc_k) >= r.bestn) {
          uint32_t m = inc_k;
          for (uint32_t t = 1; t < r.bestn; ++t) m &= m - 1;
          stop = __builtin_ctz(m);
        }
        if (ac_k) stop = min(stop, (int)__builtin_ctz(ac_k));
        if (stop == PW && (uint32_t)row + 1 + PW < n) continue;  // the row needs more partners: it is continued alone (committed stays k)
        const uint32_t proc_k = (stop < PW ? (2u << stop) - 1u : (uint32_t)RM) & ~gone;
        proc |= (uint64_t)proc_k << (PW * k);
        ++committed;
        for (uint32_t m = ap_k & proc_k; m; m &= m - 1) cset((uint32_t)row + 1 + (uint32_t)__builtin_ctz(m));   // partners found contained
        if (ac_k & proc_k) cset((uint32_t)row);
      }
      if (committed == 0) cur_row = a0, got = 0, pbase = (uint32_t)a0 + 1, row_open = true;  // (nothing done in this step)
      else done_to = committed == 1 ? a0 : committed == 2 ? a1 : committed == 3 ? a2 : a3;
    }
    skips += (uint32_t)__popcll(P & proc);
    lookups += (uint32_t)__popcll(Vg & ~P & proc);
    const uint64_t ins = A & proc;
    {  // the partners the walk really examined register as readers of their pairs (the lists are only read by k_update, after
       // this kernel); a bucket listed by an earlier evaluation is not listed again.  The list position comes from an atomic
       // whose result is only looked at after the NEXT step's loads have been issued (resolve_pending).
      bool reg = valid && ((proc >> gl) & 1);
      if (reg && slot == NONE) slot = pair_slot(r, pair);  // a pair the walk really examines gets its slot now
      if (reg && !first_eval) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
        const uint4 h1 = *reinterpret_cast<const uint4 *>(w);  // cnt, rhead, in[0], in[1]
        const uint32_t c = min(h1.x, NIN);
        if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
        for (uint32_t qq = 2; qq < c && reg; qq += 8) {
          const uint4 a = *reinterpret_cast<const uint4 *>(w + 2 + qq), b = *reinterpret_cast<const uint4 *>(w + 6 + qq);
          const uint32_t x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k)
            if (qq + k < c && x[k] == j + 1) reg = false;
        }
      }
      if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
    }
    // ---- the step's insertions: the bucket's items fill 16-aligned chunks of 16 in insertion order (lane order is the
    // sequential order; k_update walks a bucket's list a chunk at a time, one lane per item) ----
    const bool my_ins = valid && !present && accepted && ((proc >> gl) & 1);
    const uint32_t cins = (uint32_t)__popcll(ins);  // up to 64 in one step here: it may open several 16-item chunks
    static_assert(GLT == 64, "one bucket per wavefront: the chunk allocation below is wave-uniform");
    if (cins) {
      const uint32_t cur_no = num ? (num - 1) >> 4 : 0;                 // number of the chunk `chunk` (meaningless while num == 0)
      const uint32_t first_new = num ? cur_no + 1 : 0;                   // number of the first chunk this step has to open
      const uint32_t last = num + cins - 1, last_no = last >> 4;
      const uint32_t nnew = last_no + 1 > first_new ? last_no + 1 - first_new : 0;
      uint32_t fresh = 0;
      if (nnew) {
        const uint32_t want = 16u * nnew;
        if (icur + want > iend) {
          const uint32_t take = want > ICH ? want : ICH;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&r.c->item_top, take);
          base = (uint32_t)__shfl((int)base, 0, 64);
          if ((uint64_t)base + take > r.item_cap) {
            atomicOr(&r.c->overflow, OV_ITEMS);
            return;
          }
          icur = base, iend = base + take;
        }
        fresh = icur, icur += want;
      }
      auto base_of = [&](uint32_t no) { return no >= first_new ? fresh + 16u * (no - first_new) : chunk; };
      if (my_ins) {
        const uint32_t ord = num + (uint32_t)__popcll(ins & ((1ULL << gl) - 1ULL));  // insertion ordinal within the bucket
        const uint32_t idx = base_of(ord >> 4) + (ord & 15);
        uint32_t next;
        if (ord == 0) next = NIL;
        else if ((ord & 15) == 0) next = base_of((ord >> 4) - 1) + 16;  // the last item of the previous chunk, + 1
        else next = idx;                                                // the item before this one, + 1
        uint32_t info = (uint32_t)myrow | pi << 8 | type << 16;
        if (guessed) info |= I_GUESS;
        if (mslot == NONE) info |= I_UNFILED;
        r.items[idx] = Item{slot, info, mslot, next};
      }
      chunk = base_of(last_no);
      head = chunk + (last & 15) + 1;
      num += cins;
    }
    {
      const uint64_t g1 = gbits(__ballot(my_ins && guessed), gbase), g2 = gbits(__ballot(my_ins && mslot == NONE), gbase);
      any_guess |= g1 != 0, any_unfiled |= g2 != 0;
    }
  }
}
template __global__ void k_eval_rows<64, 16>(R r, uint32_t lo, uint32_t hi, uint32_t nlist);

__global__ __launch_bounds__(64 * BIG_NW) void k_eval_big(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  enum { MV = 0, MP, MPO, MA, MAO, MAC, MAP, MGU, MUF, MDF, NM };
  __shared__ uint32_t s_rid[128], s_pos[128], s_rl[128];
  __shared__ uint8_t s_dir[128];
  __shared__ uint64_t s_m[BIG_NW][NM];
  __shared__ uint32_t s_fresh, s_abort, s_bail;
  __shared__ unsigned long long s_setk[SET_CAP];      // pairs inserted by this evaluation (key + 1; 0: empty) ...
  __shared__ uint8_t s_sett[SET_CAP];                 // ... and their types
  __shared__ unsigned long long s_clk[CLAIM_CAP];     // this step's claims: pair (key + 1) ...
  __shared__ uint32_t s_clw[CLAIM_CAP];               // ... and the lowest walk index claiming it
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t wave_id = r.wbig0 + blockIdx.x * BIG_NW + (uint32_t)w;
  const uint4 wc = r.wcur[wave_id];
  uint32_t rcur = wc.x, rend = wc.y;   // reader-node arena of this wavefront
  uint32_t icur = wc.z, iend = wc.w;   // item arena of the workgroup (only thread 0's copy is used)
  if (threadIdx.x == 0) s_abort = 0;
  // its buckets: the big list of the pass.  Two kinds of entries: a bucket id, noted by a narrow kernel that ran over a RANGE and met the
  // bucket (k_eval / k_eval_rows skip big buckets); and a position in the dirty list | 2^31, noted by the count that made the list
  // (k_count_b) -- those count only in a list-mode launch, and only below `lo` = the number of list entries the narrow kernel and the
  // k_update of this pass cover (an evaluation k_update does not see would be lost)
  const uint32_t list_limit = nlist ? lo : 0u;
  const uint32_t nbig = min(r.c->nbig, LIST_CAP);
  for (uint32_t g = blockIdx.x; g < nbig; g += gridDim.x) {
  {
    const uint32_t e = r.blist[g];
    if ((e & 0x80000000u) && (e & 0x7FFFFFFFu) >= list_limit) continue;   // (workgroup-uniform)
    const uint32_t j = (e & 0x80000000u) ? (r.dlist[e & 0x7FFFFFFFu] & 0x7FFFFFFFu) : e;
    __syncthreads();   // (the previous bucket's LDS is done with)
    if (threadIdx.x == 0) {   // one thread decides for the workgroup (a flag another workgroup raises meanwhile must not split it)
      if (r.c->overflow) s_abort = 1;
      s_bail = 0, s_fresh = (j < hi && r.dirty[j]) ? 1u : 0u;   // (s_fresh doubles as "go": a listed bucket is dirty unless the list is stale)
    }
    __syncthreads();
    if (s_abort) break;
    if (!s_fresh) continue;
    const uint32_t b = r.bid[j], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
    const bool first_eval = r.ever[j] == 0;
    const bool dup = (r.bflags[j] & F_DUP) != 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t y = r.y0[s0 + i];
      const uint32_t rid = (uint32_t)(y >> 32);
      s_rid[i] = rid, s_pos[i] = (((uint32_t)y) >> 1) + 1, s_dir[i] = r.dir[s0 + i], s_rl[i] = r.rlen[rid];
    }
    if (dup) {
      for (uint32_t i = threadIdx.x; i < SET_CAP; i += blockDim.x) s_setk[i] = 0;
      for (uint32_t i = threadIdx.x; i < CLAIM_CAP; i += blockDim.x) s_clk[i] = 0, s_clw[i] = 0xFFFFFFFFu;
    }
    __syncthreads();   // (entries staged; everybody has read ever[] / dirty[] / bflags[] before thread 0 changes them)
    if (threadIdx.x == 0) {
      r.dirty[j] = 0, r.evaluated[j] = 1, r.ever[j] = 1, r.parity[j] ^= 1, r.ohead[j] = r.ihead[j];
      atomicAdd(&r.spread[(blockIdx.x % SPREAD) * 8], 1ULL);
    }
    // ---- workgroup-uniform state (every thread holds a copy and updates it identically) ----
    uint64_t clo = 0, chi = 0;   // "contained" flags of the bucket's entries
    auto cget = [&](uint32_t i) { return (((i < 64 ? clo : chi) >> (i & 63)) & 1) != 0; };
    auto cset = [&](uint32_t i) {
      if (i < 64) clo |= 1ULL << i;
      else chi |= 1ULL << (i - 64);
    };
    uint32_t head = NIL, num = 0, lookups = 0, skips = 0, chunk = 0;
    bool any_guess = false, any_unfiled = false;
    int done_to = (int)n - 1;   // rows >= done_to are finished (the first row is n - 2)
    bool row_open = false;       // one row (cur_row) is being continued alone, from partner pbase, with `got` overlaps counted so far
    int cur_row = 0;
    uint32_t pbase = 0, got = 0;
    bool p_reg = false;          // this lane has a registration whose list position (p_idx) has not been looked at yet
    uint32_t p_idx = 0, p_slot = 0;
    auto resolve_pending = [&]() {   // as in k_eval_rows; an exhausted arena raises s_abort instead of returning
      if (p_reg && p_idx < NIN) r.pc[p_slot >> r.cshift].in[p_idx] = j + 1, p_reg = false;
      const uint64_t rm = __ballot(p_reg);
      if (rm) {
        const uint32_t total = (uint32_t)__popcll(rm);
        if (rcur + total > rend) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&r.c->rnode_top, NCH);
          base = (uint32_t)__shfl((int)base, 0, 64);
          if ((uint64_t)base + NCH > r.rn_cap) {
            atomicOr(&r.c->overflow, OV_NODES);
            s_abort = 1;
            p_reg = false;
            return;
          }
          rcur = base, rend = base + NCH;
        }
        if (p_reg) {
          const uint32_t node = rcur + lane_rank(rm);
          const uint32_t old = atomicExch(&r.pc[p_slot >> r.cshift].rhead, node + 1);
          r.rn[node] = RNode{old, j};
        }
        rcur += total;
        p_reg = false;
      }
    };
#ifdef PGX_BIG_STATS
    uint32_t st_steps = 0, st_rows = 0, st_cut = 0, st_cont = 0, st_full = 0;
#endif
    for (;;) {
      // the rows of this step: the open row alone, or the next (up to four) rows that are not contained
      int a[BIG_NR], nrows = 0;
#pragma unroll
      for (int k = 0; k < BIG_NR; ++k) a[k] = -1;
      if (row_open) {
        a[0] = cur_row, nrows = 1;
      } else {
        for (int x = done_to; nrows < BIG_NR;) {
          do --x;
          while (x >= 0 && cget((uint32_t)x));
          if (x < 0) break;
          a[nrows++] = x;
        }
      }
      if (nrows == 0 || r.bestn == 0) break;
      // ---- this step's (row, partner) of the lane: slot q = wave / 2 takes row a[q], partners first + (wave % 2) * 64 + lane ----
      const int q = w >> 1, off = (w & 1) * 64 + lane;   // off: partner offset within the row's 128 (= its bit in the row's masks)
      const int myrow = a[q];
      const uint32_t first = row_open ? pbase : (uint32_t)(myrow + 1);
      const uint32_t pi = first + (uint32_t)off;
      const uint32_t wi = (uint32_t)(q * 128 + off);     // position in walk order
      bool valid = q < nrows && pi < n && !cget(pi);
      uint32_t rid0 = 0, pos0 = 0, rlen0 = 0, dir0 = 0, rid1 = 0, pos1 = 0;
      if (valid) {
        rid0 = s_rid[myrow], pos0 = s_pos[myrow], rlen0 = s_rl[myrow], dir0 = s_dir[myrow];
        rid1 = s_rid[pi], pos1 = s_pos[pi];
        if (rid1 == rid0) valid = false;
      }
      uint32_t slot = NONE, v = 0;
      const uint64_t pair = rid0 < rid1 ? ((uint64_t)rid0 << 32 | rid1) : ((uint64_t)rid1 << 32 | rid0);
      if (valid) slot = pair_find(r, pair, &v);
      bool present = false, accepted = false, guessed = false;
      uint32_t ptype = 0, type = 0, mslot = NONE;
      if (valid) {
        present = v != 0 && own_bucket(v) < j;
        ptype = present ? own_type(v) : 0;
        if (!present && dup) {   // inserted earlier in THIS evaluation?
          for (uint32_t i = (uint32_t)mix64(pair) & (SET_CAP - 1);; i = (i + 1) & (SET_CAP - 1)) {
            const unsigned long long kk = s_setk[i];
            if (kk == 0) break;
            if (kk == pair + 1) {
              present = true, ptype = s_sett[i];
              break;
            }
          }
        }
        if (!present) {
          const uint32_t rlen1 = s_rl[pi], dir1 = s_dir[pi];
          const uint32_t q_off = pos0 - pos1;
          if (q_off >= (1u << 30)) atomicOr(&r.c->overflow, OV_QOFF);
          uint32_t req = NONE;
          if (r.memo_used) mslot = memo_find(r, (unsigned long long)rid0 << 32 | rid1, q_off << 2 | dir0 << 1 | dir1, &req);
          if (req < r.settled) {
            accepted = classify(r.rq_res[req], rlen0, rlen1, q_off, &type);
          } else {
            accepted = true, guessed = true, type = T_OVERLAP;
            if (r.predict && predict_contained(rlen0, rlen1, q_off, r.predict, r.predict2)) type = rlen0 >= rlen1 ? T_CONTAINS : T_CONTAINED;
          }
        }
      }
      resolve_pending();   // (the previous step's registrations: their atomics have returned behind the loads above)
      const bool ins0 = valid && !present && accepted;
      uint32_t my_claim = NONE;
      if (dup) {   // would-be inserters claim their pair: the lowest walk index wins
        if (ins0) {
          for (uint32_t i = (uint32_t)(mix64(pair) >> 20) & (CLAIM_CAP - 1);; i = (i + 1) & (CLAIM_CAP - 1)) {
            unsigned long long kk = s_clk[i];
            if (kk == 0) kk = atomicCAS(&s_clk[i], 0ULL, (unsigned long long)pair + 1), kk = kk ? kk : pair + 1;
            if (kk == pair + 1) {
              atomicMin(&s_clw[i], wi);
              my_claim = i;
              break;
            }
          }
        }
      }
      {
        const uint64_t mv = __ballot(valid), mp = __ballot(valid && present), mpo = __ballot(valid && present && ptype == T_OVERLAP);
        const uint64_t ma = __ballot(ins0), mao = __ballot(ins0 && type == T_OVERLAP), mac = __ballot(ins0 && type == T_CONTAINED);
        const uint64_t map = __ballot(ins0 && type == T_CONTAINS), mgu = __ballot(ins0 && guessed), muf = __ballot(ins0 && mslot == NONE);
        if (lane == 0)
          s_m[w][MV] = mv, s_m[w][MP] = mp, s_m[w][MPO] = mpo, s_m[w][MA] = ma, s_m[w][MAO] = mao, s_m[w][MAC] = mac, s_m[w][MAP] = map,
          s_m[w][MGU] = mgu, s_m[w][MUF] = muf, s_m[w][MDF] = 0;
      }
      __syncthreads();
      if (dup) {   // a lane whose pair a LOWER lane of this step would insert is FLAGGED: its row is cut in front of it (below).  That
                   // holds for EVERY lane that found the pair absent, also one whose own alignment is rejected: sequentially it would
                   // have found the pair seen and skipped it.
        bool dflag = false;
        if (my_claim != NONE) {
          dflag = s_clw[my_claim] < wi;
        } else if (valid && !present) {
          for (uint32_t i = (uint32_t)(mix64(pair) >> 20) & (CLAIM_CAP - 1);; i = (i + 1) & (CLAIM_CAP - 1)) {
            const unsigned long long kk = s_clk[i];
            if (kk == 0) break;
            if (kk == pair + 1) {
              dflag = s_clw[i] < wi;
              break;
            }
          }
        }
        const uint64_t mdf = __ballot(dflag);
        if (lane == 0) s_m[w][MDF] = mdf;
        __syncthreads();
      }
      if (s_abort) break;
      // ---- the sequential semantics over this step: the rows in order, each over its 128 partners, lowest first (uniform) ----
      M128 proc[BIG_NR];
#pragma unroll
      for (int k = 0; k < BIG_NR; ++k) proc[k] = M128{0, 0};
      int committed = 0;         // rows completed in this step
      bool open_next = false;    // the row after them was cut: it is continued alone
      int open_row = 0;
      uint32_t open_pbase = 0, open_got = 0;
      for (int k = 0; k < nrows; ++k) {
        // A partner that an EARLIER row of this step found contained (or that was such a row) is not examined by this row: its lane is dropped
        // from the row's masks right here.  (Round 3 ended the step at the first row that changed a flag and looked at the rows below again
        // in the next one.  Rows themselves are never flagged by an earlier row of the step: a row's partners lie above it.)
        const uint32_t first_k = row_open ? pbase : (uint32_t)(a[k] + 1);
        const M128 gone = shr128(M128{clo, chi}, first_k);
        const M128 inc = andn128(M128{s_m[2 * k][MPO] | s_m[2 * k][MAO], s_m[2 * k + 1][MPO] | s_m[2 * k + 1][MAO]}, gone);
        const M128 ac = andn128(M128{s_m[2 * k][MAC], s_m[2 * k + 1][MAC]}, gone), ap = andn128(M128{s_m[2 * k][MAP], s_m[2 * k + 1][MAP]}, gone);
        // the row's cut: its first flagged lane.  (Round 3 took ONE cut for the step, the lowest flagged lane of all rows -- which most often lay
        // beyond the stop of its row, among lanes the walk never visits, and still ended the step there: 69 % of all steps ended with rows left,
        // 1.5 of 4 rows committed per step, profiles/r04w_big_stats_c4s.txt.  A flag whose lower claimant turns out unvisited is void but harmless:
        // the lane is looked at again in the next step.)
        const M128 df = andn128(M128{s_m[2 * k][MDF], s_m[2 * k + 1][MDF]}, gone);
        const int cut = any128(df) ? ctz128(df) : 128;   // first offset of the row that may not be processed
        if (cut == 0) break;                             // the cut lies in front of this row
        const uint32_t need = r.bestn - (row_open ? got : 0u);   // >= 1
        int stop = 128;
        if ((uint32_t)popc128(inc) >= need) stop = nth128(inc, need);
        if (any128(ac)) stop = min(stop, ctz128(ac));
        const bool complete = stop < cut || (cut == 128);   // the row ends before the cut (or there is none in it)
        proc[k] = andn128(complete ? upto128(stop) : upto128(cut - 1), gone);
        const M128 apk = and128(ap, proc[k]);
        for (uint64_t m = apk.lo; m; m &= m - 1) cset(first_k + (uint32_t)__builtin_ctzll(m));   // partners found contained
        for (uint64_t m = apk.hi; m; m &= m - 1) cset(first_k + 64u + (uint32_t)__builtin_ctzll(m));
        const bool rowc = any128(and128(ac, proc[k]));
        if (rowc) cset((uint32_t)a[k]);
        if (!complete) {
          open_next = true, open_row = a[k], open_pbase = first_k + (uint32_t)cut, open_got = (row_open ? got : 0u) + (uint32_t)popc128(and128(inc, proc[k]));
          break;
        }
        ++committed;
      }
#ifdef PGX_BIG_STATS
      ++st_steps, st_rows += (uint32_t)committed, st_cut += open_next ? 1u : 0u, st_full += (committed == nrows) ? 1u : 0u;
      st_cont += (!open_next && committed < nrows) ? 1u : 0u;
#endif
      if (committed) done_to = a[committed - 1];
      row_open = open_next;
      if (open_next) cur_row = open_row, pbase = open_pbase, got = open_got;
      // per wavefront: what the walk really visits, and the insertions in walk order (row, then partner)
      uint32_t cins = 0, before = 0;
      uint64_t myproc = 0;
      for (int ww = 0; ww < BIG_NW; ++ww) {
        const int k = ww >> 1;
        const uint64_t pw = (ww & 1) ? proc[k].hi : proc[k].lo;
        const uint32_t c = (uint32_t)__popcll(s_m[ww][MA] & pw);
        if (ww < w) before += c;
        if (ww == w) myproc = pw;
        cins += c;
        skips += (uint32_t)__popcll(s_m[ww][MP] & pw);
        lookups += (uint32_t)__popcll(s_m[ww][MV] & ~s_m[ww][MP] & pw);
        any_guess |= (s_m[ww][MGU] & pw) != 0, any_unfiled |= (s_m[ww][MUF] & pw) != 0;
      }
      // the item chunks this step opens (one allocation for the workgroup, by thread 0)
      const uint32_t cur_no = num ? (num - 1) >> 4 : 0, first_new = num ? cur_no + 1 : 0;
      const uint32_t last = num + cins - 1, last_no = last >> 4;
      const uint32_t nnew = cins && last_no + 1 > first_new ? last_no + 1 - first_new : 0;
      if (threadIdx.x == 0 && nnew) {
        const uint32_t want = 16u * nnew;
        if (icur + want > iend) {
          const uint32_t take = want > ICH ? want : ICH;
          const uint32_t base = atomicAdd(&r.c->item_top, take);
          if ((uint64_t)base + take > r.item_cap) atomicOr(&r.c->overflow, OV_ITEMS), s_abort = 1;
          icur = base, iend = base + take;
        }
        s_fresh = icur, icur += want;
      }
      const bool visited = valid && ((myproc >> lane) & 1);
      {  // the partners the walk really examined register as readers of their pairs (as in k_eval_rows)
        bool reg = visited;
        if (reg && slot == NONE) slot = pair_slot(r, pair);
        if (reg && !first_eval) {
          const uint32_t *pw = reinterpret_cast<const uint32_t *>(&r.pc[slot >> r.cshift]);
          const uint4 h1 = *reinterpret_cast<const uint4 *>(pw);  // cnt, rhead, in[0], in[1]
          const uint32_t c = min(h1.x, NIN);
          if ((c > 0 && h1.z == j + 1) || (c > 1 && h1.w == j + 1)) reg = false;
          for (uint32_t qq = 2; qq < c && reg; qq += 8) {
            const uint4 xa = *reinterpret_cast<const uint4 *>(pw + 2 + qq), xb = *reinterpret_cast<const uint4 *>(pw + 6 + qq);
            const uint32_t x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
              if (qq + k < c && x[k] == j + 1) reg = false;
          }
        }
        if (reg) p_idx = atomicAdd(&r.pc[slot >> r.cshift].cnt, 1u), p_slot = slot, p_reg = true;
      }
      const bool my_ins = ins0 && visited;
      if (dup) {
        if (my_claim != NONE) s_clk[my_claim] = 0, s_clw[my_claim] = 0xFFFFFFFFu;   // (the claim table is empty again for the next step)
        if (my_ins) {   // this evaluation's insertions, for the probes of the steps to come
          uint32_t tries = 0;
          for (uint32_t i = (uint32_t)mix64(pair) & (SET_CAP - 1);; i = (i + 1) & (SET_CAP - 1)) {
            const unsigned long long kk = atomicCAS(&s_setk[i], 0ULL, (unsigned long long)pair + 1);
            if (kk == 0 || kk == pair + 1) {
              s_sett[i] = (uint8_t)type;
              break;
            }
            if (++tries >= SET_CAP) {   // the set is full (not with <= 128 entries and bestn <= ~8; kept as a guard): leave the bucket to k_eval_rows
              s_bail = 1;
              break;
            }
          }
        }
      }
      __syncthreads();   // (s_fresh is there; nobody reads this step's masks any more; the set holds this step's insertions)
      if (s_abort) break;
      if (cins) {
        const uint32_t fresh = s_fresh;
        auto base_of = [&](uint32_t no) { return no >= first_new ? fresh + 16u * (no - first_new) : chunk; };
        if (my_ins) {
          const uint32_t ord = num + before + (uint32_t)__popcll(s_m[w][MA] & myproc & ((1ULL << lane) - 1ULL));   // insertion ordinal within the bucket
          const uint32_t idx = base_of(ord >> 4) + (ord & 15);
          uint32_t next;
          if (ord == 0) next = NIL;
          else if ((ord & 15) == 0) next = base_of((ord >> 4) - 1) + 16;  // the last item of the previous chunk, + 1
          else next = idx;                                                // the item before this one, + 1
          uint32_t info = (uint32_t)myrow | pi << 8 | type << 16;
          if (guessed) info |= I_GUESS;
          if (mslot == NONE) info |= I_UNFILED;
          r.items[idx] = Item{slot, info, mslot, next};
        }
        chunk = base_of(last_no);
        head = chunk + (last & 15) + 1;
        num += cins;
      }
      __syncthreads();   // (s_m[w][MA] was read above: the next step may overwrite the masks now)
      if (s_bail) break;
    }
    resolve_pending();
#ifdef PGX_BIG_STATS
    if (threadIdx.x == 0) {   // [3] steps, [4] rows committed, [5] steps cut at a duplicate, [6] steps ended by a containment, [7] steps that committed all their rows
      unsigned long long *line = r.spread + (blockIdx.x % SPREAD) * 8;
      atomicAdd(line + 3, (unsigned long long)st_steps), atomicAdd(line + 4, (unsigned long long)st_rows), atomicAdd(line + 5, (unsigned long long)st_cut);
      atomicAdd(line + 6, (unsigned long long)st_cont), atomicAdd(line + 7, (unsigned long long)st_full);
      atomicMax(&r.c->big_max_steps, st_steps);
      if (st_steps >= 64) atomicAdd(&r.c->big_long, 1u), atomicAdd(&r.c->big_long_n, n);
      atomicAdd(&r.c->big_evals, 1u);
    }
#endif
    if (threadIdx.x == 0) {
      if (s_bail) {   // (guard path: evaluated again by k_eval_rows, one partner at a time; the lists written so far are simply dropped)
        r.dirty[j] = 1, r.evaluated[j] = 0, r.parity[j] ^= 1;
        r.bflags[j] = (uint8_t)((r.bflags[j] & ~F_BIG));
      } else {
        r.ihead[j] = head, r.inum[j] = num, r.lookups[j] = lookups, r.skips[j] = skips;
        r.bflags[j] = (uint8_t)(F_BIG | (dup ? F_DUP : 0) | (any_guess ? F_GUESS : 0) | (any_unfiled ? F_UNFILED : 0));
      }
    }
  }
  if (s_abort) break;
  }
  if (lane == 0) r.wcur[wave_id] = make_uint4(rcur, rend, icur, iend);
}

}  // namespace rp
}  // namespace pgx
