// pgx_replay_tables.hip -- everything of the device replay that is not an evaluation: bucket flags and arena slots (k_setup, k_init_slots), applying
// the evaluated buckets' lists to the pair table (k_update), the dirty lists (k_count_a / _b), filing the alignments the lists need (k_file), checking
// the guesses (k_settle), writing the records (k_emit).  The fixed-point formulation and the schedule: pgx_replay.hip.
#include "pgx_replay.h"

namespace pgx {
namespace rp {

// ---- flags: buckets holding a read more than once (only those can meet a pair twice within one evaluation) ----------
__global__ __launch_bounds__(256) void k_setup(R r, uint32_t *hist) {   // hist (trace only): [dup][min(n / 8, 15)] bucket counts
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= r.nb) return;
  const uint32_t b = r.bid[j], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
  bool dup = false;
  for (uint32_t i = 0; i + 1 < n && !dup; ++i) {
    const uint32_t ri = (uint32_t)(r.y0[s0 + i] >> 32);
    for (uint32_t k = i + 1; k < n; ++k)
      if ((uint32_t)(r.y0[s0 + k] >> 32) == ri) {
        dup = true;
        break;
      }
  }
  const bool big = dup ? r.dup_min && n >= r.dup_min : r.big_min && n >= r.big_min;
  r.bflags[j] = (uint8_t)((dup ? F_DUP : 0) | (big ? F_BIG : 0));
  if (big) atomicAdd(&r.c->nbig_total, 1u);
  r.dirty[j] = 1;
  if (hist) atomicAdd(&hist[(dup ? 16 : 0) + min(n / 8, 15u)], 1u);
}

// every wavefront slot of k_eval starts with its own piece of the item arena (no atomic at all for its first ICH items); the
// shared counter starts behind the pieces
// (slots [0, n_dense): the dense rounds' wavefronts, GPW buckets each; [wlist0, nslots): the list-mode wavefronts; the slots
// between them are only used by the one-bucket-per-wavefront dense variant and start empty)
__global__ void k_init_slots(uint4 *__restrict__ wcur, uint32_t nslots, uint32_t n_dense, uint32_t wlist0, Counters *c) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w == 0) c->item_top = (n_dense + (nslots - wlist0)) * ICH;
  if (w >= nslots) return;
  if (w < n_dense || w >= wlist0) {
    const uint32_t piece = w < n_dense ? w : n_dense + (w - wlist0);
    wcur[w] = make_uint4(0u, 0u, piece * ICH, (piece + 1) * ICH);
  } else {
    wcur[w] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ---- apply the evaluated buckets' lists to the pair table: a group per bucket, a lane per item ---------------------------
__device__ __forceinline__ void apply_insertion(const R &r, uint32_t j, uint32_t pnew, const Item &im) {
  const uint32_t slot = im.pslot, type = (im.info >> 16) & 3;
  const uint32_t mine = own_enc(j, pnew, type);
  uint32_t v = r.ph[slot].own;
  for (;;) {
    if (v != 0 && own_bucket(v) < j) {  // an earlier bucket got in first: this evaluation is stale
      r.dirty[j] = 1;
      return;
    }
    const uint32_t prev = atomicCAS(&r.ph[slot].own, v, mine);
    if (prev == v) {
      if (v == 0 || own_bucket(v) > j) mark_readers(r, slot, j);  // absent -> present, or a later owner displaced (it reads the pair too)
      else if ((own_type(v) == T_OVERLAP) != (type == T_OVERLAP)) mark_readers(r, slot, j);  // ours before: readers see the type class
      return;
    }
    v = prev;
  }
}
__global__ __launch_bounds__(256) void k_update(R r, uint32_t lo, uint32_t hi, uint32_t nlist) {
  const int lane = threadIdx.x & 63, gl = lane & (GL - 1);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t jj = bucket_of_group(r, lo, hi, nlist, wave * GPW + (uint32_t)(lane / GL)) & 0x7FFFFFFFull;   // (a listed big bucket: k_count_b's mark off)
  const uint32_t j = (uint32_t)jj;
  const bool alive = jj < hi && r.evaluated[j] && !r.c->overflow;
  if (blockIdx.x == 0 && threadIdx.x == 0) r.c->nbig = 0;   // (k_eval_big has consumed the pass's list; the next pass starts a new one)
  // (the dirty statistics are NOT touched here: the list-mode blocks of this very launch read ndirty as their list length, and
  // the count that follows writes absolute values)
  if (!__ballot(alive)) return;
  const uint32_t pnew = alive ? r.parity[j] : 0, pold = pnew ^ 1;
  // what this evaluation inserts: take or refresh ownership (lowest bucket wins)
  uint32_t cur = alive ? r.ihead[j] : NIL;
  while (__ballot(cur != NIL)) {
    if (cur != NIL) {
      const uint32_t last = cur - 1, base = last & ~15u, cnt = (last & 15) + 1;
      const Item first = r.items[base];
      for (uint32_t o = (uint32_t)gl; o < cnt; o += GL) apply_insertion(r, j, pnew, o == 0 ? first : r.items[base + o]);
      cur = first.next;
    }
  }
  // what the previous evaluation inserted and this one did not refresh: withdraw
  cur = alive ? r.ohead[j] : NIL;
  while (__ballot(cur != NIL)) {
    if (cur != NIL) {
      const uint32_t last = cur - 1, base = last & ~15u, cnt = (last & 15) + 1;
      const uint32_t nxt = r.items[base].next;
      for (uint32_t o = (uint32_t)gl; o < cnt; o += GL) {
        const uint32_t slot = r.items[base + o].pslot;
        const uint32_t v = r.ph[slot].own;
        if (v != 0 && own_bucket(v) == j && own_parity(v) == pold)
          if (atomicCAS(&r.ph[slot].own, v, 0u) == v) mark_readers(r, slot, j);
      }
      cur = nxt;
    }
  }
  if (alive && gl == 0) r.evaluated[j] = 0, r.ohead[j] = NIL;
}

// ---- the dirty buckets: how many, in which range, and (while they fit) their list, LOWEST FIRST -------------------------
// Two small launches: blocks of CB buckets count theirs (16 flags per lane, one 16-byte load), then every block adds up the
// counts of the blocks before it and writes its ids at that offset.  The list is exactly ascending, so when more than LIST_CAP
// buckets are dirty the list keeps the LOWEST ones (evaluating those first wastes the fewest evaluations -- the round-1 form
// took list positions by atomics in arrival order, and atomics on one address cost ~12 ns each: with every block holding a
// dirty bucket a count took 0.46 ms, ten times per step).
__device__ __forceinline__ uint32_t dirty16(const R &r, uint32_t j0, uint32_t end) {  // bit i: bucket j0 + i < end is dirty (j0 a multiple of 16)
  if (j0 >= end) return 0;
  uint32_t m = 0;
  if (j0 + 16 <= end) {
    const uint4 v = *reinterpret_cast<const uint4 *>(r.dirty + j0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) m |= ((w[q] >> (8 * b)) & 0xFFu) ? 1u << (4 * q + b) : 0u;
  } else {
    for (uint32_t i = 0; j0 + i < end; ++i) m |= r.dirty[j0 + i] ? 1u << i : 0u;
  }
  return m;
}
__global__ __launch_bounds__(256) void k_count_a(R r, uint32_t rlo, uint32_t rhi, uint32_t *__restrict__ blk) {  // blk[3 b + {0, 1, 2}] = count, lowest, highest
  __shared__ uint32_t s_c[4], s_lo[4], s_hi[4];
  const uint32_t j0 = rlo + blockIdx.x * CB + threadIdx.x * 16;   // (rlo: a multiple of 16)
  const uint32_t m = dirty16(r, j0, rhi);
  uint32_t c = (uint32_t)__popc(m), lo = m ? j0 + (uint32_t)__builtin_ctz(m) : 0xFFFFFFFFu, hi = m ? j0 + 31u - (uint32_t)__builtin_clz(m) : 0u;
  for (int o = 32; o; o >>= 1) {
    c += (uint32_t)__shfl_xor((int)c, o, 64);
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_c[w] = c, s_lo[w] = lo, s_hi[w] = hi;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) r.c->nbig = 0;   // (k_count_b lists the big buckets by their POSITION in the list it writes: positions of an older list must be gone)
    blk[3 * blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    blk[3 * blockIdx.x + 1] = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
    blk[3 * blockIdx.x + 2] = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
  }
}
__global__ __launch_bounds__(256) void k_count_b(R r, uint32_t rlo, uint32_t rhi, const uint32_t *__restrict__ blk, uint32_t nblk) {
  __shared__ uint32_t s_part[4], s_lo[4], s_hi[4], s_tot[4], s_w[4];
  // the counts of the blocks before this one (and, in block 0, the totals of all of them)
  uint32_t before = 0, total = 0, lo = 0xFFFFFFFFu, hi = 0;
  const bool totals = blockIdx.x == 0;
  for (uint32_t b = threadIdx.x; b < nblk; b += 256) {
    const uint32_t c = blk[3 * b];
    if (b < blockIdx.x) before += c;
    if (totals) total += c, lo = min(lo, blk[3 * b + 1]), hi = max(hi, blk[3 * b + 2]);
  }
  for (int o = 32; o; o >>= 1) {
    before += (uint32_t)__shfl_xor((int)before, o, 64);
    if (totals) {
      total += (uint32_t)__shfl_xor((int)total, o, 64);
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, 64));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, 64));
    }
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) s_part[w] = before, s_tot[w] = total, s_lo[w] = lo, s_hi[w] = hi;
  const uint32_t j0 = rlo + blockIdx.x * CB + threadIdx.x * 16;   // (rlo: a multiple of 16)
  const uint32_t m = dirty16(r, j0, rhi);
  const uint32_t c = (uint32_t)__popc(m);
  uint32_t incl = c;   // lanes of a wavefront: inclusive scan
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  if (totals && threadIdx.x == 0) {
    const uint32_t n = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    r.c->ndirty = n;
    r.c->min_dirty = n ? min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])) : 0xFFFFFFFFu;
    r.c->max_dirty = n ? max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])) : 0u;
  }
  if (!m) return;
  uint32_t at = s_part[0] + s_part[1] + s_part[2] + s_part[3] + incl - c;
  for (int q = 0; q < w; ++q) at += s_w[q];
  // (round 3) a big bucket is marked in the list (bit 31: the narrow kernels and their wavefront slots pass it over -- its index is
  // beyond every `hi` -- and k_update takes the mark off) and entered in the big list right here, so that k_eval_big does not have
  // to wait for the narrow kernel of the pass to find it: the two run side by side
  for (uint32_t mm = m; mm && at < LIST_CAP; mm &= mm - 1, ++at) {
    const uint32_t j = j0 + (uint32_t)__builtin_ctz(mm);
    const bool big = (r.bflags[j] & F_BIG) != 0;
    r.dlist[at] = j | (big ? 0x80000000u : 0u);
    if (big) {   // (listed by POSITION: k_eval_big only takes the part of the list that the pass's narrow kernel and k_update cover)
      const uint32_t bat = atomicAdd(&r.c->nbig, 1u);
      if (bat < LIST_CAP) r.blist[bat] = at | 0x80000000u;
    }
  }
}

// Tail sweeps.  A pair whose alignment is rejected is not entered in the seen-pair table, so the next bucket holding both reads
// aligns it again from its own anchors -- and is rejected again, and so on through the ~30 buckets the two reads share: one
// sweep (one lone 0.33 ms alignment) per hand-over, which is what the last ~15 sweeps of a 4.5 Gbase set consist of.  Once
// the sweeps are small, a bucket that files an alignment therefore also files the one every other registered reader of that
// pair would ask for (its rows for the two reads, its anchors): the results are in the memo when those buckets come to it.
// A speculative request is just an alignment whose result the memo holds; at worst it is never asked for.
// file the alignment of entries (row, par) of a bucket whose records start at s0, unless the memo knows it already
__device__ __forceinline__ void file_entries(const R &r, uint32_t s0, uint32_t row, uint32_t par) {
  const Ent e0 = entry_of(r.y0[s0 + row]), e1 = entry_of(r.y0[s0 + par]);
  if (e0.pos1 < e1.pos1 || e0.rid == e1.rid) return;
  const uint32_t dir0 = r.dir[s0 + row], dir1 = r.dir[s0 + par], q_off = e0.pos1 - e1.pos1;
  if (q_off >= (1u << 30)) return;
  const unsigned long long a = (unsigned long long)e0.rid << 32 | e1.rid;
  const uint32_t bk = q_off << 2 | dir0 << 1 | dir1;
  uint32_t i = (uint32_t)mix64(a ^ mix64(bk)) & r.mmask;
  for (int probes = 0; probes < 1024; ++probes) {
    unsigned long long cur = r.mt[i].a;
    if (cur == 0) {
      cur = atomicCAS(&r.mt[i].a, 0ULL, a);
      if (cur == 0) {  // new: a request of its own
        r.mt[i].b = bk + 1;
        const uint32_t my = atomicAdd(&r.c->nreq, 1u);
        if (my >= r.req_cap) {
          atomicOr(&r.c->overflow, OV_REQS);
          return;
        }
        pgx_align_key key;
        key.rid0 = e0.rid, key.rid1 = e1.rid, key.q_off = q_off, key.dir0 = (uint8_t)dir0, key.dir1 = (uint8_t)dir1, key.pad[0] = key.pad[1] = 0;
        r.rq_key[my] = key;
        r.mt[i].req = my;
        return;
      }
    }
    if (cur == a && *(volatile uint32_t *)&r.mt[i].b == bk + 1) return;  // known already
    i = (i + 1) & r.mmask;
  }
}
__device__ __forceinline__ void file_for_reader(const R &r, uint32_t C, uint32_t rid_a, uint32_t rid_b) {
  const uint32_t b = r.bid[C], s0 = r.bstart[b], n = r.bstart[b + 1] - s0;
  int ia = -1, ib = -1;
  bool twice = false;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t rid = (uint32_t)(r.y0[s0 + i] >> 32);
    if (rid == rid_a) twice |= ia >= 0, ia = (int)i;
    else if (rid == rid_b) twice |= ib >= 0, ib = (int)i;
  }
  if (ia < 0 || ib < 0 || twice) return;
  file_entries(r, s0, (uint32_t)min(ia, ib), (uint32_t)max(ia, ib));  // (the row is the entry with the smaller index)
}

// ---- file the alignments the converged lists still need ---------------------------------------------------------------
// (Only buckets that are not dirty right now are filed -- the others are about to be evaluated again.  Filing while the sweep
// is still running is always safe: a request is just an alignment whose result the memo will hold; at worst it is never
// asked for again.)
__global__ __launch_bounds__(256) void k_file(R r, uint32_t limit) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = j < limit && (r.bflags[j] & F_UNFILED) && !r.dirty[j];
  __shared__ uint32_t s_tot[4], s_base;
  if (!__syncthreads_or(active)) return;   // (block-uniform)
  uint32_t cnt = 0;
  if (active)
    for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next) cnt += (r.items[it - 1].info & I_UNFILED) ? 1u : 0u;
  // request numbers: ONE atomic per block.  The request counter is one address, and an L2 channel serves same-address atomics one
  // wavefront-instruction at a time (~12 ns): with an add per wavefront the first sweep's launch -- 9.4 M requests at c4s, 46 M in a
  // human-scale chunk -- lasted exactly requests / 64 x 12 ns (2.0 ms, 8.7 ms)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t incl = cnt;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
    if (lane >= o) incl += t;
  }
  const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
  if (lane == 63) s_tot[wv] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t all = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    s_base = all ? atomicAdd(&r.c->nreq, all) : 0u;
  }
  __syncthreads();
  uint32_t base = s_base;
  for (int w = 0; w < wv; ++w) base += s_tot[w];
  if (!__ballot(active)) return;
  // (from here on every lane of the wavefront stays in step: the fan-out below is done by all of them together)
  bool run = active && cnt != 0;
  if (active && !cnt) r.bflags[j] &= (uint8_t)~F_UNFILED;
  if (!__ballot(run)) return;
  if ((unsigned long long)base + total > r.req_cap) {
    if (run) atomicOr(&r.c->overflow, OV_REQS);
    return;
  }
  uint32_t my = base + incl - cnt;
  const uint32_t b = run ? r.bid[j] : 0u, s0 = run ? r.bstart[b] : 0u;
  uint32_t it = run ? r.ihead[j] : NIL;
  const bool tail = r.tail != 0;
  const uint32_t nn = run ? r.bstart[b + 1] - s0 : 0u;
  for (;;) {
    // ---- this lane's next unfiled item ----
    bool fan = false;             // the item filed a NEW alignment (tail mode): its pair's other readers are looked at below
    bool ah = false;              // the item was filed in tail mode: its row's next partners are filed ahead below
    uint32_t f_slot = 0, f_a = 0, f_b = 0, ah_ai = 0, ah_pi = 0;
    while (it != NIL && !fan && !ah) {
      Item &im = r.items[it - 1];
      const uint32_t nxt = im.next;
      if (im.info & I_UNFILED) {
        const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF;
        const Ent e0 = entry_of(r.y0[s0 + ai]), e1 = entry_of(r.y0[s0 + pi]);
        const uint32_t dir0 = r.dir[s0 + ai], dir1 = r.dir[s0 + pi], q_off = e0.pos1 - e1.pos1;
        const unsigned long long a = (unsigned long long)e0.rid << 32 | e1.rid;
        const uint32_t bk = q_off << 2 | dir0 << 1 | dir1;
        pgx_align_key key;
        key.rid0 = e0.rid, key.rid1 = e1.rid, key.q_off = q_off, key.dir0 = (uint8_t)dir0, key.dir1 = (uint8_t)dir1, key.pad[0] = key.pad[1] = 0;
        // find or insert.  Another lane may be inserting the same key right now: its `b` may still read 0, in which case this
        // lane files a duplicate in another slot (same alignment, same result -- harmless).
        uint32_t i = (uint32_t)mix64(a ^ mix64(bk)) & r.mmask, found = NONE;
        bool fresh = false;
        for (int probes = 0; probes < 1024; ++probes) {
          unsigned long long cur = r.mt[i].a;
          if (cur == 0) {
            cur = atomicCAS(&r.mt[i].a, 0ULL, a);
            if (cur == 0) {
              r.mt[i].b = bk + 1;
              found = i, fresh = true;
              break;
            }
          }
          if (cur == a && *(volatile uint32_t *)&r.mt[i].b == bk + 1) {
            found = i;
            break;
          }
          i = (i + 1) & r.mmask;
        }
        if (found == NONE) {   // (this bucket stays unfiled; the overflow bit sends the walk to larger tables)
          atomicOr(&r.c->overflow, OV_MEMO);
          run = false, it = NIL;
          break;
        }
        r.rq_key[my] = key;  // (a request slot whose key was already filed by someone else just repeats that alignment)
        if (fresh) r.mt[found].req = my;
        ++my;
        im.mslot = found;
        im.info &= ~I_UNFILED;
        if (tail) ah = true, ah_ai = ai, ah_pi = pi;   // ... and the row's next partners (below)
        if (tail && fresh) fan = true, f_slot = im.pslot, f_a = e0.rid, f_b = e1.rid;
      }
      it = nxt;
    }
    const uint64_t fm = __ballot(fan), am = __ballot(ah);
    if (!fm && !am && !__ballot(it != NIL)) break;
    // ---- tail mode: the row's next partners -- if this candidate is rejected the row goes on to them (a row of a repeat-rich bucket can
    // have dozens of candidates, each rejection otherwise costing a sweep) -- a partner per lane.  (Through round 3 the filing lane walked
    // its r.tail partners itself, a dependent memo probe each, item after item: the tail sweeps' k_file launches took up to 7 ms at c4s.)
    for (uint64_t mm = am; mm; mm &= mm - 1) {
      const int L = __builtin_ctzll(mm);
      const uint32_t sL = (uint32_t)__shfl((int)s0, L, 64), aL = (uint32_t)__shfl((int)ah_ai, L, 64), pL = (uint32_t)__shfl((int)ah_pi, L, 64);
      const uint32_t nL = (uint32_t)__shfl((int)nn, L, 64);
      for (uint32_t p = pL + 1 + (uint32_t)lane; p < nL && p <= pL + r.tail; p += 64) file_entries(r, sL, aL, p);
    }
    // ---- tail mode: the alignment every OTHER reader of a newly requested pair would ask for (file_for_reader) -- by the whole
    // wavefront, a reader bucket per lane.  (Round 2 left this to the filing lane alone: a pair of a repeat-rich set has dozens of
    // readers of up to 128 entries each, scanned one after the other -- k_file was 21 ms of a c4s step and 42 ms of c5s'.)
    for (uint64_t mm = fm; mm; mm &= mm - 1) {
      const int L = __builtin_ctzll(mm);
      const uint32_t ps = (uint32_t)__shfl((int)f_slot, L, 64), ra = (uint32_t)__shfl((int)f_a, L, 64), rb2 = (uint32_t)__shfl((int)f_b, L, 64);
      const uint32_t jj = (uint32_t)__shfl((int)j, L, 64);
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&r.pc[ps >> r.cshift]);
      const uint32_t c = min(w[0], NIN);
      for (uint32_t q = (uint32_t)lane; q < c; q += 64) {
        const uint32_t rb = w[2 + q];
        if (rb != 0 && rb - 1 != jj && rb - 1 < r.nb) file_for_reader(r, rb - 1, ra, rb2);
      }
      if (c >= NIN) {   // the overflow list: every lane walks it (one broadcast load per node), node i goes to lane i % 64
        uint32_t idx = 0;
        for (uint32_t nd = w[1]; nd != NIL; nd = r.rn[nd - 1].next, ++idx)
          if ((idx & 63u) == (uint32_t)lane) {
            const uint32_t rb = r.rn[nd - 1].bucket;
            if (rb != jj && rb < r.nb) file_for_reader(r, rb, ra, rb2);
          }
      }
    }
  }
  if (run) r.bflags[j] &= (uint8_t)~F_UNFILED;
}

// ---- check the guesses against the results ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_settle(R r) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= r.nb || !(r.bflags[j] & F_GUESS)) return;
  const uint32_t b = r.bid[j], s0 = r.bstart[b];
  bool bad = false, remain = false;
  for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next) {
    Item &im = r.items[it - 1];
    if (!(im.info & I_GUESS)) continue;
    const uint32_t req = im.mslot != NONE ? r.mt[im.mslot].req : NONE;
    if (req >= r.settled) {
      remain = true;
      continue;
    }
    const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF, gtype = (im.info >> 16) & 3;
    const Ent e0 = entry_of(r.y0[s0 + ai]), e1 = entry_of(r.y0[s0 + pi]);
    uint32_t type;
    const bool acc = classify(r.rq_res[req], r.rlen[e0.rid], r.rlen[e1.rid], e0.pos1 - e1.pos1, &type);
    if (!acc || type != gtype) bad = true;
#ifdef PGX_SETTLE_STATS
    if (!acc || type != gtype) {
      const uint32_t rl0 = r.rlen[e0.rid], rl1 = r.rlen[e1.rid], qo = e0.pos1 - e1.pos1;
      const uint32_t ol = min(rl0 - qo, rl1);
      const pgx_match mm = r.rq_res[req];
      int cat = acc ? 3 : (ol <= 520 ? 4 : (mm.q_end == 0 && mm.t_end == 0 ? 5 : 6));
      atomicAdd(&r.spread[((j >> 6) % SPREAD) * 8 + cat], 1ULL);
    }
#endif
    else im.info &= ~I_GUESS;
  }
  if (bad) r.dirty[j] = 1;
  if (!remain) r.bflags[j] &= (uint8_t)~F_GUESS;
}

// ---- the ovlp_t records, bucket by bucket in visit order, each bucket's in evaluation order ---------------------------
__global__ __launch_bounds__(256) void k_emit(R r, const uint32_t *__restrict__ off, pgx_ovlp *__restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long lk = 0, sk = 0, ck = 0;
  if (j < r.nb) {
    lk = r.lookups[j], sk = r.skips[j];
    const uint32_t num = r.inum[j];
    if (num) {
      const uint32_t b = r.bid[j], s0 = r.bstart[b];
      uint32_t k = 0;
      for (uint32_t it = r.ihead[j]; it != NIL; it = r.items[it - 1].next, ++k) {  // the list runs newest first
        const Item im = r.items[it - 1];
        const uint32_t ai = im.info & 0xFF, pi = (im.info >> 8) & 0xFF;
        const uint64_t ya = r.y0[s0 + ai], yb = r.y0[s0 + pi];
        pgx_ovlp o;
        o.y0 = ya, o.y1 = yb;
        o.rl0 = r.rlen[(uint32_t)(ya >> 32)], o.rl1 = r.rlen[(uint32_t)(yb >> 32)];
        o.strand0 = r.dir[s0 + ai], o.strand1 = r.dir[s0 + pi], o.ovlp_type = (uint8_t)((im.info >> 16) & 3), o.pad0 = 0;
        o.match = r.rq_res[r.mt[im.mslot].req];
        o.pad1 = 0;
        out[(size_t)off[j] + (num - 1 - k)] = o;
        ck += record_checksum(o, (uint64_t)off[j] + (num - 1 - k));
      }
    }
  }
  // totals: one atomic pair per wavefront
  for (int o = 32; o; o >>= 1) {
    lk += (unsigned long long)__shfl_xor((int)(lk >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)lk, o, 64);
    sk += (unsigned long long)__shfl_xor((int)(sk >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)sk, o, 64);
    ck += (unsigned long long)__shfl_xor((int)(ck >> 32), o, 64) << 32 | (uint32_t)__shfl_xor((int)ck, o, 64);
  }
  if ((threadIdx.x & 63) == 0 && (lk | sk | ck)) {
    unsigned long long *line = r.spread + ((j >> 6) % SPREAD) * 8;
    atomicAdd(line + 1, lk);
    atomicAdd(line + 2, sk);
#if !defined(PGX_BIG_STATS) && !defined(PGX_SETTLE_STATS)   // (the statistics builds count in the same words)
    atomicAdd(line + 7, ck);
#endif
  }
}

// read pairs the walk has entered in the pair table (what the next stage's table is sized by: dev_replay)
__global__ __launch_bounds__(256) void k_count_pairs(const PHot *__restrict__ ph, uint32_t cap, unsigned long long *__restrict__ out) {
  uint32_t c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) c += ph[i].key != 0;
  for (int o = 32; o; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

}  // namespace rp
}  // namespace pgx
