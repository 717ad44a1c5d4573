// pgx_khash.h -- host emulation of klib khash's slot layout (shared by the overlap stage and the query helpers).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace pgx {

// ---------------------------------------------------------------------------------------------------------
// klib khash slot-order emulation, keys + a dense id payload (contract: SURVEY.md 8a-11 / Appendix A1;
// behaviour of src/khash.h:232-336 with no deletions; hash src/khash.h:373; load factor 0.77 src/khash.h:180).
// ---------------------------------------------------------------------------------------------------------
struct SlotTable {
  uint32_t nb = 0, size = 0, upper = 0;
  uint64_t *keys = nullptr;
  uint32_t *ids = nullptr;
  uint8_t *used = nullptr;
  SlotTable() = default;
  SlotTable(const SlotTable &) = delete;
  SlotTable &operator=(const SlotTable &) = delete;
  SlotTable(SlotTable &&o) noexcept { *this = std::move(o); }
  SlotTable &operator=(SlotTable &&o) noexcept {
    std::swap(nb, o.nb), std::swap(size, o.size), std::swap(upper, o.upper);
    std::swap(keys, o.keys), std::swap(ids, o.ids), std::swap(used, o.used);
    return *this;
  }
  ~SlotTable() { free(keys), free(ids), free(used); }
  static uint32_t h32(uint64_t k) { return (uint32_t)(k >> 33 ^ k ^ k << 11); }

  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    uint8_t *fresh = (uint8_t *)calloc(nn, 1);
    keys = (uint64_t *)realloc(keys, (size_t)nn * 8);
    ids = (uint32_t *)realloc(ids, (size_t)nn * 4);
    const uint32_t m = nn - 1;
    for (uint32_t j = 0; j < nb; ++j) {
      if (j + 16 < nb && used[j + 16]) {  // the destinations are random slots of a table that outgrew the caches
        const uint32_t d = h32(keys[j + 16]) & m;
        __builtin_prefetch(fresh + d, 1), __builtin_prefetch(keys + d, 1), __builtin_prefetch(ids + d, 1);
      }
      if (!used[j]) continue;
      uint64_t key = keys[j];
      uint32_t id = ids[j];
      used[j] = 0;
      for (;;) {  // move the element; an occupied, not yet moved destination is evicted and carried on
        uint32_t i = h32(key) & m, step = 0;
        while (fresh[i]) i = (i + (++step)) & m;
        fresh[i] = 1;
        if (i < nb && used[i]) {
          std::swap(key, keys[i]), std::swap(id, ids[i]);
          used[i] = 0;
        } else {
          keys[i] = key, ids[i] = id;
          break;
        }
      }
    }
    free(used);
    used = fresh, nb = nn, upper = thr;
  }
  // the load check precedes the lookup, so a put of an existing key can still trigger the resize
  uint32_t put(uint64_t key, uint32_t fresh_id, bool *absent) {
    if (size >= upper) enlarge();
    const uint32_t m = nb - 1;
    uint32_t i = h32(key) & m, step = 0;
    while (used[i] && keys[i] != key) i = (i + (++step)) & m;
    if (used[i]) {
      *absent = false;
      return ids[i];
    }
    used[i] = 1, keys[i] = key, ids[i] = fresh_id, ++size;
    *absent = true;
    return fresh_id;
  }
  void prefetch(uint64_t key) const {  // start the miss a later put(key) will take
    if (!nb) return;
    const uint32_t i = h32(key) & (nb - 1);
    __builtin_prefetch(used + i, 1), __builtin_prefetch(keys + i, 1);
  }
};


// ---------------------------------------------------------------------------------------------------------
// The same slot layout for a sequence of DISTINCT keys, computed without walking the probe chains again and again.
//
// khash probes h, h+1, h+3, h+6, ... (triangular steps) and nothing is ever deleted, so for a given home slot h the
// leading positions of its sequence that were found occupied stay occupied: a put that starts where the previous put with
// the same home left off finds exactly the slot khash would find.  `skip[h]` = number of leading sequence positions of
// home h known to be occupied.  This matters because the outer key, (small hash) << 8 | span, is a terrible input for
// khash's integer hash: its low 8 index bits are the span, so ~1/256 of the slots are the home of ALL keys, chains are
// hundreds of probes long, and the literal replay of 2.7 M keys took 1.1 s (15 Gbases) where this form takes a fraction.
// The rehash of a resize (khash.h:258-284, kick-out order preserved) uses the same trick on its `fresh` bitmap, and its
// skip counts stay valid for the resized table.  Keys MUST be distinct; a put of a present key -- which in khash still
// runs the load-factor check and may resize -- is `touch()`.
// ---------------------------------------------------------------------------------------------------------
struct DistinctSlotTable {
  // One 8-byte word per slot (round 3; round 2 kept keys / ids / occupancy in three arrays, i.e. three cache misses per put on
  // a table that has outgrown the caches): the key's 32-bit hash -- all a resize ever needs of the key, the keys being distinct
  // and never compared -- | a 2-bit state | the 30-bit id.  State 0 = empty; 1 / 2 = occupied, the value alternating from one
  // resize to the next, so that during a rehash "not moved yet" (the old value) and "placed" (the new one) are told apart in
  // place, as khash does with its two flag arrays (khash.h:258-284).
  static constexpr uint32_t ID_MASK = 0x3FFFFFFFu;
  uint32_t nb = 0, size = 0, upper = 0;
  uint64_t *slot = nullptr;
  uint32_t *skip = nullptr;
  uint32_t live = 1;
  // reserve(): the arrays at their final size from the caller's allocator (pooled huge-page mappings in the overlap stage), the
  // skip array twice (a resize builds the new one beside the old): no realloc, no calloc, no page faults per resize
  void *(*arena_alloc)(size_t) = nullptr;
  void (*arena_free)(void *, size_t) = nullptr;
  void (*slot_free)(void *, size_t) = nullptr;   // (the slot array may come from another allocator: pinned memory the device reads)
  uint32_t cap = 0;
  uint32_t *sb[2] = {nullptr, nullptr};
  int cur = 0;
  DistinctSlotTable() = default;
  DistinctSlotTable(const DistinctSlotTable &) = delete;
  DistinctSlotTable &operator=(const DistinctSlotTable &) = delete;
  ~DistinctSlotTable() {
    if (cap) {
      slot_free(slot, (size_t)cap * 8);
      for (int i = 0; i < 2; ++i) arena_free(sb[i], (size_t)cap * 4);
    } else {
      free(slot), free(skip);
    }
  }
  bool is_used(uint32_t s) const { return slot[s] != 0; }   // (outside a resize every non-empty slot is live)
  uint32_t id_at(uint32_t s) const { return (uint32_t)slot[s] & ID_MASK; }
  void reserve(size_t n_keys, void *(*al)(size_t), void (*fr)(void *, size_t), void *(*slot_al)(size_t) = nullptr,
               void (*slot_fr)(void *, size_t) = nullptr) {  // before the first put
    uint32_t nn = 4;
    while ((uint32_t)(nn * 0.77 + 0.5) <= n_keys && nn < (1u << 31)) nn <<= 1;   // the table stops growing once upper > size
    nn = nn < (1u << 31) ? nn * 2 : nn;   // (one spare doubling: a trailing touch() may still resize)
    arena_alloc = al, arena_free = fr, cap = nn;
    slot_free = slot_al ? slot_fr : fr;
    slot = (uint64_t *)(slot_al ? slot_al : al)((size_t)nn * 8);
    for (int i = 0; i < 2; ++i) sb[i] = (uint32_t *)al((size_t)nn * 4);
  }
  static inline uint32_t at(uint32_t home, uint32_t step, uint32_t m) {  // position after `step` triangular increments
    return (uint32_t)((uint64_t)home + (uint64_t)step * (step + 1) / 2) & m;
  }
  static inline uint32_t state(uint64_t e) { return (uint32_t)e >> 30; }
  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    uint32_t *fskip;
    if (cap && nn <= cap) {
      fskip = sb[cur ^ 1];
      memset(fskip, 0, (size_t)nn * 4);
    } else {
      if (cap) {  // grew beyond the reservation (cannot happen with a correct n_keys): fall back to the heap, keep the contents
        uint64_t *s2 = (uint64_t *)malloc((size_t)nn * 8);
        memcpy(s2, slot, (size_t)nb * 8);
        slot_free(slot, (size_t)cap * 8);
        for (int i = 0; i < 2; ++i) arena_free(sb[i], (size_t)cap * 4);
        slot = s2, skip = nullptr, cap = 0;
      } else {
        slot = (uint64_t *)realloc(slot, (size_t)nn * 8);
      }
      fskip = (uint32_t *)calloc(nn, sizeof(uint32_t));
    }
    memset(slot + nb, 0, (size_t)(nn - nb) * 8);   // the new half: empty
    const uint32_t m = nn - 1, old = live, nw = live ^ 3u;
    for (uint32_t j = 0; j < nb; ++j) {
      if (j + 32 < nb && slot[j + 32]) __builtin_prefetch(fskip + ((uint32_t)(slot[j + 32] >> 32) & m), 1);
      if (j + 12 < nb && slot[j + 12]) {  // (fskip of that home is in the cache by now)
        const uint32_t hh = (uint32_t)(slot[j + 12] >> 32) & m;
        __builtin_prefetch(slot + at(hh, fskip[hh], m), 1);
      }
      uint64_t e = slot[j];
      if (state(e) != old) continue;   // empty, or placed here earlier in this resize
      slot[j] = 0;
      for (;;) {  // move the element; an occupied, not yet moved destination is evicted and carried on
        const uint32_t h = (uint32_t)(e >> 32) & m;
        uint32_t step = fskip[h], i = at(h, step, m);
        while (state(slot[i]) == nw) i = (i + (++step)) & m;
        fskip[h] = step + 1;
        const uint64_t prev = slot[i];
        slot[i] = (e & ~((uint64_t)3 << 30)) | ((uint64_t)nw << 30);
        if (state(prev) != old) break;
        e = prev;
      }
    }
    if (cap) cur ^= 1;
    else free(skip);
    skip = fskip, nb = nn, upper = thr, live = nw;
  }
  void touch() {  // a put of a key that is already present: only the load-factor check has an effect
    if (size >= upper) enlarge();
  }
  void put_new(uint64_t key, uint32_t id) {  // key must not be present; id < 2^30
    if (size >= upper) enlarge();
    const uint32_t hv = SlotTable::h32(key), m = nb - 1, h = hv & m;
    uint32_t step = skip[h], i = at(h, step, m);
    while (slot[i]) i = (i + (++step)) & m;
    skip[h] = step + 1;
    slot[i] = ((uint64_t)hv << 32) | ((uint64_t)live << 30) | id, ++size;
  }
  void prefetch_home(uint64_t key) const {  // first stage: the skip count of the key's home slot
    if (nb) __builtin_prefetch(skip + (SlotTable::h32(key) & (nb - 1)), 1);
  }
  void prefetch(uint64_t key) const {  // start the miss a later put_new(key) will take
    if (!nb) return;
    const uint32_t m = nb - 1, h = SlotTable::h32(key) & m;
    __builtin_prefetch(slot + at(h, skip[h], m), 1);
  }
};

// A reusable inner table: same slot behaviour as SlotTable, but storage is recycled between key0 groups so the
// ~10^5..10^6 tiny second-level tables cost no allocation.
struct ScratchTable {
  std::vector<uint64_t> keys;
  std::vector<uint32_t> ids;
  std::vector<uint8_t> used, fresh;
  uint32_t nb = 0, size = 0, upper = 0;
  void reset() {
    if (nb) std::fill(used.begin(), used.begin() + nb, 0);
    nb = size = upper = 0;
  }
  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    if (keys.size() < nn) keys.resize(nn), ids.resize(nn), used.resize(nn, 0), fresh.resize(nn, 0);
    std::fill(fresh.begin(), fresh.begin() + nn, 0);
    const uint32_t m = nn - 1;
    for (uint32_t j = 0; j < nb; ++j) {
      if (!used[j]) continue;
      uint64_t key = keys[j];
      uint32_t id = ids[j];
      used[j] = 0;
      for (;;) {
        uint32_t i = SlotTable::h32(key) & m, step = 0;
        while (fresh[i]) i = (i + (++step)) & m;
        fresh[i] = 1;
        if (i < nb && used[i]) {
          std::swap(key, keys[i]), std::swap(id, ids[i]);
          used[i] = 0;
        } else {
          keys[i] = key, ids[i] = id;
          break;
        }
      }
    }
    std::copy(fresh.begin(), fresh.begin() + nn, used.begin());
    nb = nn, upper = thr;
  }
  uint32_t put(uint64_t key, uint32_t fresh_id, bool *absent) {
    if (size >= upper) enlarge();
    const uint32_t m = nb - 1;
    uint32_t i = SlotTable::h32(key) & m, step = 0;
    while (used[i] && keys[i] != key) i = (i + (++step)) & m;
    if (used[i]) {
      *absent = false;
      return ids[i];
    }
    used[i] = 1, keys[i] = key, ids[i] = fresh_id, ++size;
    *absent = true;
    return fresh_id;
  }
};


}  // namespace pgx
