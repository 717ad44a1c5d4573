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
  uint32_t nb = 0, size = 0, upper = 0;
  uint64_t *keys = nullptr;
  uint32_t *ids = nullptr, *skip = nullptr;
  uint8_t *used = nullptr;
  // reserve(): all arrays at their final size from the caller's allocator (pooled huge-page mappings in the overlap stage), the
  // occupancy / skip arrays twice (a resize builds the new ones beside the old): no realloc, no calloc, no page faults per resize
  void *(*arena_alloc)(size_t) = nullptr;
  void (*arena_free)(void *, size_t) = nullptr;
  uint32_t cap = 0;
  uint8_t *ub[2] = {nullptr, nullptr};
  uint32_t *sb[2] = {nullptr, nullptr};
  int cur = 0;
  DistinctSlotTable() = default;
  DistinctSlotTable(const DistinctSlotTable &) = delete;
  DistinctSlotTable &operator=(const DistinctSlotTable &) = delete;
  ~DistinctSlotTable() {
    if (cap) {
      arena_free(keys, (size_t)cap * 8), arena_free(ids, (size_t)cap * 4);
      for (int i = 0; i < 2; ++i) arena_free(ub[i], cap), arena_free(sb[i], (size_t)cap * 4);
    } else {
      free(keys), free(ids), free(used), free(skip);
    }
  }
  void reserve(size_t n_keys, void *(*al)(size_t), void (*fr)(void *, size_t)) {  // before the first put
    uint32_t nn = 4;
    while ((uint32_t)(nn * 0.77 + 0.5) <= n_keys && nn < (1u << 31)) nn <<= 1;   // the table stops growing once upper > size
    nn = nn < (1u << 31) ? nn * 2 : nn;   // (one spare doubling: a trailing touch() may still resize)
    arena_alloc = al, arena_free = fr, cap = nn;
    keys = (uint64_t *)al((size_t)nn * 8), ids = (uint32_t *)al((size_t)nn * 4);
    for (int i = 0; i < 2; ++i) ub[i] = (uint8_t *)al(nn), sb[i] = (uint32_t *)al((size_t)nn * 4);
  }
  static inline uint32_t at(uint32_t home, uint32_t step, uint32_t m) {  // position after `step` triangular increments
    return (uint32_t)((uint64_t)home + (uint64_t)step * (step + 1) / 2) & m;
  }
  void enlarge() {
    const uint32_t nn = nb ? nb * 2 : 4;
    const uint32_t thr = (uint32_t)(nn * 0.77 + 0.5);
    if (size >= thr) return;
    uint8_t *fresh;
    uint32_t *fskip;
    if (cap && nn <= cap) {
      fresh = ub[cur ^ 1], fskip = sb[cur ^ 1];
      memset(fresh, 0, nn), memset(fskip, 0, (size_t)nn * 4);
    } else {
      if (cap) {  // grew beyond the reservation (cannot happen with a correct n_keys): fall back to the heap, keep the contents
        uint64_t *k2 = (uint64_t *)malloc((size_t)nn * 8);
        uint32_t *i2 = (uint32_t *)malloc((size_t)nn * 4);
        uint8_t *u2 = (uint8_t *)malloc(nb ? nb : 1);
        memcpy(k2, keys, (size_t)nb * 8), memcpy(i2, ids, (size_t)nb * 4), memcpy(u2, used, nb);
        arena_free(keys, (size_t)cap * 8), arena_free(ids, (size_t)cap * 4);
        for (int i = 0; i < 2; ++i) arena_free(ub[i], cap), arena_free(sb[i], (size_t)cap * 4);
        keys = k2, ids = i2, used = u2, skip = nullptr, cap = 0;
      } else {
        keys = (uint64_t *)realloc(keys, (size_t)nn * 8);
        ids = (uint32_t *)realloc(ids, (size_t)nn * 4);
      }
      fresh = (uint8_t *)calloc(nn, 1);
      fskip = (uint32_t *)calloc(nn, sizeof(uint32_t));
    }
    const uint32_t m = nn - 1;
    for (uint32_t j = 0; j < nb; ++j) {
      if (j + 32 < nb && used[j + 32]) __builtin_prefetch(fskip + (SlotTable::h32(keys[j + 32]) & m), 1);
      if (j + 12 < nb && used[j + 12]) {  // (fskip of that home is in the cache by now)
        const uint32_t hh = SlotTable::h32(keys[j + 12]) & m, d = at(hh, fskip[hh], m);
        __builtin_prefetch(fresh + d, 1), __builtin_prefetch(keys + d, 1), __builtin_prefetch(ids + d, 1);
      }
      if (!used[j]) continue;
      uint64_t key = keys[j];
      uint32_t id = ids[j];
      used[j] = 0;
      for (;;) {  // move the element; an occupied, not yet moved destination is evicted and carried on
        const uint32_t h = SlotTable::h32(key) & m;
        uint32_t step = fskip[h], i = at(h, step, m);
        while (fresh[i]) i = (i + (++step)) & m;
        fskip[h] = step + 1;
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
    if (cap) cur ^= 1;
    else free(used), free(skip);
    used = fresh, skip = fskip, nb = nn, upper = thr;
  }
  void touch() {  // a put of a key that is already present: only the load-factor check has an effect
    if (size >= upper) enlarge();
  }
  void put_new(uint64_t key, uint32_t id) {  // key must not be present
    if (size >= upper) enlarge();
    const uint32_t m = nb - 1, h = SlotTable::h32(key) & m;
    uint32_t step = skip[h], i = at(h, step, m);
    while (used[i]) i = (i + (++step)) & m;
    skip[h] = step + 1;
    used[i] = 1, keys[i] = key, ids[i] = id, ++size;
  }
  void prefetch_home(uint64_t key) const {  // first stage: the skip count of the key's home slot
    if (nb) __builtin_prefetch(skip + (SlotTable::h32(key) & (nb - 1)), 1);
  }
  void prefetch(uint64_t key) const {  // start the misses a later put_new(key) will take
    if (!nb) return;
    const uint32_t m = nb - 1, h = SlotTable::h32(key) & m;
    const uint32_t i = at(h, skip[h], m);
    __builtin_prefetch(used + i, 1), __builtin_prefetch(keys + i, 1), __builtin_prefetch(ids + i, 1);
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
