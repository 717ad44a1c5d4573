#include <chrono>
#include <cstdio>
#include <random>
#include <unordered_set>
#include "../peregrine_amd/csrc/pgx_khash.h"
using namespace pgx;
#include <sys/mman.h>
static void *al(size_t n) { n = (n + (2u<<20) - 1) & ~((size_t)(2u<<20) - 1); void *p = mmap(nullptr, n, PROT_READ|PROT_WRITE, MAP_PRIVATE|MAP_ANONYMOUS, -1, 0); madvise(p, n, MADV_HUGEPAGE); return p; }
static void fr(void *p, size_t n) { n = (n + (2u<<20) - 1) & ~((size_t)(2u<<20) - 1); munmap(p, n); }
int main(int argc, char **argv) {
  size_t n = argc > 1 ? atol(argv[1]) : 4400000;
  int d1 = argc > 2 ? atoi(argv[2]) : 24, d2 = argc > 3 ? atoi(argv[3]) : 8;
  std::mt19937_64 rng(7);
  std::vector<uint64_t> k;
  std::unordered_set<uint64_t> seen;
  if (const char *kf = getenv("KB_KEYS")) {   // keys dumped by the library (a debugging dump that round 3 had)
    FILE *f = fopen(kf, "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    n = (size_t)ftell(f) / 8;
    fseek(f, 0, SEEK_SET);
    k.resize(n);
    if (fread(k.data(), 8, n, f) != n) return 1;
    fclose(f);
    unsigned long long orv = 0, spans = 0;
    for (uint64_t x : k) orv |= x >> 8, spans |= 1ull << ((x & 0xFF) & 63);
    printf("%zu keys from %s: OR of the hashes %llx, span bit set %llx\n", n, kf, orv, spans);
  }
  while (k.size() < n) {
    uint64_t key = getenv("KB_REAL") ? ((((rng() & 0xFFFFFFFFull) >> 7) << 8) | 16) : (((rng() & 0xFFFFFFFFull) << 8) | (20 + rng() % 100));   // KB_REAL: the reference's keys -- hash << 8 | k, the hash a minimizer's (top bits zero)
    if (seen.insert(key).second) k.push_back(key);
  }
  for (int rep = 0; rep < 3; ++rep) {
    DistinctSlotTable t;
    t.reserve(n, al, fr);
    auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i) {
      if (i + d1 < n) t.prefetch_home(k[i + d1]);
      if (i + d2 < n) t.prefetch(k[i + d2]);
      t.put_new(k[i], (uint32_t)i);
    }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint64_t cs = 0;
    for (uint32_t s = 0; s < t.nb; ++s) if (t.is_used(s)) cs = cs * 1000003 + t.id_at(s);
    printf("n %zu d %d/%d: %.1f ms (%.1f ns/key) nb %u cs %llx\n", n, d1, d2, ms, ms * 1e6 / n, t.nb, (unsigned long long)cs);
  }
}
