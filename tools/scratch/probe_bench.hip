// micro-benchmark: how fast can lanes chase dependent random probes (16-B slot load + atomicExch + node store) on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Slot { unsigned long long key; unsigned own, rhead; };
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int MODE>
__global__ void k_probe(Slot *tab, uint64_t mask, uint2 *nodes, unsigned *node_top, int steps, uint64_t n, unsigned *sink) {
  const uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (b >= n) return;
  unsigned acc = 0;
  unsigned base = 0;
  if (MODE >= 1) base = atomicAdd(node_top, (unsigned)steps);
  for (int s = 0; s < steps; ++s) {
    const uint64_t pair = mix(b * 64 + s + acc);  // dependent chain
    Slot *sl = tab + (pair & mask);
    // load the slot (16 B)
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(sl);
    if (MODE >= 2 && v.x == 0) atomicCAS(&sl->key, 0ULL, pair | 1);
    if (MODE >= 1) {
      const unsigned prev = atomicExch(&sl->rhead, base + s + 1);
      nodes[base + s] = make_uint2(prev, (unsigned)b);
    }
    acc += (unsigned)(v.y & 1);
  }
  if (acc == 0xFFFFFFFF) *sink = acc;
}
int main(int argc, char **argv) {
  const uint64_t slots = 1ULL << 26;  // 1 GiB
  Slot *tab; CK(hipMalloc(&tab, slots * sizeof(Slot))); CK(hipMemset(tab, 0, slots * sizeof(Slot)));
  uint2 *nodes; CK(hipMalloc(&nodes, (size_t)64 << 20 << 3));
  unsigned *top, *sink; CK(hipMalloc(&top, 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (uint64_t n : {1330000ULL, 262144ULL, 65536ULL, 16384ULL}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int bs : {64, 256}) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(top, 0, 4));
          CK(hipEventRecord(e0));
          const int steps = 34;
          dim3 g((unsigned)((n + bs - 1) / bs));
          if (mode == 0) hipLaunchKernelGGL(k_probe<0>, g, dim3(bs), 0, 0, tab, slots - 1, nodes, top, steps, n, sink);
          if (mode == 1) hipLaunchKernelGGL(k_probe<1>, g, dim3(bs), 0, 0, tab, slots - 1, nodes, top, steps, n, sink);
          if (mode == 2) hipLaunchKernelGGL(k_probe<2>, g, dim3(bs), 0, 0, tab, slots - 1, nodes, top, steps, n, sink);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        printf("n=%llu lanes x 34 dependent probes, mode %d (0 load, 1 +exch+node, 2 +cas), block %d: %.3f ms = %.2f G probes/s\n",
               (unsigned long long)n, mode, bs, best, n * 34 / best / 1e6);
      }
    }
  }
  return 0;
}
