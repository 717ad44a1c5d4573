// tools/tlb_probe.hip -- how many distinct address ranges can a CU walk before its translation cache (UTCL1) thrashes, and how large
// is the range one entry covers?  One wavefront per workgroup, `nwg` workgroups; lane 0 chases a ring of N pointers that are `stride`
// bytes apart (so N lines of 128 B: the data always hits the L2; what changes with the stride is the number of translations).
// Prints ns per dependent load for every (stride, N).   hipcc --offload-arch=gfx950 -O2 tools/tlb_probe.hip -o gpurun_out/tlb_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));     \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void k_init(uint8_t *base, size_t stride, uint32_t n, size_t wg_off, uint32_t nwg) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * nwg) return;
  const uint32_t w = i / n, j = i % n;
  // a fixed odd step through the ring so that consecutive loads are not neighbours (defeats any next-range prefetch)
  const uint32_t step = n > 4 ? (n / 2) | 1u : 1u;
  uint32_t g = step;   // step must be coprime with n: bump until it is
  for (;;) {
    uint32_t a = g, b = n;
    while (b) { const uint32_t t = a % b; a = b; b = t; }
    if (a == 1) break;
    g += 2;
  }
  const uint32_t nxt = (j + g) % n;
  uint8_t *p = base + (size_t)j * stride + (size_t)w * wg_off;
  *reinterpret_cast<uint64_t *>(p) = reinterpret_cast<uint64_t>(base + (size_t)nxt * stride + (size_t)w * wg_off);
}
__global__ __launch_bounds__(64) void k_chase(uint8_t *base, size_t wg_off, uint32_t iters, uint64_t *sink, uint32_t *hw) {
  const uint8_t *p = base + (size_t)blockIdx.x * wg_off;
  uint64_t v = reinterpret_cast<uint64_t>(p);
  if (threadIdx.x == 0) {
    for (uint32_t i = 0; i < iters; ++i) v = *reinterpret_cast<const volatile uint64_t *>(v);
    sink[blockIdx.x] = v;
    if (hw) {
      hw[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, all 32 bits
      hw[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
  }
}

int main(int argc, char **argv) {
  const size_t gb = argc > 1 ? atol(argv[1]) : 40;
  const uint32_t nwg = argc > 2 ? atoi(argv[2]) : 256;
  uint8_t *buf;
  CK(hipMalloc(&buf, gb << 30));
  CK(hipMemset(buf, 0, gb << 30));
  uint64_t *sink;
  uint32_t *hw;
  CK(hipMalloc(&sink, 8 * 65536));
  CK(hipMalloc(&hw, 8 * 65536));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {   // where do the workgroups of a persistent grid land?  (xcc, se, cu) of workgroup i of an 8192-wavefront launch
    const uint32_t g = 8192;
    hipLaunchKernelGGL(k_init, dim3((g + 255) / 256), dim3(256), 0, 0, buf, (size_t)4096, 1u, (size_t)256, g);
    hipLaunchKernelGGL(k_chase, dim3(g), dim3(64), 0, 0, buf, (size_t)256, 20000u, sink, hw);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(2 * g);
    CK(hipMemcpy(h.data(), hw, 8 * g, hipMemcpyDeviceToHost));
    printf("# workgroup -> xcc se sh cu simd wave  (first 48 and every 512th)\n");
    for (uint32_t i = 0; i < g; ++i)
      if (i < 48 || i % 512 == 0) {
        const uint32_t id = h[2 * i], xc = h[2 * i + 1];
        printf("  wg %5u  xcc %u  se %u sh %u cu %2u simd %u wave %2u   (hw_id %08x xcc_id %08x)\n", i, xc & 15, (id >> 13) & 7, (id >> 12) & 1, (id >> 8) & 15,
               (id >> 4) & 3, id & 15, id, xc);
      }
    // distinct (xcc, se, sh, cu) tuples and workgroups per tuple
    std::vector<int> cnt(1 << 12, 0);
    for (uint32_t i = 0; i < g; ++i) cnt[((h[2 * i + 1] & 15) << 8) | ((h[2 * i] >> 8) & 0xFF)]++;
    int tuples = 0, mn = 1 << 30, mx = 0;
    for (int c : cnt)
      if (c) ++tuples, mn = c < mn ? c : mn, mx = c > mx ? c : mx;
    printf("# %d distinct (xcc, se, sh, cu); workgroups per CU %d..%d\n", tuples, mn, mx);
  }
  const size_t strides[] = {4096 + 128, (size_t)64 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)8 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30};
  printf("# ns per dependent load; %u workgroups (one wavefront each), workgroup w starts 256 w bytes into each range\n", nwg);
  printf("%12s", "stride \\ N");
  const uint32_t Ns[] = {4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048, 4096};
  for (uint32_t n : Ns) printf("%7u", n);
  printf("\n");
  for (size_t s : strides) {
    printf("%10zuK ", s >> 10);
    for (uint32_t n : Ns) {
      if ((size_t)n * s + (size_t)nwg * 256 + 4096 > (gb << 30)) {
        printf("%7s", "-");
        continue;
      }
      hipLaunchKernelGGL(k_init, dim3((n * nwg + 255) / 256), dim3(256), 0, 0, buf, s, n, (size_t)256, nwg);
      const uint32_t iters = 20000;
      hipLaunchKernelGGL(k_chase, dim3(nwg), dim3(64), 0, 0, buf, (size_t)256, iters, sink, (uint32_t *)nullptr);   // warm
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_chase, dim3(nwg), dim3(64), 0, 0, buf, (size_t)256, iters, sink, (uint32_t *)nullptr);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%7.0f", ms * 1e6 / iters);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
