// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count for the access patterns of this repository's kernels?
//
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced stream; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  The alignment kernels read 8 and 16 unaligned
// bytes per lane at data-dependent offsets, so bench.py reported their traffic twice (raw and x2).  This program issues loads
// whose footprint is known exactly -- every offset is a hash of the load's index, recomputed on the host, where the distinct
// 32-, 64- and 128-byte granules they touch are counted -- in one kernel per pattern over a 2 GiB buffer (8x the Infinity
// Cache, every granule touched at most a handful of times), so that
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/build/fetch_calib
// gives FETCH_SIZE per pattern to hold against the three counts (tools/fetch_calib.sh does that).
//   k_cal_stream16  : 16 B per lane, coalesced, the whole buffer once             (the guide's calibrated case: expect 1/2)
//   k_cal_rand16    : 16 unaligned bytes per lane at independent random offsets   (a lane-private stream)
//   k_cal_rand8     : 8 unaligned bytes per lane at independent random offsets
//   k_cal_probe8x8  : 8-lane groups, lane j reads 8 bytes at base + j            (k_align_ph's probe: neighbouring diagonals)
//   k_cal_snake8x16 : 8-lane groups, lane j reads 16 bytes at base + 16 j        (k_align_ph's 128-code extension)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ inline uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline uint64_t offset_of(uint64_t i, uint64_t span, uint64_t salt) { return mix(i * 2 + salt) % span; }

__global__ void k_cal_stream16(const uint4 *__restrict__ buf, size_t n16, uint32_t *out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = buf[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int BYTES>
__global__ void k_cal_rand(const uint8_t *__restrict__ buf, uint64_t span, uint64_t nloads, uint32_t *out) {
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nloads; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint8_t *p = buf + offset_of(i, span, BYTES);
    if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, p, 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    else { uint2 v; __builtin_memcpy(&v, p, 8); acc ^= v.x ^ v.y; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// groups of 8 lanes: group g reads at base(g) + STRIDE * lane_in_group, BYTES each
template <int BYTES, int STRIDE>
__global__ void k_cal_group(const uint8_t *__restrict__ buf, uint64_t span, uint64_t ngroups, uint32_t *out) {
  uint32_t acc = 0;
  const uint64_t lane = threadIdx.x & 7;
  for (uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; g < ngroups; g += ((uint64_t)gridDim.x * blockDim.x) >> 3) {
    const uint8_t *p = buf + offset_of(g, span, 100 + BYTES + STRIDE) + STRIDE * lane;
    if (BYTES == 16) { uint4 v; __builtin_memcpy(&v, p, 16); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    else { uint2 v; __builtin_memcpy(&v, p, 8); acc ^= v.x ^ v.y; }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

struct Granules {
  std::vector<uint64_t> b32, b64, b128;
  explicit Granules(size_t bytes) : b32((bytes / 32 + 63) / 64), b64((bytes / 64 + 63) / 64), b128((bytes / 128 + 63) / 64) {}
  void touch(uint64_t off, int len) {
    for (uint64_t a = off / 32; a <= (off + len - 1) / 32; ++a) b32[a >> 6] |= 1ULL << (a & 63);
    for (uint64_t a = off / 64; a <= (off + len - 1) / 64; ++a) b64[a >> 6] |= 1ULL << (a & 63);
    for (uint64_t a = off / 128; a <= (off + len - 1) / 128; ++a) b128[a >> 6] |= 1ULL << (a & 63);
  }
  static uint64_t pop(const std::vector<uint64_t> &v) { uint64_t s = 0; for (uint64_t w : v) s += __builtin_popcountll(w); return s; }
  void report(const char *name, uint64_t requested) const {
    printf("CAL %-16s requested_bytes %llu distinct32 %llu distinct64 %llu distinct128 %llu\n", name, (unsigned long long)requested,
           (unsigned long long)(pop(b32) * 32), (unsigned long long)(pop(b64) * 64), (unsigned long long)(pop(b128) * 128));
  }
};

int main(int argc, char **argv) {
  const size_t bytes = (size_t)2 << 30;
  const uint64_t span = bytes - 4096;
  const uint64_t nloads = argc > 1 ? strtoull(argv[1], nullptr, 10) : (4ULL << 20);
  uint8_t *buf; uint32_t *out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0x5a, bytes));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 16, block = 256;

  hipLaunchKernelGGL(k_cal_stream16, dim3(grid), dim3(block), 0, 0, (const uint4 *)buf, bytes / 16, out);
  CHECK(hipDeviceSynchronize());
  printf("CAL %-16s requested_bytes %llu distinct32 %llu distinct64 %llu distinct128 %llu\n", "k_cal_stream16", (unsigned long long)bytes,
         (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes);

  hipLaunchKernelGGL((k_cal_rand<16>), dim3(grid), dim3(block), 0, 0, buf, span, nloads, out);
  CHECK(hipDeviceSynchronize());
  { Granules g(bytes); for (uint64_t i = 0; i < nloads; ++i) g.touch(offset_of(i, span, 16), 16); g.report("k_cal_rand<16>", nloads * 16); }

  hipLaunchKernelGGL((k_cal_rand<8>), dim3(grid), dim3(block), 0, 0, buf, span, nloads, out);
  CHECK(hipDeviceSynchronize());
  { Granules g(bytes); for (uint64_t i = 0; i < nloads; ++i) g.touch(offset_of(i, span, 8), 8); g.report("k_cal_rand<8>", nloads * 8); }

  const uint64_t ngroups = nloads / 8;
  hipLaunchKernelGGL((k_cal_group<8, 1>), dim3(grid), dim3(block), 0, 0, buf, span, ngroups, out);
  CHECK(hipDeviceSynchronize());
  { Granules g(bytes); for (uint64_t i = 0; i < ngroups; ++i) g.touch(offset_of(i, span, 100 + 8 + 1), 8 + 7); g.report("k_cal_group<8, 1>", ngroups * 64); }

  hipLaunchKernelGGL((k_cal_group<16, 16>), dim3(grid), dim3(block), 0, 0, buf, span, ngroups, out);
  CHECK(hipDeviceSynchronize());
  { Granules g(bytes); for (uint64_t i = 0; i < ngroups; ++i) g.touch(offset_of(i, span, 100 + 16 + 16), 128); g.report("k_cal_group<16, 16>", ngroups * 128); }
  return 0;
}
