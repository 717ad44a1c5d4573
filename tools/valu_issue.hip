// valu_issue.hip -- how many wave64 VALU instructions does one gfx950 SIMD issue per cycle?
//
// DESIGN.md took the ceiling of the integer kernels as 0.25 wavefront-instructions per cycle and
// SIMD (a wave64 operation holding a SIMD16 for four cycles); MI355X_MICROARCH.md says the SIMDs are 32 wide (two cycles).
// This measures it per opcode: S independent register streams per wave (S = 8: no dependent issue; S = 1: a dependent
// chain), 1 / 2 / 4 / 8 waves per SIMD, every CU loaded.  Cycles = s_memtime delta of the launch's slowest wave (the tick is
// the shader cycle); the wall clock of the launch is printed beside it.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_issue.hip -o tools/build/valu_issue && tools/build/valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one instruction of the op under test on register R (reads R and the loop-invariant B, C; writes R)
#define OP_ADD(R)      "v_add_u32 " R ", " R ", %[b]\n"
#define OP_LSHLADD(R)  "v_lshl_add_u32 " R ", " R ", 3, %[b]\n"
#define OP_MIN(R)      "v_min_u32 " R ", " R ", %[b]\n"
#define OP_MIN3(R)     "v_min3_u32 " R ", " R ", %[b], %[c]\n"
#define OP_XOR(R)      "v_xor_b32 " R ", " R ", %[b]\n"
#define OP_ANDOR(R)    "v_and_or_b32 " R ", " R ", %[b], %[c]\n"
#define OP_CNDMASK(R)  "v_cndmask_b32 " R ", " R ", %[b], vcc\n"
#define OP_PERM(R)     "v_perm_b32 " R ", " R ", %[b], %[c]\n"
#define OP_ALIGNBIT(R) "v_alignbit_b32 " R ", " R ", %[b], 7\n"
#define OP_ALIGNBYTE(R) "v_alignbyte_b32 " R ", " R ", %[b], %[c]\n"
#define OP_FFBL(R)     "v_ffbl_b32 " R ", " R "\n"
#define OP_BFREV(R)    "v_bfrev_b32 " R ", " R "\n"
#define OP_LSHRREV(R)  "v_lshrrev_b32 " R ", %[c], " R "\n"
#define OP_MULLO(R)    "v_mul_lo_u32 " R ", " R ", %[b]\n"
#define OP_MAD24(R)    "v_mad_u32_u24 " R ", " R ", %[b], %[c]\n"
#define OP_FMA(R)      "v_fma_f32 " R ", " R ", %[b], %[c]\n"
#define OP_PKADD16(R)  "v_pk_add_u16 " R ", " R ", %[b]\n"
#define OP_PKMAX16(R)  "v_pk_max_u16 " R ", " R ", %[b]\n"
#define OP_PKMIN16(R)  "v_pk_min_u16 " R ", " R ", %[b]\n"
#define OP_PKFMA32(R)  "v_pk_fma_f32 " R ", " R ", %[b2], %[c2]\n"   /* (64-bit operands: see k_pkfma) */
#define OP_CMP(R)      "v_cmp_lt_u32 vcc, " R ", %[b]\n"
#define OP_DPPMAX(R)   "v_max_i32_dpp " R ", " R ", " R " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP_MOVDPP(R)   "v_mov_b32_dpp " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_SDWA(R)     "v_xor_b32_sdwa " R ", " R ", %[b] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define OP_BPERM(R)    "ds_bpermute_b32 " R ", %[b], " R "\n"
#define OP_READLANE(R) "v_readlane_b32 s20, " R ", 3\n"

#define REP8(OP) OP("%[r0]") OP("%[r1]") OP("%[r2]") OP("%[r3]") OP("%[r4]") OP("%[r5]") OP("%[r6]") OP("%[r7]")
#define CHAIN8(OP) OP("%[r0]") OP("%[r0]") OP("%[r0]") OP("%[r0]") OP("%[r0]") OP("%[r0]") OP("%[r0]") OP("%[r0]")

#define DEFINE_KERNEL(NAME, BODY, TAIL)                                                                                   \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, unsigned long long *cyc, int iters) {                         \
    uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    uint32_t b = (threadIdx.x * 4u) & 0xfcu, c = 1u + (blockIdx.x & 1u);                                                   \
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(r0), "v"(b) : "vcc");                                                   \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                           \
    for (int i = 0; i < iters; ++i) {                                                                                     \
      asm volatile(BODY BODY BODY BODY TAIL                                                                                \
                   : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [r4] "+v"(r4), [r5] "+v"(r5),              \
                     [r6] "+v"(r6), [r7] "+v"(r7)                                                                          \
                   : [b] "v"(b), [c] "v"(c)                                                                                \
                   : "vcc", "s20");                                                                                        \
    }                                                                                                                     \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                           \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                      \
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                          \
  }

#define BOTH(NAME, OP)                                  \
  DEFINE_KERNEL(k_##NAME##_ind, REP8(OP), "")           \
  DEFINE_KERNEL(k_##NAME##_dep, CHAIN8(OP), "")

BOTH(add, OP_ADD)
BOTH(lshladd, OP_LSHLADD)
BOTH(min, OP_MIN)
BOTH(min3, OP_MIN3)
BOTH(xor_, OP_XOR)
BOTH(andor, OP_ANDOR)
BOTH(cndmask, OP_CNDMASK)
BOTH(perm, OP_PERM)
BOTH(alignbit, OP_ALIGNBIT)
BOTH(alignbyte, OP_ALIGNBYTE)
BOTH(ffbl, OP_FFBL)
BOTH(bfrev, OP_BFREV)
BOTH(lshrrev, OP_LSHRREV)
BOTH(mullo, OP_MULLO)
BOTH(mad24, OP_MAD24)
BOTH(fma, OP_FMA)
BOTH(pkadd16, OP_PKADD16)
BOTH(pkmax16, OP_PKMAX16)
BOTH(pkmin16, OP_PKMIN16)
BOTH(cmp, OP_CMP)
BOTH(dppmax, OP_DPPMAX)
BOTH(movdpp, OP_MOVDPP)
BOTH(sdwa, OP_SDWA)
DEFINE_KERNEL(k_bperm_ind, REP8(OP_BPERM), "s_waitcnt lgkmcnt(0)\n")
DEFINE_KERNEL(k_readlane_ind, REP8(OP_READLANE), "")

// v_pk_fma_f32: two f32 FMAs per lane and instruction (the datasheet's 157 TF vector peak counts it)
__global__ __launch_bounds__(256) void k_pkfma_ind(uint32_t *out, unsigned long long *cyc, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 r0 = {1.f, 2.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
  f2 b2 = {1.0001f, 0.9999f}, c2 = {(float)threadIdx.x, 1.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8(OP_PKFMA32) REP8(OP_PKFMA32) REP8(OP_PKFMA32) REP8(OP_PKFMA32)
                 : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [r4] "+v"(r4), [r5] "+v"(r5), [r6] "+v"(r6), [r7] "+v"(r7)
                 : [b2] "v"(b2), [c2] "v"(c2));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  f2 s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(s.x + s.y);
}

// a mix shaped like the integer kernels of this repository (hash chain / min-max chains): 2 independent streams of dependent ops
DEFINE_KERNEL(k_mix2_dep,
              "v_lshl_add_u32 %[r0], %[r0], 3, %[b]\n v_lshl_add_u32 %[r1], %[r1], 3, %[b]\n v_xor_b32 %[r0], %[r0], %[c]\n v_xor_b32 %[r1], %[r1], %[c]\n"
              "v_min_u32 %[r0], %[r0], %[b]\n v_min_u32 %[r1], %[r1], %[b]\n v_cndmask_b32 %[r0], %[r0], %[c], vcc\n v_cndmask_b32 %[r1], %[r1], %[c], vcc\n", "")

typedef void (*kern_t)(uint32_t *, unsigned long long *, int);
struct Entry { const char *name; kern_t ind, dep; };

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("# device %s, %d CUs, clockRate %d kHz; %d loop iterations x 32 instructions per wave\n", prop.name, ncu, prop.clockRate, iters);
  printf("# rate = wave64 instructions per cycle per SIMD (cycles: s_memtime of the slowest wave); ind = 8 independent registers, dep = one dependent chain\n");
  printf("# a SIMD that holds a wave64 op for 4 cycles tops out at 0.25, for 2 cycles at 0.50\n");
  uint32_t *out; unsigned long long *cyc;
  CHECK(hipMalloc(&out, (size_t)ncu * 8 * 256 * 4));
  CHECK(hipMalloc(&cyc, (size_t)ncu * 8 * 4 * 8));
  std::vector<unsigned long long> h((size_t)ncu * 8 * 4);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define E(n) {#n, k_##n##_ind, k_##n##_dep}
  Entry tab[] = {E(add), E(lshladd), E(min), E(min3), E(xor_), E(andor), E(cndmask), E(perm), E(alignbit), E(alignbyte), E(ffbl), E(bfrev),
                 E(lshrrev), E(mullo), E(mad24), E(fma), E(pkadd16), E(pkmax16), E(pkmin16), E(cmp), E(dppmax), E(movdpp), E(sdwa),
                 {"pk_fma_f32", k_pkfma_ind, nullptr}, {"ds_bpermute", k_bperm_ind, nullptr}, {"readlane", k_readlane_ind, nullptr},
                 {"mix2(lshl_add,xor,min,cndmask)", nullptr, k_mix2_dep}};
  printf("%-32s %-4s", "op", "form");
  for (int k : {1, 2, 4, 8}) printf("  %dw/SIMD: rate   GHz ", k);
  printf("\n");
  for (const Entry &en : tab) {
    for (int form = 0; form < 2; ++form) {
      kern_t f = form == 0 ? en.ind : en.dep;
      if (!f) continue;
      printf("%-32s %-4s", en.name, form == 0 ? "ind" : "dep");
      for (int k : {1, 2, 4, 8}) {
        const int grid = ncu * k;
        hipLaunchKernelGGL(f, dim3(grid), dim3(256), 0, 0, out, cyc, 200);  // warm-up
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(f, dim3(grid), dim3(256), 0, 0, out, cyc, iters);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), cyc, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost));
        const unsigned long long mx = *std::max_element(h.begin(), h.begin() + (size_t)grid * 4);
        const double rate = (double)iters * 32.0 * k / (double)mx;
        printf("  %14.3f %6.2f", rate, (double)mx / (ms * 1e6));
      }
      printf("\n");
    }
  }
  return 0;
}
