// Does a cheap VALU instruction cost issue time when the SIMD is saturated with v_mad_u64_u32?  (Round 3: the lazy-limb change removed
// 175 of 2292 VALU instructions per G1 mixed addition -- all of them single-pass adds / ands / shifts -- and the kernel got ~1 %
// faster, not the 4-5 % an additive cycle model predicts.)  Each variant issues, per loop iteration and wave, 64 multiply-adds on 8
// independent accumulator chains plus K instructions of one other class on independent registers, 3 waves per SIMD:
//     mad only | + 32 v_add_u32 | + 64 v_add_u32 | + 32 v_and_b32 | + 32 v_lshrrev_b64 | + 32 v_mul_lo_u32 | + 32 v_lshl_add_u64
// If the first three take the same time, single-pass instructions ride in the shadow of the multiplier and further instruction
// diets of that class are pointless; whatever raises the time shares the multiplier's issue slots.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mix.hip -o tools/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2000;

#define MAD8(i) asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
#define OTHER4(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x))
#define OTHER4_64(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x))
#define I_ADD(r) "v_add_u32 " #r ", " #r ", %4"
#define I_AND(r) "v_and_b32 " #r ", 0x1fffffff, " #r
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %4"
#define I_SHR64(r) "v_lshrrev_b64 " #r ", 1, " #r
#define I_ADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, " #r

template <int MODE>
__global__ void __launch_bounds__(256, 3) k_mix(uint32_t* out, uint32_t seed) {
  uint32_t x = seed + threadIdx.x, y = seed * 3u + blockIdx.x;
  uint64_t a0 = x, a1 = y, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
  uint32_t c0 = x, c1 = y, c2 = x ^ y, c3 = 11;
  uint64_t d0 = x, d1 = y, d2 = 5, d3 = 9;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {                         // 8 x (8 mads + extras) = 64 mads per iteration
      MAD8(r);
      if (MODE == 1) { OTHER4(I_ADD); }
      if (MODE == 2) { OTHER4(I_ADD); OTHER4(I_ADD); }
      if (MODE == 3) { OTHER4(I_AND); }
      if (MODE == 4) { OTHER4_64(I_SHR64); }
      if (MODE == 5) { OTHER4(I_MULLO); }
      if (MODE == 6) { OTHER4_64(I_ADD64); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ c0 ^ c1 ^ c2 ^ c3 ^ (uint32_t)(d0 ^ d1 ^ d2 ^ d3);
}

template <int MODE>
int run(const char* what, uint32_t* dout, int blocks, double clk) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_mix<MODE>, dim3(blocks), dim3(256), 0, 0, dout, 12345u + rep);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  }
  // per SIMD: 3 waves x ITERS x 64 mads
  printf("%-28s %7.3f ms   %6.2f cycles per multiply-add per SIMD @%.1f GHz nominal\n", what, ms, ms * 1e-3 * clk / (3.0 * ITERS * 64), clk / 1e9);
  return 0;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 3;
  uint32_t* dout;
  CK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
  const double clk = prop.clockRate * 1e3;
  printf("%s, %d CUs, 3 waves per SIMD; every variant: 64 v_mad_u64_u32 per iteration on 8 independent chains\n", prop.gcnArchName, prop.multiProcessorCount);
  run<0>("mad only", dout, blocks, clk);
  run<1>("+ 32 v_add_u32", dout, blocks, clk);
  run<2>("+ 64 v_add_u32", dout, blocks, clk);
  run<3>("+ 32 v_and_b32 (literal)", dout, blocks, clk);
  run<4>("+ 32 v_lshrrev_b64", dout, blocks, clk);
  run<5>("+ 32 v_mul_lo_u32", dout, blocks, clk);
  run<6>("+ 32 v_lshl_add_u64", dout, blocks, clk);
  return 0;
}
