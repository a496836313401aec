// What does one wave64 instruction of each class cost a SIMD in REAL shader cycles?  (`roofline_issue` priced every
// VALU instruction at 4 cycles of the NOMINAL clock; tools/ubench_valu*.hip printed "cycles @2.4 GHz nominal", i.e. wall time x a clock
// the chip does not run at under load.)  Here every wave brackets its loop with s_memtime (the shader clock: /opt/skills/guides/
// MI355X_MICROARCH.md) and the host also times the launch, so each line gives
//     real cycles per instruction per SIMD  =  (t1 - t0) of one wave / (instructions the SIMD's 3 waves issued in that span)
//     effective clock                        =  those cycles / wall time
// Classes: v_mad_u64_u32 (8 chains) | v_lshrrev_b64, v_mul_lo_u32, v_lshl_add_u64, v_add3_u32, v_bfe_u32 (VOP3) | v_and_b32, v_add_u32,
// v_lshrrev_b32 (VOP2) | the mix of the G1 mixed addition's common path as tools/isa_histogram.py counts it (1474 mad : 313 VOP3 :
// 337 VOP2, with and without its 249 s_nop) | s_nop alone.  3 waves per SIMD, like k_bucket_accumulate<G1>.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 1000;

#define MAD8 asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
// ... with the density of padding the compiler leaves in the G1 kernel (254 s_nop 0 for 2511 multiply-adds: about one per ten)
#define MAD8_NOP asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n s_nop 0\n" \
    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
#define OTHER4(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x))
#define OTHER4_64(INS) asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(x))
#define I_ADD(r) "v_add_u32 " #r ", " #r ", " #r
#define I_AND(r) "v_and_b32 " #r ", 0x1fffffff, " #r
#define I_SHR32(r) "v_lshrrev_b32 " #r ", 1, " #r
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", " #r
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", " #r ", " #r
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 1, 29"
#define I_SHR64(r) "v_lshrrev_b64 " #r ", 1, " #r
#define I_ADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, " #r
#define NOP4 asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")

// instructions of the mode's class(es) one wave issues per loop iteration
__host__ __device__ constexpr int per_iter(int mode) {
  return mode == 0 ? 64 : mode <= 8 ? 64 : mode == 9 ? 64 + 14 + 15 : mode == 10 ? 64 + 14 + 15 : mode == 11 ? 64 : 64;
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) k_issue(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
  uint32_t x = seed + threadIdx.x, y = seed * 3u + blockIdx.x;
  uint64_t a0 = x, a1 = y, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
  uint32_t c0 = x, c1 = y, c2 = x ^ y, c3 = 11;
  uint64_t d0 = x, d1 = y, d2 = 5, d3 = 9;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) MAD8;
    } else if constexpr (MODE >= 1 && MODE <= 8) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MODE == 1) OTHER4_64(I_SHR64);
        if (MODE == 2) OTHER4(I_MULLO);
        if (MODE == 3) OTHER4_64(I_ADD64);
        if (MODE == 4) OTHER4(I_ADD3);
        if (MODE == 5) OTHER4(I_BFE);
        if (MODE == 6) OTHER4(I_AND);
        if (MODE == 7) OTHER4(I_ADD);
        if (MODE == 8) OTHER4(I_SHR32);
      }
    } else if constexpr (MODE == 9 || MODE == 10) {
      // the G1 addition's mix, scaled to 64 mads: 1474 : 313 : 337 : 249  ->  64 : 13.6 : 14.6 : 10.8  (14 VOP3 = 8 shr64 + 4 mul_lo + 2 add64; 15 VOP2 = 8 and + 4 add + 3 shr32)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == 9) { MAD8; MAD8; } else { MAD8_NOP; MAD8_NOP; }
        asm volatile(I_SHR64(%0) "\n" I_SHR64(%1) "\n" : "+v"(d0), "+v"(d1) : "v"(x));
        asm volatile(I_AND(%0) "\n" I_AND(%1) "\n" I_MULLO(%2) "\n" I_ADD(%3) "\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
      }
      asm volatile(I_ADD64(%0) "\n" I_ADD64(%1) "\n" : "+v"(d2), "+v"(d3) : "v"(x));
      asm volatile(I_SHR32(%0) "\n" I_SHR32(%1) "\n" I_SHR32(%2) "\n" : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(x));
    } else if constexpr (MODE == 11) {
#pragma unroll
      for (int r = 0; r < 16; ++r) NOP4;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ c0 ^ c1 ^ c2 ^ c3 ^ (uint32_t)(d0 ^ d1 ^ d2 ^ d3);
}

template <int MODE>
int run(const char* what, uint32_t* dout, unsigned long long* dcyc, int blocks) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, 12345u + rep);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  }
  const int waves = blocks * 4;
  std::vector<unsigned long long> h(waves);
  CK(hipMemcpy(h.data(), dcyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double wave_cycles = sum / waves;                                 // one wave's span; its SIMD carried 3 such waves in it
  const double per_instr = wave_cycles / (3.0 * ITERS * per_iter(MODE));
  printf("%-44s %7.3f ms  %7.3f real cycles per instruction per SIMD  (wave span %.0f cycles -> %.2f GHz effective)\n", what, ms, per_instr, wave_cycles,
         wave_cycles / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 3;          // 12 waves per CU = 3 per SIMD, one round
  uint32_t* dout;
  unsigned long long* dcyc;
  CK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
  CK(hipMalloc(&dcyc, (size_t)blocks * 4 * 8));
  printf("%s, %d CUs, 3 waves per SIMD, %d iterations; cycles from s_memtime, clock = span / wall time\n", prop.gcnArchName, prop.multiProcessorCount, ITERS);
  run<0>("v_mad_u64_u32 (8 chains)", dout, dcyc, blocks);
  run<1>("v_lshrrev_b64 (VOP3)", dout, dcyc, blocks);
  run<2>("v_mul_lo_u32 (VOP3)", dout, dcyc, blocks);
  run<3>("v_lshl_add_u64 (VOP3)", dout, dcyc, blocks);
  run<4>("v_add3_u32 (VOP3)", dout, dcyc, blocks);
  run<5>("v_bfe_u32 (VOP3)", dout, dcyc, blocks);
  run<6>("v_and_b32 literal (VOP2)", dout, dcyc, blocks);
  run<7>("v_add_u32 (VOP2)", dout, dcyc, blocks);
  run<8>("v_lshrrev_b32 (VOP2)", dout, dcyc, blocks);
  run<9>("G1-addition mix 64 mad : 14 VOP3 : 15 VOP2", dout, dcyc, blocks);
  run<10>("same mix + one s_nop 0 per eight mads", dout, dcyc, blocks);
  run<11>("s_nop 0", dout, dcyc, blocks);
  return 0;
}
