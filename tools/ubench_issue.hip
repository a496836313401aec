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
#include <algorithm>
#include <chrono>
#include <cstdlib>
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
// the same 8 multiply-adds on TWO / THREE accumulators (what a wave of the G1 kernel has: two- and three-chain interleaved products)
#define MAD8_2CH asm volatile( \
    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n" \
    "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n" \
    : "+v"(a0), "+v"(a1) : "v"(x), "v"(y) : "vcc")
#define MAD9_3CH asm volatile( \
    "v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %3, %4, %1\n v_mad_u64_u32 %2, vcc, %3, %4, %2\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n" \
    "v_mad_u64_u32 %1, vcc, %3, %4, %1\n v_mad_u64_u32 %2, vcc, %3, %4, %2\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %1, vcc, %3, %4, %1\n" \
    : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x), "v"(y) : "vcc")
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
#define I_FMA64(r) "v_fma_f64 " #r ", " #r ", %4, %4"        /* f <- f / 2 + 1 / 2: stays finite */
#define I_MULHI24(r) "v_mul_hi_u32_u24 " #r ", " #r ", " #r
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", " #r ", " #r
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", " #r
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", " #r
#define NOP4 asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")

// instructions of the mode's class(es) one wave issues per loop iteration
__host__ __device__ constexpr int per_iter(int mode) {
  return mode == 16 || mode == 17 ? 64 + 12 : mode == 15 ? 64 + 15 + 34 : mode == 9 || mode == 10 || (mode >= 12 && mode <= 14) ? 64 + 14 + 15 : 64;      // (15: + the copies that create the dependence)
}

template <int MODE>
__global__ void __launch_bounds__(256, 3) k_issue(uint32_t* out, unsigned long long* cyc, uint32_t seed, unsigned long long* real) {
  uint32_t x = seed + threadIdx.x, y = seed * 3u + blockIdx.x;
  uint64_t a0 = x, a1 = y, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;
  uint32_t c0 = x, c1 = y, c2 = x ^ y, c3 = 11;
  uint64_t d0 = x, d1 = y, d2 = 5, d3 = 9;
  double f0 = 1.0 + x * 1e-9, f1 = 1.0 + y * 1e-9, f2 = 0.999999, f3 = 1.000001;
  const double fh = 0.5;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();        // the constant 100 MHz counter: the wave's own wall time
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
    } else if constexpr (MODE >= 18 && MODE <= 22) {
      // round 6: the instructions the rejected representations would be made of -- f64 FMA on 52-bit limbs
      // (v_fma_f64, 4 independent chains like OTHER4_64), 24-bit limbs (v_mul_hi_u32_u24 / v_mad_u32_u24 / v_mul_u32_u24), and the
      // high half of a 32 x 32 product (v_mul_hi_u32) that a saturated-limb product needs
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (MODE == 18) asm volatile(I_FMA64(%0) "\n" I_FMA64(%1) "\n" I_FMA64(%2) "\n" I_FMA64(%3) "\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fh));
        if (MODE == 19) OTHER4(I_MULHI24);
        if (MODE == 20) OTHER4(I_MAD24);
        if (MODE == 21) OTHER4(I_MUL24);
        if (MODE == 22) OTHER4(I_MULHI);
      }
    } else if constexpr (MODE == 16 || MODE == 17) {
      // multiply-adds whose multiplicands come from 2 x 9 DIFFERENT registers, like acc += a[i] * b[j] of a product (the modes above
      // multiply the same two registers all the time): does the register file's banking cost issue cycles?  16: 8 chains, 17: 2 chains
      uint32_t av[9], bv[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) { av[i] = x + i * 7u + (uint32_t)it; bv[i] = y ^ (i * 13u); }
      uint64_t acc[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
      if constexpr (MODE == 16) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %17, %0\n" "v_mad_u64_u32 %1, vcc, %9, %22, %1\n" "v_mad_u64_u32 %2, vcc, %10, %18, %2\n" "v_mad_u64_u32 %3, vcc, %11, %23, %3\n" "v_mad_u64_u32 %4, vcc, %12, %19, %4\n" "v_mad_u64_u32 %5, vcc, %13, %24, %5\n" "v_mad_u64_u32 %6, vcc, %14, %20, %6\n" "v_mad_u64_u32 %7, vcc, %15, %25, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %16, %21, %0\n" "v_mad_u64_u32 %1, vcc, %8, %18, %1\n" "v_mad_u64_u32 %2, vcc, %9, %23, %2\n" "v_mad_u64_u32 %3, vcc, %10, %19, %3\n" "v_mad_u64_u32 %4, vcc, %11, %24, %4\n" "v_mad_u64_u32 %5, vcc, %12, %20, %5\n" "v_mad_u64_u32 %6, vcc, %13, %25, %6\n" "v_mad_u64_u32 %7, vcc, %14, %21, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %15, %17, %0\n" "v_mad_u64_u32 %1, vcc, %16, %22, %1\n" "v_mad_u64_u32 %2, vcc, %8, %19, %2\n" "v_mad_u64_u32 %3, vcc, %9, %24, %3\n" "v_mad_u64_u32 %4, vcc, %10, %20, %4\n" "v_mad_u64_u32 %5, vcc, %11, %25, %5\n" "v_mad_u64_u32 %6, vcc, %12, %21, %6\n" "v_mad_u64_u32 %7, vcc, %13, %17, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %14, %22, %0\n" "v_mad_u64_u32 %1, vcc, %15, %18, %1\n" "v_mad_u64_u32 %2, vcc, %16, %23, %2\n" "v_mad_u64_u32 %3, vcc, %8, %20, %3\n" "v_mad_u64_u32 %4, vcc, %9, %25, %4\n" "v_mad_u64_u32 %5, vcc, %10, %21, %5\n" "v_mad_u64_u32 %6, vcc, %11, %17, %6\n" "v_mad_u64_u32 %7, vcc, %12, %22, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %13, %18, %0\n" "v_mad_u64_u32 %1, vcc, %14, %23, %1\n" "v_mad_u64_u32 %2, vcc, %15, %19, %2\n" "v_mad_u64_u32 %3, vcc, %16, %24, %3\n" "v_mad_u64_u32 %4, vcc, %8, %21, %4\n" "v_mad_u64_u32 %5, vcc, %9, %17, %5\n" "v_mad_u64_u32 %6, vcc, %10, %22, %6\n" "v_mad_u64_u32 %7, vcc, %11, %18, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %12, %23, %0\n" "v_mad_u64_u32 %1, vcc, %13, %19, %1\n" "v_mad_u64_u32 %2, vcc, %14, %24, %2\n" "v_mad_u64_u32 %3, vcc, %15, %20, %3\n" "v_mad_u64_u32 %4, vcc, %16, %25, %4\n" "v_mad_u64_u32 %5, vcc, %8, %22, %5\n" "v_mad_u64_u32 %6, vcc, %9, %18, %6\n" "v_mad_u64_u32 %7, vcc, %10, %23, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %11, %19, %0\n" "v_mad_u64_u32 %1, vcc, %12, %24, %1\n" "v_mad_u64_u32 %2, vcc, %13, %20, %2\n" "v_mad_u64_u32 %3, vcc, %14, %25, %3\n" "v_mad_u64_u32 %4, vcc, %15, %21, %4\n" "v_mad_u64_u32 %5, vcc, %16, %17, %5\n" "v_mad_u64_u32 %6, vcc, %8, %23, %6\n" "v_mad_u64_u32 %7, vcc, %9, %19, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %10, %24, %0\n" "v_mad_u64_u32 %1, vcc, %11, %20, %1\n" "v_mad_u64_u32 %2, vcc, %12, %25, %2\n" "v_mad_u64_u32 %3, vcc, %13, %21, %3\n" "v_mad_u64_u32 %4, vcc, %14, %17, %4\n" "v_mad_u64_u32 %5, vcc, %15, %22, %5\n" "v_mad_u64_u32 %6, vcc, %16, %18, %6\n" "v_mad_u64_u32 %7, vcc, %8, %24, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
      } else {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %17, %0\n" "v_mad_u64_u32 %1, vcc, %9, %22, %1\n" "v_mad_u64_u32 %0, vcc, %10, %18, %0\n" "v_mad_u64_u32 %1, vcc, %11, %23, %1\n" "v_mad_u64_u32 %0, vcc, %12, %19, %0\n" "v_mad_u64_u32 %1, vcc, %13, %24, %1\n" "v_mad_u64_u32 %0, vcc, %14, %20, %0\n" "v_mad_u64_u32 %1, vcc, %15, %25, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %16, %21, %0\n" "v_mad_u64_u32 %1, vcc, %8, %18, %1\n" "v_mad_u64_u32 %0, vcc, %9, %23, %0\n" "v_mad_u64_u32 %1, vcc, %10, %19, %1\n" "v_mad_u64_u32 %0, vcc, %11, %24, %0\n" "v_mad_u64_u32 %1, vcc, %12, %20, %1\n" "v_mad_u64_u32 %0, vcc, %13, %25, %0\n" "v_mad_u64_u32 %1, vcc, %14, %21, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %15, %17, %0\n" "v_mad_u64_u32 %1, vcc, %16, %22, %1\n" "v_mad_u64_u32 %0, vcc, %8, %19, %0\n" "v_mad_u64_u32 %1, vcc, %9, %24, %1\n" "v_mad_u64_u32 %0, vcc, %10, %20, %0\n" "v_mad_u64_u32 %1, vcc, %11, %25, %1\n" "v_mad_u64_u32 %0, vcc, %12, %21, %0\n" "v_mad_u64_u32 %1, vcc, %13, %17, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %14, %22, %0\n" "v_mad_u64_u32 %1, vcc, %15, %18, %1\n" "v_mad_u64_u32 %0, vcc, %16, %23, %0\n" "v_mad_u64_u32 %1, vcc, %8, %20, %1\n" "v_mad_u64_u32 %0, vcc, %9, %25, %0\n" "v_mad_u64_u32 %1, vcc, %10, %21, %1\n" "v_mad_u64_u32 %0, vcc, %11, %17, %0\n" "v_mad_u64_u32 %1, vcc, %12, %22, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %13, %18, %0\n" "v_mad_u64_u32 %1, vcc, %14, %23, %1\n" "v_mad_u64_u32 %0, vcc, %15, %19, %0\n" "v_mad_u64_u32 %1, vcc, %16, %24, %1\n" "v_mad_u64_u32 %0, vcc, %8, %21, %0\n" "v_mad_u64_u32 %1, vcc, %9, %17, %1\n" "v_mad_u64_u32 %0, vcc, %10, %22, %0\n" "v_mad_u64_u32 %1, vcc, %11, %18, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %12, %23, %0\n" "v_mad_u64_u32 %1, vcc, %13, %19, %1\n" "v_mad_u64_u32 %0, vcc, %14, %24, %0\n" "v_mad_u64_u32 %1, vcc, %15, %20, %1\n" "v_mad_u64_u32 %0, vcc, %16, %25, %0\n" "v_mad_u64_u32 %1, vcc, %8, %22, %1\n" "v_mad_u64_u32 %0, vcc, %9, %18, %0\n" "v_mad_u64_u32 %1, vcc, %10, %23, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %11, %19, %0\n" "v_mad_u64_u32 %1, vcc, %12, %24, %1\n" "v_mad_u64_u32 %0, vcc, %13, %20, %0\n" "v_mad_u64_u32 %1, vcc, %14, %25, %1\n" "v_mad_u64_u32 %0, vcc, %15, %21, %0\n" "v_mad_u64_u32 %1, vcc, %16, %17, %1\n" "v_mad_u64_u32 %0, vcc, %8, %23, %0\n" "v_mad_u64_u32 %1, vcc, %9, %19, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %10, %24, %0\n" "v_mad_u64_u32 %1, vcc, %11, %20, %1\n" "v_mad_u64_u32 %0, vcc, %12, %25, %0\n" "v_mad_u64_u32 %1, vcc, %13, %21, %1\n" "v_mad_u64_u32 %0, vcc, %14, %17, %0\n" "v_mad_u64_u32 %1, vcc, %15, %22, %1\n" "v_mad_u64_u32 %0, vcc, %16, %18, %0\n" "v_mad_u64_u32 %1, vcc, %8, %24, %1\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(av[0]), "v"(av[1]), "v"(av[2]), "v"(av[3]), "v"(av[4]), "v"(av[5]), "v"(av[6]), "v"(av[7]), "v"(av[8]), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]), "v"(bv[4]), "v"(bv[5]), "v"(bv[6]), "v"(bv[7]), "v"(bv[8]) : "vcc");
      }
      a0 = acc[0]; a1 = acc[1]; a2 = acc[2]; a3 = acc[3]; a4 = acc[4]; a5 = acc[5]; a6 = acc[6]; a7 = acc[7];
    } else if constexpr (MODE >= 12 && MODE <= 15) {
      // the same mix with the multiply-adds on 2 (12), 3 (13) or 1 (14) dependent chains per wave; 15: 2 chains, every other instruction
      // DEPENDS on the chains' low words (the carry splits / m derivations of a Montgomery column read the accumulator)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (MODE == 12 || MODE == 15) { MAD8_2CH; MAD8_2CH; }
        if (MODE == 13) { MAD9_3CH; MAD9_3CH; }
        if (MODE == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n"
                                     : "+v"(a0) : "v"(x), "v"(y) : "vcc");
        if (MODE == 15) {
          d0 = a0; d1 = a1;                                  // the shifts below now wait for the chains
          asm volatile(I_SHR64(%0) "\n" I_SHR64(%1) "\n" : "+v"(d0), "+v"(d1) : "v"(x));
          c0 = (uint32_t)d0; c1 = (uint32_t)d1;
          asm volatile(I_AND(%0) "\n" I_AND(%1) "\n" I_MULLO(%2) "\n" I_ADD(%3) "\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
          x ^= c0 & 1u;                                      // ... and the next multiply-adds for them
        } else {
          asm volatile(I_SHR64(%0) "\n" I_SHR64(%1) "\n" : "+v"(d0), "+v"(d1) : "v"(x));
          asm volatile(I_AND(%0) "\n" I_AND(%1) "\n" I_MULLO(%2) "\n" I_ADD(%3) "\n" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(x));
        }
      }
      asm volatile(I_ADD64(%0) "\n" I_ADD64(%1) "\n" : "+v"(d2), "+v"(d3) : "v"(x));
      asm volatile(I_SHR32(%0) "\n" I_SHR32(%1) "\n" I_SHR32(%2) "\n" : "+v"(c0), "+v"(c1), "+v"(c2) : "v"(x));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x % 64 == 0) { cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0; real[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ c0 ^ c1 ^ c2 ^ c3 ^ (uint32_t)(d0 ^ d1 ^ d2 ^ d3) ^ (uint32_t)__double_as_longlong(f0 + f1 + f2 + f3);
}

// g_seconds > 0 (argv: <mode> <seconds>): keep launching ONE mode for that long, so that a power / clock sampler beside the process
// (tools/power_trace.py) sees a steady state; the line then reports the LAST launch
static double g_seconds = 0;
static int g_only = -1;
// argv[3] = CUs: run on a stream restricted to that many CUs (hipExtStreamCreateWithCUMask; 3 workgroups = 3 waves per SIMD on each of
// them, the rest of the chip idle).  Round 6: an instruction costs the same cycles on 8 CUs as on 256 -- its price is the SIMD's own
// rate for its class, not a chip-wide power or current limit.
static hipStream_t g_stream = nullptr;
static unsigned long long* g_real = nullptr;
template <int MODE>
int run(const char* what, uint32_t* dout, unsigned long long* dcyc, int blocks) {
  if (g_only >= 0 && g_only != MODE) return 0;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms = 0;
  const auto t_begin = std::chrono::steady_clock::now();
  for (int rep = 0; rep < 3 || std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < g_seconds; ++rep) {
    CK(hipEventRecord(a, g_stream));
    hipLaunchKernelGGL(k_issue<MODE>, dim3(blocks), dim3(256), 0, g_stream, dout, dcyc, 12345u + rep, g_real);
    CK(hipEventRecord(b, g_stream)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  }
  const int waves = blocks * 4;
  std::vector<unsigned long long> h(waves);
  CK(hipMemcpy(h.data(), dcyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double wave_cycles = sum / waves;                                 // one wave's span; its SIMD carried 3 such waves in it
  std::vector<unsigned long long> hr(waves);
  CK(hipMemcpy(hr.data(), g_real, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double rsum = 0, cmax = 0, cmin = 1e300;
  for (auto v : hr) rsum += (double)v;
  for (auto v : h) { cmax = std::max(cmax, (double)v); cmin = std::min(cmin, (double)v); }
  const double wave_ns = rsum / waves * 10.0;                               // s_memrealtime ticks at 100 MHz
  const double ghz = wave_cycles / wave_ns;                                 // the clock s_memtime really ticks at: 2.38-2.40 GHz in every loop
  const double instr = 3.0 * ITERS * per_iter(MODE);                        // instructions a SIMD's three waves issue
  // Round 6: the three waves of a SIMD do NOT finish together -- issue is oldest-first, the first wave is done after ~40 % of the
  // kernel, the second after ~70 % (tools/ubench_placement.hip) -- so only the LONGEST span (= the kernel) covers all three waves'
  // instructions.  Round 5 divided the MEAN span by them ("3.15 real cycles per v_mad_u64_u32") and then the mean span by the
  // kernel's time ("1.2-1.6 GHz"): both wrong.  Cycles per instruction per SIMD = longest span / instructions; the clock is sclk.
  printf("%-44s kernel %7.3f ms | %6.3f cycles per instruction per SIMD (longest wave span %.0f cycles / %.0f instructions; kernel time x %.2f GHz gives %.3f) | "
         "first wave done after %.0f %%, mean wave after %.0f %% of the longest span | s_memtime / s_memrealtime = %.2f GHz\n",
         what, ms, cmax / instr, cmax, instr, ghz, ms * 1e6 * ghz / instr, 100.0 * cmin / cmax, 100.0 * wave_cycles / cmax, ghz);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3) { g_only = atoi(argv[1]); g_seconds = atof(argv[2]); }
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int blocks = prop.multiProcessorCount * 3;                // 12 waves per CU = 3 per SIMD, one round
  if (argc >= 4 && atoi(argv[3]) > 0) {
    const int cus = atoi(argv[3]);
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
    // the FIRST `cus` bits.  (Until late in round 6 the bits were spread over the mask, one every 256 / cus: tools/ubench_placement.hip
    // `masked` shows that such a mask does NOT confine the launch -- 8 spread bits put the 24 workgroups on 24 CUs, one wave per SIMD --
    // so the "1.8-1.9 cycles per instruction on 8 / 32 CUs" of that form were a lone wave's 5.5-5.8 cycles divided by three.  With
    // the first bits the waves sit 3 per SIMD on exactly `cus` CUs, and every class costs what it costs on the full chip.)
    for (int i = 0; i < cus; ++i) mask[i / 32] |= 1u << (i % 32);
    CK(hipExtStreamCreateWithCUMask(&g_stream, (uint32_t)mask.size(), mask.data()));
    blocks = cus * 3;
    printf("stream restricted to %d of %d CUs, %d workgroups\n", cus, prop.multiProcessorCount, blocks);
  }
  uint32_t* dout;
  unsigned long long* dcyc;
  CK(hipMalloc(&dout, (size_t)blocks * 256 * 4));
  CK(hipMalloc(&dcyc, (size_t)blocks * 4 * 8));
  CK(hipMalloc(&g_real, (size_t)blocks * 4 * 8));
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
  run<12>("mix, multiply-adds on 2 chains per wave", dout, dcyc, blocks);
  run<13>("mix, multiply-adds on 3 chains per wave", dout, dcyc, blocks);
  run<14>("mix, multiply-adds on 1 chain per wave", dout, dcyc, blocks);
  run<15>("mix, 2 chains, the other ops depend on them", dout, dcyc, blocks);
  run<16>("64 mad on 8 chains, multiplicands from 2 x 9 registers (+ 18 setup ops)", dout, dcyc, blocks);
  run<17>("64 mad on 2 chains, multiplicands from 2 x 9 registers (+ 18 setup ops)", dout, dcyc, blocks);
  run<18>("v_fma_f64 (4 chains)", dout, dcyc, blocks);
  run<19>("v_mul_hi_u32_u24 (VOP2)", dout, dcyc, blocks);
  run<20>("v_mad_u32_u24 (VOP3)", dout, dcyc, blocks);
  run<21>("v_mul_u32_u24 (VOP2)", dout, dcyc, blocks);
  run<22>("v_mul_hi_u32 (VOP3)", dout, dcyc, blocks);
  return 0;
}
