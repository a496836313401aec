// Micro-benchmark: per-instruction VALU issue rates on gfx950 that decide the
// big-integer representation (32-bit mad chains vs 24-bit vs FP64-FMA limbs).
// Build: hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITERS = 2000;
constexpr int UNROLL = 8;   // independent chains

#define KERNEL(name, DECL, BODY, SINK)                                         \
__global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {   \
  DECL;                                                                        \
  for (int it = 0; it < ITERS; ++it) {                                         \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { BODY; }               \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { BODY; }               \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { BODY; }               \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { BODY; }               \
  }                                                                            \
  SINK;                                                                        \
}

// 64-bit accumulators a[u], multipliers x,y
#define DECL64 uint64_t a[UNROLL]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; \
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u;
#define SINK64 { uint64_t s = 0; for (int u = 0; u < UNROLL; ++u) s ^= a[u]; if (s == 0x1234567) out[0] = (uint32_t)s; }
#define DECL32 uint32_t a[UNROLL]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; \
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u;
#define SINK32 { uint32_t s = 0; for (int u = 0; u < UNROLL; ++u) s ^= a[u]; if (s == 0x1234567) out[0] = s; }
#define DECLF64 double a[UNROLL]; double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-7 * seed; \
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u;
#define SINKF64 { double s = 0; for (int u = 0; u < UNROLL; ++u) s += a[u]; if (s == 0.1234567) out[0] = (uint32_t)s; }

KERNEL(k_mad_u64_u32, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[u]) : "v"(x), "v"(y) : "vcc"), SINK64)
KERNEL(k_mul_lo_u32, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_mul_hi_u32, DECL32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_mad_u32_u24, DECL32, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_mul_hi_u32_u24, DECL32, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_add_u32, DECL32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_add_co_u32, DECL32, asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[u]) : "v"(x) : "vcc"), SINK32)
KERNEL(k_addc_co_u32, DECL32, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[u]) : "v"(x) : "vcc"), SINK32)
KERNEL(k_add3_u32, DECL32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_lshl_add_u64, DECL64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[u]) : "v"((uint64_t)x)), SINK64)
KERNEL(k_mov_b32, DECL32, asm volatile("v_mov_b32 %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_cndmask, DECL32, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(x) : "vcc"), SINK32)
KERNEL(k_fma_f64, DECLF64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINKF64)
KERNEL(k_mul_f64, DECLF64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINKF64)
KERNEL(k_add_f64, DECLF64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[u]) : "v"(y)), SINKF64)
KERNEL(k_fma_f32, DECL32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_mad_u64_u32_dep, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y) : "vcc"), SINK64)
KERNEL(k_mad_i32_i24, DECL32, asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_mul_u32_u24, DECL32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_dot4_u8, DECL32, asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; };

int main(int argc, char** argv) {
  uint32_t* d; CK(hipMalloc(&d, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  std::vector<Entry> es = {
    {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(dep chain)", k_mad_u64_u32_dep},
    {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
    {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_i32_i24", k_mad_i32_i24}, {"v_mul_u32_u24", k_mul_u32_u24},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24}, {"v_dot4_u32_u8", k_dot4_u8},
    {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32}, {"v_addc_co_u32", k_addc_co_u32},
    {"v_add3_u32", k_add3_u32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mov_b32", k_mov_b32},
    {"v_cndmask_b32", k_cndmask}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
    {"v_fma_f32", k_fma_f32},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wpc : {4, 8, 16, 32}) {     // waves per CU
    int blocks = prop.multiProcessorCount * wpc / 4;   // 256 threads = 4 waves
    printf("--- %d waves/CU (%d blocks of 256)\n", wpc, blocks);
    for (auto& e : es) {
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double insts_per_wave = (double)ITERS * UNROLL * 4;
      double waves = (double)blocks * 4;
      double wave_insts = insts_per_wave * waves;
      // cycles per wave-instruction per SIMD at the nominal 2.4 GHz
      double simds = prop.multiProcessorCount * 4.0;
      double cyc = (ms * 1e-3 * 2.4e9) * simds / wave_insts;
      printf("%-26s %8.3f ms  %7.2f Tlane-op/s  %6.2f cyc/wave-inst/SIMD@2.4GHz\n", e.name, ms,
             wave_insts * 64 / (ms * 1e-3) / 1e12, cyc);
    }
  }
  return 0;
}
