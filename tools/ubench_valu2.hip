// Micro-benchmark, part 2: issue cost of the NON-multiply instructions of the Montgomery / point-addition code on gfx950
// (64-bit shifts, masks, selects), to decide which of them are worth removing.  Same harness as tools/ubench_valu.hip.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_valu2.hip -o ubench_valu2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITERS = 2000;
constexpr int UNROLL = 8;

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
#define DECL64 uint64_t a[UNROLL]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; \
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u;
#define SINK64 { uint64_t s = 0; for (int u = 0; u < UNROLL; ++u) s ^= a[u]; if (s == 0x1234567) out[0] = (uint32_t)s; }
#define DECL32 uint32_t a[UNROLL]; uint32_t x = seed + threadIdx.x, y = seed * 3 + 1; \
  for (int u = 0; u < UNROLL; ++u) a[u] = seed + u;
#define SINK32 { uint32_t s = 0; for (int u = 0; u < UNROLL; ++u) s ^= a[u]; if (s == 0x1234567) out[0] = s; }
// condition in an SGPR pair computed once outside the loop
#define DECLC DECL32 uint64_t cond = __ballot((threadIdx.x ^ seed) & 1);

KERNEL(k_mad_ref, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[u]) : "v"(x), "v"(y) : "vcc"), SINK64)
KERNEL(k_lshrrev_b64, DECL64, asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(a[u])), SINK64)
KERNEL(k_lshlrev_b64, DECL64, asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[u])), SINK64)
KERNEL(k_alignbit, DECL32, asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_and, DECL32, asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(a[u])), SINK32)
KERNEL(k_bfe, DECL32, asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a[u])), SINK32)
KERNEL(k_lshrrev_b32, DECL32, asm volatile("v_lshrrev_b32 %0, 29, %0" : "+v"(a[u])), SINK32)
KERNEL(k_xad, DECL32, asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_and_or, DECL32, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "v"(y)), SINK32)
KERNEL(k_lshl_or, DECL32, asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_sub, DECL32, asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_cndmask_sgpr, DECLC, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(x), "s"(cond)), SINK32)
KERNEL(k_cndmask_vcc_set, DECLC, asm volatile("s_mov_b64 vcc, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(x), "s"(cond) : "vcc"), SINK32)
KERNEL(k_cmp_cndmask, DECL32, asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[u]) : "v"(x) : "vcc"), SINK32)
KERNEL(k_mov, DECL32, asm volatile("v_mov_b32 %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_mul_lo, DECL32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[u]) : "v"(x)), SINK32)
KERNEL(k_mad_sgpr, DECL64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[u]) : "v"(x), "s"(seed) : "vcc"), SINK64)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int insts; };

int main() {
  uint32_t* d; CK(hipMalloc(&d, 4096));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  std::vector<Entry> es = {
    {"v_mul_lo_u32", k_mul_lo, 1},
    {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_alignbit_b32", k_alignbit, 1},
    {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_and_b32 (literal)", k_and, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_xad_u32", k_xad, 1},
    {"v_and_or_b32", k_and_or, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_sub_u32", k_sub, 1}, {"v_mov_b32", k_mov, 1},
    {"v_cndmask_b32 (sgpr cond)", k_cndmask_sgpr, 1}, {"s_mov vcc + v_cndmask_b32", k_cndmask_vcc_set, 1},
    {"v_cmp_gt_u32 + v_cndmask_b32 (2 VALU)", k_cmp_cndmask, 2},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wpc : {12, 32}) {     // 12 waves per CU = 3 per SIMD: the accumulate kernel's occupancy
    int blocks = prop.multiProcessorCount * wpc / 4;
    printf("--- %d waves/CU (%d blocks of 256)\n", wpc, blocks);
    for (auto& e : es) {
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double wave_insts = (double)ITERS * UNROLL * 4 * blocks * 4;
      double cyc = (ms * 1e-3 * 2.4e9) * prop.multiProcessorCount * 4.0 / wave_insts;
      printf("%-40s %8.3f ms  %6.2f cyc per body per SIMD @2.4GHz (%d VALU inst)\n", e.name, ms, cyc, e.insts);
    }
  }
  return 0;
}
