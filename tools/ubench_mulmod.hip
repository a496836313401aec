// Standalone Montgomery-product throughput on gfx950 (a mulmod micro-benchmark gates every change of the
// field arithmetic).  Every thread runs a dependent sequence of products x <- x * y, (two / three independent sequences for
// the interleaved variants), 3 waves per SIMD like k_bucket_accumulate<G1>; all variants must end on the same values.
//   mul          fp29.h mul as the compiler schedules it (re-associated: a 64-bit add per column)
//   dots2/dots3  two / three independent products with interleaved column chains, accumulators pinned by empty asm
//   asm1/2/3     the same chains with the multiply-add itself written as inline asm (no pin, no compiler hazard padding)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I go-snark-study_amd/csrc tools/ubench_mulmod.hip -o tools/ubench_mulmod
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>
#include "fp29.h"
using namespace gs;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITERS = 400;
// round 5: every wave also brackets its loop with s_memtime (the shader clock), so the line reports REAL cycles per product per SIMD and
// the clock the chip ran at, not wall time x a nominal 2.4 GHz
// round 6: ... and with s_memrealtime (constant 100 MHz), because the "clock" of round 5 -- mean wave span / kernel time -- was an artefact:
// a SIMD issues oldest-first, its three waves finish after ~40 / 70 / 100 % of the kernel, so only the LONGEST span covers all the
// products, and s_memtime ticks at sclk (2.38-2.40 GHz by s_memtime / s_memrealtime, = hwmon freq1_input) in every one of these loops
__device__ unsigned long long g_span[65536], g_real[65536];
#define SPAN_BEGIN const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime(), t0_ = __builtin_amdgcn_s_memtime()
#define SPAN_END do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
                      if (threadIdx.x % 64 == 0) { g_span[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1_ - t0_; \
                                                   g_real[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = r1_ - r0_; } } while (0)

#if defined(__HIP_DEVICE_COMPILE__)
#define MAD(acc, a, b) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
#define MADS(acc, a, b) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc")
#else
#define MAD(acc, a, b) acc += (uint64_t)(a) * (b)
#define MADS(acc, a, b) acc += (uint64_t)(a) * (b)
#endif

// NC independent products r[c] = REDC(a[c] * b[c]) with the multiply-adds as inline asm, chains interleaved
template <class M, int NC>
__device__ __forceinline__ void asm_mul(const Fe<M, 2> (&a)[NC], const Fe<M, 2> (&b)[NC], Fe<M, 2> (&r)[NC]) {
  uint32_t m[NC][NL];
  uint64_t acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i <= k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) MAD(acc[c], a[c].l[i], b[c].l[k - i]);
      }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i < k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) MADS(acc[c], m[c][i], M::p(k - i));
      }
#pragma unroll
    for (int c = 0; c < NC; ++c) m[c][k] = ((uint32_t)acc[c] * M::kPinv29) & LMASK;
#pragma unroll
    for (int c = 0; c < NC; ++c) MADS(acc[c], m[c][k], M::p(0));
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) MAD(acc[c], a[c].l[i], b[c].l[k - i]);
      }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) MADS(acc[c], m[c][i], M::p(k - i));
      }
#pragma unroll
    for (int c = 0; c < NC; ++c) { r[c].l[k - NL] = (uint32_t)acc[c] & LMASK; acc[c] >>= LB; }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) r[c].l[NL - 1] = (uint32_t)acc[c];
}

__device__ __forceinline__ void load3(const uint32_t* in, Fe<ModQ, 2> (&x)[3], Fe<ModQ, 2>& y) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < NL; ++i) x[c].l[i] = (in[(t * 4 + c) % 4096 * NL + i]) & (i == NL - 1 ? 0x3fffffu : LMASK);
  for (int i = 0; i < NL; ++i) y.l[i] = in[(t * 4 + 3) % 4096 * NL + i] & (i == NL - 1 ? 0x3fffffu : LMASK);
}
__device__ __forceinline__ void store3(uint32_t* out, const Fe<ModQ, 2> (&x)[3]) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < NL; ++i) out[(t * 3 + c) * NL + i] = x[c].l[i];
}

// every kernel advances three sequences by ITERS products each (3 * ITERS products per thread)
__global__ void __launch_bounds__(256, 3) k_mul(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[3], y; load3(in, x, y);
  SPAN_BEGIN;
  for (int it = 0; it < ITERS; ++it) { x[0] = mul(x[0], y); x[1] = mul(x[1], y); x[2] = mul(x[2], y); }
  SPAN_END;
  store3(out, x);
}
__global__ void __launch_bounds__(256, 3) k_dots2(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[3], y; load3(in, x, y);
  SPAN_BEGIN;
  for (int it = 0; it < ITERS; it += 2) {            // pairs: (0,1) (2,0) (1,2): three sequences, two chains at a time
    Fe<ModQ, 2> a, b;
    dots2<ModQ>(dot_of(x[0], y), dot_of(x[1], y), a, b); x[0] = a; x[1] = b;
    dots2<ModQ>(dot_of(x[2], y), dot_of(x[0], y), a, b); x[2] = a; x[0] = b;
    dots2<ModQ>(dot_of(x[1], y), dot_of(x[2], y), a, b); x[1] = a; x[2] = b;
  }
  SPAN_END;
  store3(out, x);
}
__global__ void __launch_bounds__(256, 3) k_dots3(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[3], y; load3(in, x, y);
  SPAN_BEGIN;
  for (int it = 0; it < ITERS; ++it) {
    Fe<ModQ, 2> a, b, c;
    dots3<ModQ>(dot_of(x[0], y), dot_of(x[1], y), dot_of(x[2], y), a, b, c); x[0] = a; x[1] = b; x[2] = c;
  }
  SPAN_END;
  store3(out, x);
}
template <int NC>
__global__ void __launch_bounds__(256, 3) k_asm(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[3], y; load3(in, x, y);
  SPAN_BEGIN;
  if constexpr (NC == 1) {
    for (int it = 0; it < ITERS; ++it)
      for (int c = 0; c < 3; ++c) { Fe<ModQ, 2> a[1] = {x[c]}, b[1] = {y}, r[1]; asm_mul<ModQ, 1>(a, b, r); x[c] = r[0]; }
  } else if constexpr (NC == 2) {
    for (int it = 0; it < ITERS; it += 2) {
      const int order[3][2] = {{0, 1}, {2, 0}, {1, 2}};
      for (int g = 0; g < 3; ++g) {
        Fe<ModQ, 2> a[2] = {x[order[g][0]], x[order[g][1]]}, b[2] = {y, y}, r[2];
        asm_mul<ModQ, 2>(a, b, r);
        x[order[g][0]] = r[0]; x[order[g][1]] = r[1];
      }
    }
  } else {
    for (int it = 0; it < ITERS; ++it) { Fe<ModQ, 2> b[3] = {y, y, y}, r[3]; asm_mul<ModQ, 3>(x, b, r); x[0] = r[0]; x[1] = r[1]; x[2] = r[2]; }
  }
  SPAN_END;
  store3(out, x);
}

typedef void (*kern_t)(const uint32_t*, uint32_t*);
// argv: <variant index 0..5> <seconds>: keep launching that one variant for that long (a steady state for tools/power_trace.py)
int main(int argc, char** argv) {
  const int only = argc >= 3 ? atoi(argv[1]) : -1;
  const double seconds = argc >= 3 ? atof(argv[2]) : 0;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 3, threads = 256;          // 3 waves per SIMD
  const size_t nthreads = (size_t)blocks * threads;
  std::vector<uint32_t> h(4096 * NL);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)s; }
  uint32_t *din, *dout;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, nthreads * 3 * NL * 4));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  struct { const char* name; kern_t k; } es[] = {
    {"mul (compiler-scheduled)", k_mul}, {"dots2 (2 chains, pinned)", k_dots2}, {"dots3 (3 chains, pinned)", k_dots3},
    {"asm mad, 1 chain", k_asm<1>}, {"asm mad, 2 chains", k_asm<2>}, {"asm mad, 3 chains", k_asm<3>}};
  std::vector<uint32_t> ref, got(nthreads * 3 * NL);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("device %s, %d blocks x %d threads, %d products per thread\n", prop.name, blocks, threads, 3 * ITERS);
  int index = -1;
  for (auto& e : es) {
    if (only >= 0 && ++index != only) continue;
    hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, din, dout);
    CK(hipDeviceSynchronize());
    float ms = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    do {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, din, dout);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < seconds);
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    // compare canonical residues: representatives may differ by multiples of p between variants? no -- same algorithm, same
    // intermediate values: bit-identical limbs expected
    bool same = true;
    if (ref.empty() || only >= 0) ref = got; else same = ref == got;
    const double prods = (double)nthreads * 3 * ITERS;
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> span(waves), real(waves);
    CK(hipMemcpyFromSymbol(span.data(), HIP_SYMBOL(g_span), waves * sizeof(unsigned long long)));
    CK(hipMemcpyFromSymbol(real.data(), HIP_SYMBOL(g_real), waves * sizeof(unsigned long long)));
    double sum = 0, rsum = 0, longest = 0;
    for (auto v : span) { sum += (double)v; longest = std::max(longest, (double)v); }
    for (auto v : real) rsum += (double)v;
    const double ghz = sum / (rsum * 10.0);                     // s_memtime ticks per ns of the waves' own wall time
    printf("%-28s %8.3f ms  %7.2f G mulmod/s | %6.1f cycles per product per SIMD (longest wave span / %d products of 3 waves; kernel time x %.2f GHz: %.1f) | "
           "mean wave span %.0f %% of the longest | %s\n", e.name, ms, prods / ms / 1e6, longest / (3.0 * 3 * ITERS), 3 * 3 * ITERS, ghz,
           ms * 1e6 * ghz / (3.0 * 3 * ITERS), 100.0 * sum / waves / longest, same ? "results identical" : "RESULTS DIFFER");
  }
  return 0;
}
