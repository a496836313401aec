// What does ONE dependent point addition cost a wave?  (Round 6, VERDICT r5 next #4: the tails of a small MSM -- bucket combine, block
// reduce -- are chains of dependent additions on waves that sit alone on their SIMDs; "build the log-depth fold or bound it".)
// A wave runs N dependent acc += B[i] (ec.h, xyzz_add_mem -- the tail kernels' own addition -- with B in LDS, and xyzz_madd, the
// accumulation kernel's mixed addition, with its affine operand in registers) over 2^k G, k < 64, so no addition degenerates; every wave
// brackets its loop with s_memtime / s_memrealtime.  Reported: microseconds and sclk cycles per dependent addition for 1, 2 and 3 waves
// per SIMD on the whole chip and on 8 CUs (hipExtStreamCreateWithCUMask: a tail kernel of a SMALL MSM runs on an otherwise idle chip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I go-snark-study_amd/csrc tools/ubench_add_latency.hip -o tools/ubench_add_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "ec.h"
#include "point_io.h"
using namespace gs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int N = 48, K = 64;

template <class T>
__device__ Affine<T> generator() {
  Affine<T> g;
  if constexpr (PointIO<T>::kAffineWords == 16) {
#pragma unroll
    for (int i = 0; i < NL; ++i) { g.x.l[i] = Gen::g1x(i); g.y.l[i] = Gen::g1y(i); }
  } else {
#pragma unroll
    for (int i = 0; i < NL; ++i) { g.x.c0.l[i] = Gen::g2x0(i); g.x.c1.l[i] = Gen::g2x1(i); g.y.c0.l[i] = Gen::g2y0(i); g.y.c1.l[i] = Gen::g2y1(i); }
  }
  return g;
}
// chain[k] = 2^k G as raw XYZZ limbs, aff[k] = the same point packed affine
template <class T>
__global__ void k_setup(uint32_t* chain, uint32_t* aff) {
  constexpr int pw = PointIO<T>::kXyzzWords;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    Xyzz<T> acc = xyzz_from_affine(generator<T>());
    for (int k = 0; k < K; ++k) { store_xyzz<T>(chain + (size_t)k * pw, acc); xyzz_dbl(acc); }
  }
  __threadfence_block();
  __syncthreads();
  if (threadIdx.x < K) {
    const Xyzz<T> mine = load_xyzz<T>(chain + (size_t)threadIdx.x * pw);
    PointIO<T>::store_affine(aff + (size_t)threadIdx.x * PointIO<T>::kAffineWords, xyzz_to_affine(mine));
  }
}
// MIXED = false: xyzz_add_mem from LDS (tail kernels); true: xyzz_madd with the affine operand loaded from global memory (accumulation)
template <class T, bool MIXED>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* chain, const uint32_t* aff, uint32_t* out, unsigned long long* span, unsigned long long* real) {
  constexpr int pw = PointIO<T>::kXyzzWords, aw = PointIO<T>::kAffineWords;
  __shared__ uint32_t sh[K * pw];
  for (int i = threadIdx.x; i < K * pw; i += blockDim.x) sh[i] = chain[i];
  __syncthreads();
  Xyzz<T> acc = load_xyzz<T>(chain + (size_t)((threadIdx.x + blockIdx.x) % 7) * pw);       // a small multiple: the sums below never meet it again
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N; ++i) {
    const int k = 8 + (i * 5 + (int)threadIdx.x) % (K - 8);
    if constexpr (MIXED) xyzz_madd(acc, unpack_affine<T>(load_raw_affine<T>(aff + (size_t)k * aw)), (i & 1) != 0);
    else xyzz_add_mem<T>(acc, sh + (size_t)k * pw);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x % 64 == 0) { const int w = (blockIdx.x * blockDim.x + threadIdx.x) / 64; span[w] = t1 - t0; real[w] = r1 - r0; }
  store_xyzz<T>(out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * pw, acc);
}

template <class T, bool MIXED>
int run(const char* what, hipStream_t stream, int cus, int per_cu) {
  constexpr int pw = PointIO<T>::kXyzzWords, aw = PointIO<T>::kAffineWords;
  const int blocks = cus * per_cu, waves = blocks * 4;
  uint32_t *chain, *aff, *out;
  unsigned long long *span, *real;
  CK(hipMalloc(&chain, (size_t)K * pw * 4)); CK(hipMalloc(&aff, (size_t)K * aw * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * pw * 4));
  CK(hipMalloc(&span, waves * 8)); CK(hipMalloc(&real, waves * 8));
  hipLaunchKernelGGL(k_setup<T>, dim3(1), dim3(256), 0, stream, chain, aff);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a, stream));
    hipLaunchKernelGGL((k_chain<T, MIXED>), dim3(blocks), dim3(256), 0, stream, chain, aff, out, span, real);
    CK(hipEventRecord(b, stream)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
  }
  std::vector<unsigned long long> hs(waves), hr(waves);
  CK(hipMemcpy(hs.data(), span, waves * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), real, waves * 8, hipMemcpyDeviceToHost));
  double smax = 0, rmax = 0, ssum = 0, rsum = 0;
  for (int i = 0; i < waves; ++i) { smax = std::max(smax, (double)hs[i]); rmax = std::max(rmax, (double)hr[i]); ssum += hs[i]; rsum += hr[i]; }
  printf("%-34s %3d CUs x %d waves per SIMD: kernel %7.1f us | per dependent addition: longest wave %6.2f us = %6.0f cycles, mean wave %6.2f us | clock %.2f GHz\n",
         what, cus, per_cu, ms * 1e3, rmax * 0.01 / N, smax / N, rsum / waves * 0.01 / N, ssum / (rsum * 10.0));
  CK(hipFree(chain)); CK(hipFree(aff)); CK(hipFree(out)); CK(hipFree(span)); CK(hipFree(real));
  return 0;
}

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int all = prop.multiProcessorCount;
  hipStream_t few;
  std::vector<uint32_t> mask((all + 31) / 32, 0u);
  for (int i = 0; i < 8; ++i) { const int bit = i * all / 8; mask[bit / 32] |= 1u << (bit % 32); }
  CK(hipExtStreamCreateWithCUMask(&few, (uint32_t)mask.size(), mask.data()));
  for (int rep = 0; rep < 2; ++rep) {                       // (twice: the first round also ramps the clock out of idle)
    for (int per_cu : {1, 2, 3}) {
      run<FqTag, false>("G1 xyzz_add_mem (LDS operand)", few, 8, per_cu);
      run<FqTag, false>("G1 xyzz_add_mem (LDS operand)", nullptr, all, per_cu);
      run<FqTag, true>("G1 xyzz_madd (affine, global)", few, 8, per_cu);
      run<FqTag, true>("G1 xyzz_madd (affine, global)", nullptr, all, per_cu);
      if (per_cu <= 2) {
        run<Fq2Tag, false>("G2 xyzz_add_mem (LDS operand)", few, 8, per_cu);
        run<Fq2Tag, false>("G2 xyzz_add_mem (LDS operand)", nullptr, all, per_cu);
        run<Fq2Tag, true>("G2 xyzz_madd (affine, global)", few, 8, per_cu);
        run<Fq2Tag, true>("G2 xyzz_madd (affine, global)", nullptr, all, per_cu);
      }
    }
  }
  return 0;
}
