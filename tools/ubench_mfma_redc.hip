// Pricing the one dense contraction on the prover's hot path: the reduction half of a Montgomery product is a
// product of the low half T_lo of a * b by a CONSTANT matrix, and over the 64 lanes of a wave that is a GEMM
//      U[lane][j] = sum_k digit_k(T_lo[lane]) * C[k][j],        C[k] = the digits of (2^(7k) / R mod p)
// (T / R = T_hi + sum_k digit_k(T_lo) 2^(7k) / R  (mod p): the non-sequential form of Montgomery's reduction, no m_k chain).
// gfx950's v_mfma_i32_16x16x64_i8 does 32768 8-bit multiply-adds per instruction -- per SIMD ~16x the rate v_mad_u64_u32 delivers in
// 32-bit products -- so the question is not the matrix pipe, it is what it costs to GET THERE AND BACK from 29-bit limbs in VGPRs:
//   1. T_lo (9 x 29-bit limbs) -> 38 unsigned 7-bit digits (i8 operands are signed: radix 128), four per dword
//   2. lane-major -> the MFMA A-operand layout (lane l holds row l & 15, K-block l >> 4): through LDS
//   3. 4 row tiles x 3 column tiles = 12 MFMAs per wave (K = 64 covers the 38 digits, 48 columns cover 37 output digits)
//   4. C/D layout (col = lane & 15, row = 4 (lane >> 4) + reg) -> lane-major: through LDS
//   5. 37 column sums of weight 2^(7j) -> 29-bit limbs: one 64-bit multiply-add by a power of two each, + carries
//   6. + T_hi, and one more small fold of the bits above 2^254
// This file runs exactly that data flow (the GEMM is checked against the host: layouts are real; the constant matrix is pseudo-random
// 7-bit data, so the result is not reduced mod p -- this prices the route, it does not implement it) against the product it would
// replace: fp29.h's dots2 (two interleaved chains, 162 multiply-adds per product).  Both at 3 waves per SIMD like the G1 accumulation.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I go-snark-study_amd/csrc tools/ubench_mfma_redc.hip -o tools/ubench_mfma_redc
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "fp29.h"
using namespace gs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 200;
constexpr int ND = 38, NJ = 37;                    // digits of T_lo (261 bits / 7), output digits (259 bits / 7)
using v4i = __attribute__((ext_vector_type(4))) int;

// the 28-bit span q of a 9 x 29-bit limb array (bits [28 q, 28 q + 28)), then its four 7-bit fields spread into four bytes
__device__ __forceinline__ uint32_t digits4(const uint32_t (&t)[NL], int q) {
  const int bit = 28 * q, i = bit / LB, a = bit % LB;
  uint32_t x = t[i] >> a;
  if (a > 1 && i + 1 < NL) x |= t[i + 1] << (LB - a);
  x &= 0x0fffffffu;
  return (x & 0x7fu) | ((x & 0x3f80u) << 1) | ((x & 0x1fc000u) << 2) | ((x & 0xfe00000u) << 3);
}

// U[j] for this lane's element (j < 40; entries >= NJ are zero): steps 1 - 4.  One LDS buffer per wave serves both transposes (the
// four A fragments are in registers before the first sum is written): 64 x 40 dwords = 10 KiB per wave, 3 workgroups per CU fit.
constexpr int NCOL = 40;
__device__ __forceinline__ void fold_gemm(const uint32_t (&tlo)[NL], const v4i (&bfrag)[3], uint32_t* sh, uint32_t (&U)[NCOL]) {
  const int lane = threadIdx.x & 63;
  uint32_t* buf = sh + (threadIdx.x >> 6) * 64 * NCOL;
  uint4* row = reinterpret_cast<uint4*>(buf + lane * 16);
  row[0] = make_uint4(digits4(tlo, 0), digits4(tlo, 1), digits4(tlo, 2), digits4(tlo, 3));
  row[1] = make_uint4(digits4(tlo, 4), digits4(tlo, 5), digits4(tlo, 6), digits4(tlo, 7));
  row[2] = make_uint4(digits4(tlo, 8), digits4(tlo, 9) & 0x00007f7fu, 0u, 0u);       // digits 36, 37 are the last
  row[3] = make_uint4(0, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
  v4i a[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint4 a4 = *reinterpret_cast<const uint4*>(buf + (16 * g + (lane & 15)) * 16 + 4 * (lane >> 4));
    a[g] = v4i{(int)a4.x, (int)a4.y, (int)a4.z, (int)a4.w};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      v4i c = {0, 0, 0, 0};
      c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[g], bfrag[n], c, 0, 0, 0);
      const int col = 16 * n + (lane & 15);
      if (col < NCOL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(16 * g + 4 * (lane >> 4) + r) * NCOL + col] = (uint32_t)c[r];
      }
    }
  __builtin_amdgcn_wave_barrier();
  const uint4* mine = reinterpret_cast<const uint4*>(buf + lane * NCOL);
#pragma unroll
  for (int q = 0; q < NCOL / 4; ++q) { const uint4 v = mine[q]; U[4 * q] = v.x; U[4 * q + 1] = v.y; U[4 * q + 2] = v.z; U[4 * q + 3] = v.w; }
  __builtin_amdgcn_wave_barrier();
}

// steps 5 - 6: sum_j U[j] 2^(7j) + T_hi -> nearly normal limbs, then the small fold of the bits above 2^254 (9 more multiply-adds)
__device__ __forceinline__ Fe<ModQ, 2> recombine(const uint32_t (&U)[NCOL], const uint32_t (&thi)[NL]) {
  Fe<ModQ, 2> r;
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (7 * j >= LB * i && 7 * j < LB * (i + 1)) acc += (uint64_t)U[j] * (1u << (7 * j - LB * i));
    acc += thi[i];
    r.l[i] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  const uint32_t top = r.l[NL - 1] >> 22;                    // what sticks out above 2^254: times (2^254 mod p), 9 multiply-adds
  r.l[NL - 1] &= (1u << 22) - 1u;
  uint64_t c2 = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    c2 += (uint64_t)top * ModQ::one(i) + r.l[i];              // (a stand-in constant of the right size)
    r.l[i] = (uint32_t)c2 & LMASK;
    c2 >>= LB;
  }
  return r;
}

// the product columns of a * b (81 multiply-adds): low half as carried limbs, high half as carried limbs
__device__ __forceinline__ void product_columns(const Fe<ModQ, 2>& a, const Fe<ModQ, 2>& b, uint32_t (&lo)[NL], uint32_t (&hi)[NL]) {
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j >= 0 && j < NL) acc += (uint64_t)a.l[i] * b.l[j];
    }
    if (k < NL) lo[k] = (uint32_t)acc & LMASK; else hi[k - NL] = (uint32_t)acc & LMASK;
    acc >>= LB;
  }
  hi[NL - 1] = (uint32_t)acc;
}

__global__ void __launch_bounds__(256, 3) k_mfma_route(const uint32_t* __restrict__ xin, const uint32_t* __restrict__ cmat, uint32_t* __restrict__ xout,
                                                        uint32_t* __restrict__ ucheck, int iters) {
  __shared__ uint32_t sh[4 * 64 * NCOL];                                   // 40 KiB per workgroup of four waves
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  Fe<ModQ, 2> x, y;
#pragma unroll
  for (int i = 0; i < NL; ++i) { x.l[i] = xin[t * 2 * NL + i]; y.l[i] = xin[t * 2 * NL + NL + i]; }
  // B operand: lane l holds column 16 n + (l & 15), K-block l >> 4 (16 consecutive digits); constant for the whole kernel
  v4i bfrag[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const uint4 v = *reinterpret_cast<const uint4*>(cmat + ((16 * n + (lane & 15)) * 16 + 4 * (lane >> 4)));
    bfrag[n] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
  uint32_t U[NCOL];
  for (int it = 0; it < iters; ++it) {
    uint32_t lo[NL], hi[NL];
    product_columns(x, y, lo, hi);
    fold_gemm(lo, bfrag, sh, U);
    if (it == 0 && ucheck) {
      for (int i = 0; i < NL; ++i) ucheck[t * 64 + i] = lo[i];
      for (int j = 0; j < NCOL; ++j) ucheck[t * 64 + 16 + j] = U[j];
    }
    x = recombine(U, hi);
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) xout[t * NL + i] = x.l[i];
}

// layout probe: lane t feeds T_lo = 2^(7 p), p = t % 38 (a single digit equal to 1): its sums must be row p of the constant matrix
__global__ void __launch_bounds__(256, 3) k_probe(const uint32_t* __restrict__ cmat, uint32_t* __restrict__ ucheck) {
  __shared__ uint32_t sh[4 * 64 * NCOL];
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, p = (int)(t % ND);
  uint32_t lo[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) lo[i] = (7 * p) / LB == i ? 1u << ((7 * p) % LB) : 0u;
  v4i bfrag[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const uint4 v = *reinterpret_cast<const uint4*>(cmat + ((16 * n + (lane & 15)) * 16 + 4 * (lane >> 4)));
    bfrag[n] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
  }
  uint32_t U[NCOL];
  fold_gemm(lo, bfrag, sh, U);
  for (int j = 0; j < NCOL; ++j) ucheck[t * 64 + j] = U[j];
}

__global__ void __launch_bounds__(256, 3) k_valu_route(const uint32_t* __restrict__ xin, uint32_t* __restrict__ xout, int iters) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  Fe<ModQ, 2> x0, y0, x1, y1;
#pragma unroll
  for (int i = 0; i < NL; ++i) { x0.l[i] = xin[t * 2 * NL + i]; y0.l[i] = xin[t * 2 * NL + NL + i]; x1.l[i] = y0.l[i] ^ 5u; y1.l[i] = x0.l[i] ^ 9u; }
  for (int it = 0; it < iters; it += 2) {                    // two products per round, interleaved chains (what the kernels use)
    Fe<ModQ, 2> r0, r1;
    dots2<ModQ>(dot_of(x0, y0), dot_of(x1, y1), r0, r1);
    x0 = r0; x1 = r1;
  }
#pragma unroll
  for (int i = 0; i < NL; ++i) xout[t * NL + i] = x0.l[i] ^ x1.l[i];
}

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int blocks = prop.multiProcessorCount * 3, threads = blocks * 256;       // 12 waves per CU = 3 per SIMD
  std::vector<uint32_t> h((size_t)threads * 2 * NL), cm(48 * 16, 0);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 20); };
  for (auto& v : h) v = rnd() & LMASK;
  std::vector<int> C(64 * 48, 0);                              // C[k][j], 7-bit entries, k < 38, j < 37
  for (int k = 0; k < ND; ++k) for (int j = 0; j < NJ; ++j) C[k * 48 + j] = (int)(rnd() & 0x7f);
  for (int j = 0; j < 48; ++j) for (int k = 0; k < 64; ++k) cm[j * 16 + k / 4] |= (uint32_t)C[k * 48 + j] << (8 * (k % 4));   // column j: 64 bytes along K
  uint32_t *dx, *dc, *dout, *du;
  CK(hipMalloc(&dx, h.size() * 4)); CK(hipMalloc(&dc, cm.size() * 4)); CK(hipMalloc(&dout, (size_t)threads * NL * 4)); CK(hipMalloc(&du, (size_t)threads * 64 * 4));
  CK(hipMemcpy(dx, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cm.data(), cm.size() * 4, hipMemcpyHostToDevice));
  {   // probe: which row of C does digit p meet?
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, dc, du);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> hp(256 * 64);
    CK(hipMemcpy(hp.data(), du, hp.size() * 4, hipMemcpyDeviceToHost));
    int wrong = 0;
    for (int t = 0; t < 256; ++t) {
      const int p = t % ND;
      int found = -1;
      for (int k = 0; k < 64 && found < 0; ++k) {
        bool eq = true;
        for (int j = 0; j < NJ; ++j) eq = eq && (uint32_t)C[k * 48 + j] == hp[t * 64 + j];
        if (eq) found = k;
      }
      if (found != p) { if (wrong < 12) printf("  probe: thread %d digit %d meets row %d of C\n", t, p, found); ++wrong; }
    }
    printf("probe: %d of 256 lanes paired with the wrong row\n", wrong);
  }
  // layout check of the GEMM on the first iteration
  hipLaunchKernelGGL(k_mfma_route, dim3(blocks), dim3(256), 0, 0, dx, dc, dout, du, 1);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> hu((size_t)threads * 64);
  CK(hipMemcpy(hu.data(), du, hu.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int t = 0; t < threads; t += 37) {
    auto digit = [&](int k) {                                    // bits [7k, 7k + 7) of the carried 29-bit limbs
      uint32_t d = 0;
      for (int b = 0; b < 7; ++b) {
        const int bit = 7 * k + b;
        if (bit < LB * NL) d |= ((hu[(size_t)t * 64 + bit / LB] >> (bit % LB)) & 1u) << b;
      }
      return d;
    };
    for (int j = 0; j < NJ; ++j) {
      uint32_t want = 0;
      for (int k = 0; k < ND; ++k) want += digit(k) * (uint32_t)C[k * 48 + j];
      if (want != hu[(size_t)t * 64 + 16 + j]) {
        if (bad < 6) printf("  t %d (lane %d) j %d: want %u got %u\n", t, t & 63, j, want, hu[(size_t)t * 64 + 16 + j]);
        ++bad;
      }
    }
  }
  printf("GEMM layout check (digits -> A fragments -> 12 MFMAs -> lane-major sums vs host): %s\n", bad ? "MISMATCH" : "ok");
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms_m = 0, ms_v = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_mfma_route, dim3(blocks), dim3(256), 0, 0, dx, dc, dout, (uint32_t*)nullptr, ITERS);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_m, a, b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_valu_route, dim3(blocks), dim3(256), 0, 0, dx, dout, ITERS);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_v, a, b));
  }
  const double waves_per_simd = 3.0, clk = prop.clockRate * 1e3;
  auto cyc = [&](float ms) { return ms * 1e-3 * clk / (ITERS * waves_per_simd); };
  printf("device %s, %d CUs, 3 waves per SIMD, %d products per lane\n", prop.gcnArchName, prop.multiProcessorCount, ITERS);
  printf("VALU route   (162 multiply-adds per product, two interleaved chains): %.3f ms  = %6.0f cycles per product per SIMD @%.1f GHz\n", ms_v, cyc(ms_v), clk / 1e9);
  printf("MFMA route   (81 multiply-adds + digits + LDS + 12 x v_mfma_i32_16x16x64_i8 per 64 products + LDS + 37 + 9 multiply-adds): %.3f ms  = %6.0f cycles per product per SIMD\n",
         ms_m, cyc(ms_m));
  printf("ratio MFMA / VALU = %.2f\n", ms_m / ms_v);
  return bad ? 2 : 0;
}
