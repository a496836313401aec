// Kernels of the s_nop experiment (built and edited by tools/snop/build.py, run by tools/snop/run.hip).
// fp29.h's two-chain products (dots2: U2|S2, P^2|R^2, P^3|Q of the G1 mixed addition; the Fq2 products of the G2 one) compile to
//   v_mad A; v_mad B; s_nop 0; v_mad A; ...      (a v_mad_u64_u32 result may not be consumed two instructions later)
// while the three-chain form (dots3) needs none.  Each kernel advances independent sequences of Montgomery products; the waves-per-
// SIMD bound comes from the build (-DWAVES=2 / 3: the occupancy of the G2 / G1 accumulation kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fp29.h"
using namespace gs;
#ifndef WAVES
#define WAVES 3
#endif
constexpr int ITERS = 300;

__device__ __forceinline__ void load4(const uint32_t* in, Fe<ModQ, 2> (&x)[6], Fe<ModQ, 2>& y) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = 0; c < 6; ++c)
    for (int i = 0; i < NL; ++i) x[c].l[i] = (in[(t * 7 + c) % 4096 * NL + i]) & (i == NL - 1 ? 0x3fffffu : LMASK);
  for (int i = 0; i < NL; ++i) y.l[i] = in[(t * 7 + 6) % 4096 * NL + i] & (i == NL - 1 ? 0x3fffffu : LMASK);
}
__device__ __forceinline__ void store6(uint32_t* out, const Fe<ModQ, 2> (&x)[6]) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  for (int c = 0; c < 6; ++c)
    for (int i = 0; i < NL; ++i) out[(t * 6 + c) * NL + i] = x[c].l[i];
}
// six sequences x[c] <- x[c] * y; per iteration six products, as three two-chain groups or two three-chain groups
extern "C" __global__ void __launch_bounds__(256, WAVES) k_two_chains(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[6], y; load4(in, x, y);
  for (int it = 0; it < ITERS; ++it) {
    Fe<ModQ, 2> a, b;
    dots2<ModQ>(dot_of(x[0], y), dot_of(x[1], y), a, b); x[0] = a; x[1] = b;
    dots2<ModQ>(dot_of(x[2], y), dot_of(x[3], y), a, b); x[2] = a; x[3] = b;
    dots2<ModQ>(dot_of(x[4], y), dot_of(x[5], y), a, b); x[4] = a; x[5] = b;
  }
  store6(out, x);
}
extern "C" __global__ void __launch_bounds__(256, WAVES) k_three_chains(const uint32_t* in, uint32_t* out) {
  Fe<ModQ, 2> x[6], y; load4(in, x, y);
  for (int it = 0; it < ITERS; ++it) {
    Fe<ModQ, 2> a, b, c;
    dots3<ModQ>(dot_of(x[0], y), dot_of(x[1], y), dot_of(x[2], y), a, b, c); x[0] = a; x[1] = b; x[2] = c;
    dots3<ModQ>(dot_of(x[3], y), dot_of(x[4], y), dot_of(x[5], y), a, b, c); x[3] = a; x[4] = b; x[5] = c;
  }
  store6(out, x);
}
