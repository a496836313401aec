// Test-only device kernels that expose the field / curve primitives of csrc/fp29.h, fq2.h and ec.h one operation at a
// time, so that tests/test_gpu_primitives.py can compare them ON THE DEVICE with the oracle (VERDICT r1 next #7b: until now
// they were covered on the device only through whole MSMs).  Not part of libgosnark_hip.so.
// Build (hipcc cross-compiles without a GPU): see __graft_entry__.build().
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../go-snark-study_amd/csrc/ec.h"
#include "../../go-snark-study_amd/csrc/point_io.h"

using namespace gs;

namespace {

template <class M>
__device__ Fe<M, 2> ld(const uint32_t* p) {
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) w[i] = p[i];
  return to_mont(unpack32<M>(w));
}
template <class M, int B>
__device__ void st(uint32_t* p, const Fe<M, B>& a) {
  uint32_t w[8];
  pack32<M>(from_mont(a), w);
  for (int i = 0; i < 8; ++i) p[i] = w[i];
}

// op: 0 add 1 sub 2 neg 3 dbl 4 mul 5 sqr 6 inv 7 mul_add(a,b,a,a) 8 reduce2-after-lazy-chain 9 is_zero(a - b) 10 dots2 / sqr2
template <class M>
__global__ void k_field(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe<M, 2> x = ld<M>(a + 8 * i), y = ld<M>(b + 8 * i);
  uint32_t* o = out + 8 * i;
  switch (op) {
    case 0: st<M>(o, add(x, y)); break;
    case 1: st<M>(o, sub(x, y)); break;
    case 2: st<M>(o, neg(x)); break;
    case 3: st<M>(o, dbl(x)); break;
    case 4: st<M>(o, mul(x, y)); break;
    case 5: st<M>(o, sqr(x)); break;
    case 6: st<M>(o, inv(x)); break;
    case 7: st<M>(o, mul_add(x, y, x, x)); break;                                 // x y + x^2, one reduction
    case 8: st<M>(o, reduce2(add(dbl(dbl(x)), sub(dbl(y), x)))); break;             // 4x + 2y - x, lazily, then reduced
    case 9: { Fe<M, 2> z = fe_zero<M, 2>(); z.l[0] = equal(x, y) ? 1u : 0u; for (int k = 0; k < 8; ++k) o[k] = k == 0 ? z.l[0] : 0u; break; }
    case 10: {                                                                     // interleaved chains == plain products
      Fe<M, 2> p, q, s, t, u, v, w;
      dots2<M>(dot_of(x, y), dot_of(y, y), p, q);
      sqr2(x, y, s, t);
      dots3<M>(dot_of(x, y, x, x), dot_of(x, x), dot_of(y, x), u, v, w);
      const bool ok = equal(p, mul(x, y)) && equal(q, sqr(y)) && equal(s, sqr(x)) && equal(t, sqr(y)) && equal(u, mul_add(x, y, x, x)) &&
                      equal(v, sqr(x)) && equal(w, mul(x, y));
      for (int k = 0; k < 8; ++k) o[k] = k == 0 ? (ok ? 1u : 0u) : 0u;
      break;
    }
  }
}

__device__ Fq2e<2> ld2(const uint32_t* p) { return {ld<ModQ>(p), ld<ModQ>(p + 8)}; }
template <int B> __device__ void st2(uint32_t* p, const Fq2e<B>& a) { st<ModQ>(p, a.c0); st<ModQ>(p + 8, a.c1); }

// op: 0 add 1 sub 2 neg 3 dbl 4 mul 5 sqr 6 inv 7 mul_sub(a,b,b,a) 8 mul2 / sqr2 agree with mul / sqr
__global__ void k_fq2(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fq2e<2> x = ld2(a + 16 * i), y = ld2(b + 16 * i);
  uint32_t* o = out + 16 * i;
  switch (op) {
    case 0: st2(o, add(x, y)); break;
    case 1: st2(o, sub(x, y)); break;
    case 2: st2(o, neg(x)); break;
    case 3: st2(o, dbl(x)); break;
    case 4: st2(o, mul(x, y)); break;
    case 5: st2(o, sqr(x)); break;
    case 6: st2(o, inv(x)); break;
    case 7: st2(o, mul_sub(x, y, y, x)); break;                                   // = 0
    case 8: {
      Fq2e<2> p, q, s, t;
      mul2(x, y, y, y, p, q);
      sqr2(x, y, s, t);
      auto same = [](const Fq2e<2>& u, const Fq2e<2>& v) { return is_zero(sub(u, v)); };
      const bool ok = same(p, mul(x, y)) && same(q, mul(y, y)) && same(s, sqr(x)) && same(t, sqr(y));
      for (int k = 0; k < 16; ++k) o[k] = k == 0 ? (ok ? 1u : 0u) : 0u;
      break;
    }
  }
}

// points: Jacobian standard-form triples in, affine Jacobian [x, y, 1] / [0, 0, 0] out.
// op: 0 madd (P + Q, Q affine)  1 madd with negate (P - Q)  2 dbl (2 P)  3 add (P + Q, both XYZZ)  4 k * P (k = first 8 words of Q.x)
//     5 on_curve(P) -> [flag, 0, ..]   6 add with the second operand in memory (xyzz_add_mem, the MSM tail kernels' addition)
template <class T>
__global__ void k_curve(int op, const uint32_t* a, const uint32_t* b, uint32_t* out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int cw = PointIO<T>::kCoordWords;
  const uint32_t *pa = a + (size_t)i * 3 * cw, *pb = b + (size_t)i * 3 * cw;
  const Affine<T> P = jacobian_to_affine<T>(PointIO<T>::load_std(pa), PointIO<T>::load_std(pa + cw), PointIO<T>::load_std(pa + 2 * cw));
  const Affine<T> Q = jacobian_to_affine<T>(PointIO<T>::load_std(pb), PointIO<T>::load_std(pb + cw), PointIO<T>::load_std(pb + 2 * cw));
  Xyzz<T> acc = xyzz_from_affine(P);
  uint32_t* o = out + (size_t)i * 3 * cw;
  for (int k = 0; k < 3 * cw; ++k) o[k] = 0;
  switch (op) {
    case 0: xyzz_madd(acc, Q, false); break;
    case 1: xyzz_madd(acc, Q, true); break;
    case 2: xyzz_dbl(acc); break;
    case 3: { Xyzz<T> q = xyzz_from_affine(Q); xyzz_dbl(q); xyzz_madd(q, Q, true); xyzz_add(acc, q); break; }   // Q presented as 2Q - Q: a non-trivial XYZZ operand
    case 4: { uint32_t k[8]; for (int j = 0; j < 8; ++j) k[j] = pb[j]; scalar_canon(k); acc = xyzz_mul_words_w4(acc, k); break; }
    case 5: o[0] = on_curve(P) ? 1u : 0u; return;
    case 6: {                                    // the tails' form: second operand read from MEMORY coordinate by coordinate (both XYZZ non-trivial)
      uint32_t buf[PointIO<T>::kXyzzWords];
      Xyzz<T> q = xyzz_from_affine(Q); xyzz_dbl(q); xyzz_madd(q, Q, true);
      store_xyzz<T>(buf, q);
      if (!is_inf(P)) { xyzz_dbl(acc); xyzz_madd(acc, P, true); }
      xyzz_add_mem<T>(acc, buf);
      break;
    }
  }
  const Affine<T> r = xyzz_to_affine(acc);
  if (is_inf(r)) return;
  PointIO<T>::store_std(o, r.x);
  PointIO<T>::store_std(o + cw, r.y);
  o[2 * cw] = 1u;
}

int run(int kind, int op, const void* a, const void* b, void* out, uint32_t n, size_t words_in, size_t words_out) {
  uint32_t *da = nullptr, *db = nullptr, *dout = nullptr;
  if (hipMalloc(&da, n * words_in * 4) != hipSuccess || hipMalloc(&db, n * words_in * 4) != hipSuccess || hipMalloc(&dout, n * words_out * 4) != hipSuccess) return -1;
  hipMemcpy(da, a, n * words_in * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b, n * words_in * 4, hipMemcpyHostToDevice);
  const dim3 grid((n + 63) / 64), block(64);
  switch (kind) {
    case 0: hipLaunchKernelGGL(k_field<ModQ>, grid, block, 0, 0, op, da, db, dout, n); break;
    case 1: hipLaunchKernelGGL(k_field<ModR>, grid, block, 0, 0, op, da, db, dout, n); break;
    case 2: hipLaunchKernelGGL(k_fq2, grid, block, 0, 0, op, da, db, dout, n); break;
    case 3: hipLaunchKernelGGL(k_curve<FqTag>, grid, block, 0, 0, op, da, db, dout, n); break;
    case 4: hipLaunchKernelGGL(k_curve<Fq2Tag>, grid, block, 0, 0, op, da, db, dout, n); break;
    default: return -2;
  }
  const hipError_t e = hipDeviceSynchronize();
  hipMemcpy(out, dout, n * words_out * 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  return e == hipSuccess ? 0 : -3;
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int gs_prim_run(int kind /* 0 Fq, 1 Fr, 2 Fq2, 3 G1, 4 G2 */, int op, const void* a, const void* b, void* out, uint32_t n) {
  static const size_t win[5] = {8, 8, 16, 24, 48}, wout[5] = {8, 8, 16, 24, 48};
  if (kind < 0 || kind > 4) return -2;
  return run(kind, op, a, b, out, n, win[kind], wout[kind]);
}
