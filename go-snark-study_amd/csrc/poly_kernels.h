// Fr polynomial kernels for gfx950: fused radix-2 NTT passes (batched), point-wise products, series
// helpers.  Replaces the reference's dense big.Int polynomial arithmetic
// (r1csqap/r1csqap.go:57-126: schoolbook Mul O(n^2), long Div O(n^3), Eval with Exp per term).
//
// Element format in HBM: 8 x u32 words, value < 2^256 (any representative of the residue; the
// kernels keep values < 2r).  The NTT is linear, so data may be in standard OR Montgomery form:
// twiddles are always Montgomery, and mont_mul(x, w R) = x w preserves the form of x.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fp29.h"

namespace gs {

using Fr6 = Fe<ModR, 6>;     // anything loaded from memory (2^256 < 6 r)
using Fr2 = Fe<ModR, 2>;

GS_HD Fr6 load_fr(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 lo = q[0], hi = q[1];
  uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return unpack32<ModR>(w);
}
// store a value < 2r (fully carried, not necessarily canonical)
GS_HD void store_fr(uint32_t* p, const Fr2& a) {
  Fr2 x = a;
  carry_full(x);
  Fe<ModR, 1> y;
#pragma unroll
  for (int i = 0; i < NL; ++i) y.l[i] = x.l[i];
  uint32_t w[8];
  pack32<ModR>(y, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
GS_HD void store_fr_canon(uint32_t* p, const Fe<ModR, 1>& a) {
  uint32_t w[8];
  pack32<ModR>(a, w);
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(w[0], w[1], w[2], w[3]);
  q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

struct FrConst { uint32_t l[NL]; };      // kernel-argument form of an Fr element (Montgomery limbs)
GS_HD Fr2 from_const(const FrConst& c) {
  Fr2 r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c.l[i];
  return r;
}

// tw[i] = omega^i (Montgomery), i < count, omega given as a constant.  Stored pre-split: 9 canonical 29-bit limbs padded to
// 12 words (48 B), so a butterfly loads its twiddle with three 16-byte loads and no unpacking.
constexpr int kTwWords = 12;
GS_HD Fe<ModR, 1> load_twiddle(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1], c = q[2];
  Fe<ModR, 1> r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = c.x;
  return r;
}
__global__ void __launch_bounds__(256) k_twiddle_gen(uint32_t* __restrict__ tw, uint32_t count, FrConst omega) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fr2 base = from_const(omega), acc = relax<2>(fe_one<ModR>());
  for (uint32_t e = i; e != 0; e >>= 1) {
    if (e & 1u) acc = mul(acc, base);
    base = sqr(base);
  }
  const Fe<ModR, 1> cv = canon(acc);
  uint4* q = reinterpret_cast<uint4*>(tw + (size_t)i * kTwWords);
  q[0] = make_uint4(cv.l[0], cv.l[1], cv.l[2], cv.l[3]);
  q[1] = make_uint4(cv.l[4], cv.l[5], cv.l[6], cv.l[7]);
  q[2] = make_uint4(cv.l[8], 0, 0, 0);
}

// ---- fused NTT pass: k consecutive radix-2 stages on a 1024-element tile held in LDS ---------------------------------
// One workgroup loads a tile of R = 2^k rows x C = 2^clog columns (rows are the indices that differ in bits
// [s_lo, s_lo + k), columns are consecutive low indices -- or, for the s_lo = 0 pass, C adjacent contiguous groups),
// converts once to 29-bit limbs, runs the k butterfly stages out of LDS (limb-major layout: conflict-free for the unit-
// stride column index) and writes the tile back: a 2^21-point transform moves 3 x 128 MiB instead of 21 x 128 MiB.
constexpr int kNttTileLog = 10;
constexpr int kNttTile = 1 << kNttTileLog;
constexpr int kNttMaxStages = 7;

constexpr int kNttEndBound = 34;     // >= the bound any stage sequence of <= 7 stages can leave (DIF: 32, or 2 * 16 + 1 after a twiddle-free last stage)
// (Radix-4 rounds -- two stages per LDS round trip -- were measured in round 2: 17.5 vs 16.5 ms for the 2^20 px stage: more registers
// per thread for the same VALU work; multiplication by the fourth root of unity is a full product in a prime field.)
// A thread owns TWO butterflies of a stage (a 1024-element tile has 512, the workgroup 256 threads): their twiddle products are two
// independent Montgomery dot products whose column chains run interleaved (fp29.h, dots2) -- no chain follows itself, so the compiler
// neither re-associates the sums (17 v_lshl_add_u64 per product) nor pads them with wait states.  Differences that only feed the
// twiddle product skip their carry pass (fp29.h, Lz).
struct Bfly { uint32_t a[NL], b[NL]; };          // in: the two inputs; out: the two outputs (same slots)

// DIF: (a, b) -> (a + b, (a - b) w).  B = value bound of the inputs; the sum is reduced below 2r when `reduce_sum`.
template <int B, int N>
GS_HD void ntt_dif_step(Bfly (&x)[N], const Fe<ModR, 1> (&w)[N], bool reduce_sum, bool trivial) {
  static_assert(N == 1 || N == 2, "one or two butterflies per thread");
  Fe<ModR, B> a[N], b[N];
#pragma unroll
  for (int t = 0; t < N; ++t)
#pragma unroll
    for (int l = 0; l < NL; ++l) { a[t].l[l] = x[t].a[l]; b[t].l[l] = x[t].b[l]; }
  if (trivial) {                                   // stage 0 of a transform: every twiddle is 1
#pragma unroll
    for (int t = 0; t < N; ++t) {
      const auto d = sub(a[t], b[t]);              // value bound 2 B + 1 <= kNttEndBound: only ever the last stage of a pass
#pragma unroll
      for (int l = 0; l < NL; ++l) x[t].b[l] = d.l[l];
    }
  } else {
    Fr2 d[N];
    if constexpr (N == 2) dots2<ModR>(dot_of(sub_lazy(a[0], b[0]), w[0]), dot_of(sub_lazy(a[1], b[1]), w[1]), d[0], d[1]);
    else d[0] = mul_lazy(sub_lazy(a[0], b[0]), w[0]);
#pragma unroll
    for (int t = 0; t < N; ++t)
#pragma unroll
      for (int l = 0; l < NL; ++l) x[t].b[l] = d[t].l[l];
  }
#pragma unroll
  for (int t = 0; t < N; ++t) {
    if (reduce_sum) {
      const Fr2 s = reduce2(add(a[t], b[t]));
#pragma unroll
      for (int l = 0; l < NL; ++l) x[t].a[l] = s.l[l];
    } else {
      const Fe<ModR, 2 * B> s = add(a[t], b[t]);
#pragma unroll
      for (int l = 0; l < NL; ++l) x[t].a[l] = s.l[l];
    }
  }
}
// step 0,1,2,3 of every group of four: input bounds 2, 4, 8, 16; the fourth one reduces its sum back to 2.  ONE code body for all
// four (instantiated at the largest bound: the bound only selects the bias constant and feeds the static checks) -- four copies of a
// ~1300-instruction butterfly per kernel, with co-resident workgroups in different stages, do not fit the instruction cache well.
template <int N>
GS_HD void ntt_dif_butterfly(int step, Bfly (&x)[N], const Fe<ModR, 1> (&w)[N], bool trivial) {
  ntt_dif_step<16, N>(x, w, (step & 3) == 3, trivial);
}
// DIT: (a, b) -> (a + b w, a - b w).  The subtrahend is always the fresh product t = b w (nearly normal), so an EVEN step may leave
// both outputs un-carried (limbs < 3 * 2^29: fine as the next step's addend and, weight 3, as its twiddle-product operand) and the
// ODD step after it carries: one carry pass per element every second stage.  Value bound after each stage: 2 -> 5 -> 8 -> ... -> 23.
template <int B, int N>
GS_HD void ntt_dit_step(Bfly (&x)[N], const Fe<ModR, 1> (&w)[N], bool lazy_in, bool lazy_out, bool trivial) {
  static_assert(N == 1 || N == 2, "one or two butterflies per thread");
  Fr2 t[N];
  if (trivial) {                                   // stage 0: twiddle 1, and the first stage of a pass: inputs nearly normal, bound 2
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int l = 0; l < NL; ++l) t[k].l[l] = x[k].b[l];
  } else if (lazy_in) {
    Lz<ModR, B, 3> b[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int l = 0; l < NL; ++l) b[k].l[l] = x[k].b[l];
    if constexpr (N == 2) dots2<ModR>(dot_of(b[0], w[0]), dot_of(b[1], w[1]), t[0], t[1]);
    else t[0] = mul_lazy(b[0], w[0]);
  } else {
    Fe<ModR, B> b[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
      for (int l = 0; l < NL; ++l) b[k].l[l] = x[k].b[l];
    if constexpr (N == 2) dots2<ModR>(dot_of(b[0], w[0]), dot_of(b[1], w[1]), t[0], t[1]);
    else t[0] = mul_lazy(b[0], w[0]);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    // raw limb arithmetic: a (limbs < 3 * 2^29 + 16) + t resp. a + (bias - t) stay below 2^32; t < 2r, so the bias is 3r (tbias(3))
    uint32_t p[NL], m[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) { p[l] = x[k].a[l] + t[k].l[l]; m[l] = x[k].a[l] + (ModR::tbias(3, l) - t[k].l[l]); }
    static_assert(ModR::tbias_k(3) == 3, "the DIT bound table (B + 2, B + 3) assumes the 3r bias");
    if (!lazy_out) {
      Fe<ModR, B + 3> pp, mm;
#pragma unroll
      for (int l = 0; l < NL; ++l) { pp.l[l] = p[l]; mm.l[l] = m[l]; }
      carry_save(pp); carry_save(mm);
#pragma unroll
      for (int l = 0; l < NL; ++l) { p[l] = pp.l[l]; m[l] = mm.l[l]; }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) { x[k].a[l] = p[l]; x[k].b[l] = m[l]; }
  }
}
template <int N>
GS_HD void ntt_dit_butterfly(int step, int nsteps, Bfly (&x)[N], const Fe<ModR, 1> (&w)[N], bool trivial) {
  // even steps leave their outputs un-carried when an odd step follows inside this pass; odd steps take lazy inputs and carry.
  // Value bound after each stage: 2 -> 5 -> 8 -> 11 -> 14 -> 17 -> 20 -> 23; one code body, instantiated at the largest (see the DIF).
  const bool lazy_out = (step & 1) == 0 && step + 1 < nsteps;
  ntt_dit_step<20, N>(x, w, true, lazy_out, trivial);
}

// What a pass loads (round 5: two point-wise kernels of the H-values stage fused into the pass that follows them):
//   kNttLoadPlain   x itself
//   kNttLoadScaled  x[i] * aux0[i & mask]: the spectrum product of a convolution inside the FIRST inverse pass (was k_pw_mul_bcast:
//                   a read and a write of the whole batch)
//   kNttLoadWeighed the zero-padded, weighed input of the node-extension convolution inside the FIRST forward pass: element
//                   (v, j) of vector v is aux0[v * n + j] * aux1[j] for j < n and 0 above (was k_hx_weigh: a write and a re-read of
//                   the whole batch, half of it zeros)
constexpr int kNttLoadPlain = 0, kNttLoadScaled = 1, kNttLoadWeighed = 2;
struct NttLoadAux {
  const uint32_t* aux0;
  const uint32_t* aux1;
  uint32_t n;          // kNttLoadWeighed: values per vector
  uint32_t logN;       // log2 of the transform size: element i of the batch is (i >> logN, i & (N - 1))
};

template <bool kInverse, int kLoad = kNttLoadPlain>
__global__ void __launch_bounds__(256) k_ntt_pass(uint32_t* __restrict__ x, const uint32_t* __restrict__ tw, int tw_logn, int s_lo, int k, int clog, NttLoadAux aux) {
  wave_priority<GS_PRIO_POLY>();
  __shared__ uint32_t sh[NL * kNttTile];
  const uint32_t R = 1u << k, C = 1u << clog, E = R << clog;
  const uint32_t tile = blockIdx.x;
  size_t base;                       // global index of (q = 0, c = 0)
  uint32_t qstride, cstride;         // index = base + q * qstride + c * cstride
  if (s_lo == 0) { base = (size_t)tile * E; qstride = 1; cstride = R; }
  else {
    const uint32_t chunks = (1u << s_lo) >> clog;
    const uint32_t hi = tile / chunks, lc = tile - hi * chunks;
    base = ((size_t)hi << (s_lo + k)) + ((size_t)lc << clog); qstride = 1u << s_lo; cstride = 1;
  }
  // load: memory-contiguous index fastest across threads
  for (uint32_t e = threadIdx.x; e < E; e += 256) {
    uint32_t q, cc;
    if (s_lo == 0) { q = e & (R - 1); cc = e >> k; } else { cc = e & (C - 1); q = e >> clog; }
    const size_t gi = base + (size_t)q * qstride + (size_t)cc * cstride;
    Fr2 v;
    if constexpr (kLoad == kNttLoadWeighed) {
      const size_t vec = gi >> aux.logN;
      const uint32_t j = (uint32_t)(gi & (((size_t)1 << aux.logN) - 1));
      if (j < aux.n) v = mul(load_fr(aux.aux0 + (vec * aux.n + j) * 8), load_fr(aux.aux1 + (size_t)j * 8));
      else v = fe_zero<ModR, 2>();
    } else if constexpr (kLoad == kNttLoadScaled) {
      v = mul(reduce2(load_fr(x + gi * 8)), load_fr(aux.aux0 + (gi & (((size_t)1 << aux.logN) - 1)) * 8));
    } else {
      v = reduce2(load_fr(x + gi * 8));
    }
    const uint32_t le = (q << clog) + cc;
#pragma unroll
    for (int l = 0; l < NL; ++l) sh[l * kNttTile + le] = v.l[l];
  }
  __syncthreads();
  // Lazy reduction: LDS holds raw limbs whose VALUE bound is tracked per stage at compile time.  DIT: (a, b) -> (a + bw, a - bw)
  // with bw < 2r grows the bound by 3 per stage (2, 5, ..., 23 after 7 stages): no reduction inside a pass.  DIF: a + b doubles
  // it (2, 4, 8, 16, 32), (a - b) w resets to 2: one reduction every fourth stage.  The final store reduces below 2r.
  // twiddle of the butterfly whose lower element sits at tile row q0 (stage bit b of this pass)
  auto twiddle_at = [&](uint32_t q0, uint32_t cc, int b) {
    const size_t gi = base + (size_t)q0 * qstride + (size_t)cc * cstride;
    const int s = s_lo + b;
    const uint32_t j = (uint32_t)gi & ((1u << s) - 1u);
    return load_twiddle(tw + ((size_t)j << (tw_logn - 1 - s)) * kTwWords);
  };
  int step = 0;
  for (; step < k; ++step) {
    const int b = kInverse ? step : k - 1 - step;          // DIF runs the stages downwards, DIT upwards
    const bool trivial = (s_lo + b) == 0;                  // stage 0 of the transform: every twiddle is 1
    auto slots = [&](uint32_t u, uint32_t& e0, uint32_t& e1, uint32_t& q0, uint32_t& cc) {
      cc = u & (C - 1);
      const uint32_t qq = u >> clog;
      q0 = ((qq >> b) << (b + 1)) | (qq & ((1u << b) - 1u));
      e0 = (q0 << clog) + cc; e1 = e0 + ((1u << b) << clog);
    };
    uint32_t u = threadIdx.x;
    for (; u + 256 < E / 2; u += 512) {                    // two butterflies of this thread at once (the full-tile case: one round)
      uint32_t e0[2], e1[2], q0[2], cc[2];
      slots(u, e0[0], e1[0], q0[0], cc[0]);
      slots(u + 256, e0[1], e1[1], q0[1], cc[1]);
      Fe<ModR, 1> w[2];
      if (!trivial) { w[0] = twiddle_at(q0[0], cc[0], b); w[1] = twiddle_at(q0[1], cc[1], b); }
      Bfly x[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int l = 0; l < NL; ++l) { x[t].a[l] = sh[l * kNttTile + e0[t]]; x[t].b[l] = sh[l * kNttTile + e1[t]]; }
      if constexpr (kInverse) ntt_dit_butterfly<2>(step, k, x, w, trivial);
      else ntt_dif_butterfly<2>(step, x, w, trivial);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int l = 0; l < NL; ++l) { sh[l * kNttTile + e0[t]] = x[t].a[l]; sh[l * kNttTile + e1[t]] = x[t].b[l]; }
    }
    for (; u < E / 2; u += 256) {                          // small tiles: one at a time
      uint32_t e0, e1, q0, cc;
      slots(u, e0, e1, q0, cc);
      Fe<ModR, 1> w[1];
      if (!trivial) w[0] = twiddle_at(q0, cc, b);
      Bfly x[1];
#pragma unroll
      for (int l = 0; l < NL; ++l) { x[0].a[l] = sh[l * kNttTile + e0]; x[0].b[l] = sh[l * kNttTile + e1]; }
      if constexpr (kInverse) ntt_dit_butterfly<1>(step, k, x, w, trivial);
      else ntt_dif_butterfly<1>(step, x, w, trivial);
#pragma unroll
      for (int l = 0; l < NL; ++l) { sh[l * kNttTile + e0] = x[0].a[l]; sh[l * kNttTile + e1] = x[0].b[l]; }
    }
    __syncthreads();
  }
  for (uint32_t e = threadIdx.x; e < E; e += 256) {
    uint32_t q, cc;
    if (s_lo == 0) { q = e & (R - 1); cc = e >> k; } else { cc = e & (C - 1); q = e >> clog; }
    const uint32_t le = (q << clog) + cc;
    Fe<ModR, kNttEndBound> v;                                 // >= the bound any stage sequence of <= 7 stages can leave
#pragma unroll
    for (int l = 0; l < NL; ++l) v.l[l] = sh[l * kNttTile + le];
    store_fr(x + (base + (size_t)q * qstride + (size_t)cc * cstride) * 8, reduce2(v));
  }
}

// out[i] = a[i] * b[i] (Montgomery product: a b / R)
__global__ void __launch_bounds__(256) k_pw_mul(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 uint32_t* __restrict__ out, uint32_t n) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_fr(out + (size_t)i * 8, mul(load_fr(a + (size_t)i * 8), load_fr(b + (size_t)i * 8)));
}
// out[i] = a[i] * k
__global__ void __launch_bounds__(256) k_pw_mul_const(const uint32_t* __restrict__ a, FrConst k, uint32_t* __restrict__ out, uint32_t n) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  store_fr(out + (size_t)i * 8, mul(load_fr(a + (size_t)i * 8), from_const(k)));
}
// out[i] = a[i] +- b[i]; missing tails are zero (lengths na, nb; out has max(na, nb))
__global__ void __launch_bounds__(256) k_addsub(const uint32_t* __restrict__ a, uint32_t na, const uint32_t* __restrict__ b, uint32_t nb,
                                                 int subtract, uint32_t* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr6 x = (i < na) ? load_fr(a + (size_t)i * 8) : fe_zero<ModR, 6>();
  const Fr6 y = (i < nb) ? load_fr(b + (size_t)i * 8) : fe_zero<ModR, 6>();
  if (subtract) store_fr(out + (size_t)i * 8, reduce2(sub(x, y)));
  else store_fr(out + (size_t)i * 8, reduce2(add(x, y)));
}
// dst[i] = src[first + count - 1 - i] for i < count, 0 for count <= i < total
__global__ void __launch_bounds__(256) k_copy_reversed(const uint32_t* __restrict__ src, uint32_t first, uint32_t count,
                                                        uint32_t* __restrict__ dst, uint32_t total) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  uint4* d = reinterpret_cast<uint4*>(dst + (size_t)i * 8);
  if (i < count) {
    const uint4* s = reinterpret_cast<const uint4*>(src + (size_t)(first + count - 1 - i) * 8);
    d[0] = s[0]; d[1] = s[1];
  } else {
    d[0] = make_uint4(0, 0, 0, 0); d[1] = d[0];
  }
}
// in place: x[i] -> canonical representative in [0, r); optional Montgomery conversions
// mode 0: canon only, 1: to Montgomery (x R), 2: from Montgomery (x / R)
__global__ void __launch_bounds__(256) k_convert(uint32_t* __restrict__ x, uint32_t n, int mode) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fr6 v = load_fr(x + (size_t)i * 8);
  if (mode == 1) store_fr_canon(x + (size_t)i * 8, canon(to_mont(v)));
  else if (mode == 2) store_fr_canon(x + (size_t)i * 8, from_mont(v));
  else store_fr_canon(x + (size_t)i * 8, canon(v));
}
// Newton step helper: u = 2 - e  (coefficient-wise: u_0 = two - e_0, u_i = -e_i), i < n
__global__ void __launch_bounds__(256) k_two_minus(const uint32_t* __restrict__ e, FrConst two, uint32_t* __restrict__ out, uint32_t n, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (i >= n) { store_fr(out + (size_t)i * 8, fe_zero<ModR, 2>()); return; }
  const Fr6 v = load_fr(e + (size_t)i * 8);
  if (i == 0) store_fr(out, reduce2(sub(from_const(two), v)));
  else store_fr(out + (size_t)i * 8, reduce2(neg(v)));
}

// ---- subproduct tree of monic polynomials with the leading 1 implicit ------------------------------------
// (Z(x) = prod (x - i), r1csqap.go:177-186 / groth16.go:122-131, built by pairwise NTT products instead of
// the reference's m-2 schoolbook multiplications by a linear factor.)
// leaves: out[i] = -(i + 1) for i < deg, 0 (factor x) for deg <= i < total; Montgomery form.
__global__ void __launch_bounds__(256) k_zp_leaves(uint32_t* __restrict__ out, uint32_t deg, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (i >= deg) { store_fr(out + (size_t)i * 8, fe_zero<ModR, 2>()); return; }
  uint32_t w[8] = {i + 1u, 0, 0, 0, 0, 0, 0, 0};
  store_fr_canon(out + (size_t)i * 8, canon(neg(to_mont(unpack32<ModR>(w)))));
}
// dst[blk * 2d + t] = t < d ? src[blk * d + t] : 0          (blocks of d -> zero-padded blocks of 2d)
__global__ void __launch_bounds__(256) k_expand_blocks(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t d, uint32_t total2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total2) return;
  const uint32_t blk = i / (2 * d), t = i - blk * 2 * d;
  uint4* q = reinterpret_cast<uint4*>(dst + (size_t)i * 8);
  if (t < d) {
    const uint4* s = reinterpret_cast<const uint4*>(src + ((size_t)blk * d + t) * 8);
    q[0] = s[0]; q[1] = s[1];
  } else {
    q[0] = make_uint4(0, 0, 0, 0); q[1] = q[0];
  }
}
// out[p * d2 + t] = buf[(2p) * d2 + t] * buf[(2p + 1) * d2 + t]       (spectra of adjacent blocks)
__global__ void __launch_bounds__(256) k_pw_mul_pairs(const uint32_t* __restrict__ buf, uint32_t* __restrict__ out, uint32_t d2, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t p = i / d2, t = i - p * d2;
  const uint32_t* a = buf + ((size_t)(2 * p) * d2 + t) * 8;
  store_fr(out + (size_t)i * 8, mul(load_fr(a), load_fr(a + (size_t)d2 * 8)));
}
// (x^d + a)(x^d + b) = x^2d + [a b + x^d (a + b)]:  out[p*2d + t] = prod[p*2d + t] * scale + (t >= d ? a[t-d] + b[t-d] : 0)
__global__ void __launch_bounds__(256) k_monic_combine(const uint32_t* __restrict__ prod, const uint32_t* __restrict__ src, FrConst scale,
                                                        uint32_t* __restrict__ out, uint32_t d, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t p = i / (2 * d), t = i - p * 2 * d;
  Fr2 v = mul(load_fr(prod + (size_t)i * 8), from_const(scale));
  if (t >= d) {
    const uint32_t* a = src + ((size_t)(2 * p) * d + (t - d)) * 8;
    v = reduce2(add(v, add(load_fr(a), load_fr(a + (size_t)d * 8))));
  }
  store_fr(out + (size_t)i * 8, v);
}

// The same combination for the interpolation tree, writing the result twice: compact (the next level's P_L / P_R source) and
// already zero-padded to blocks of 4d (the next level's NTT input) -- no separate copy and expansion passes.
__global__ void __launch_bounds__(256) k_interp_combine(const uint32_t* __restrict__ prod, const uint32_t* __restrict__ src, FrConst scale,
                                                         uint32_t* __restrict__ out, uint32_t* __restrict__ out_wide, uint32_t d, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t p = i / (2 * d), t = i - p * 2 * d;
  Fr2 v = mul(load_fr(prod + (size_t)i * 8), from_const(scale));
  if (t >= d) {
    const uint32_t* a = src + ((size_t)(2 * p) * d + (t - d)) * 8;
    v = reduce2(add(v, add(load_fr(a), load_fr(a + (size_t)d * 8))));
  }
  store_fr(out + (size_t)i * 8, v);
  if (out_wide) {
    uint32_t* w = out_wide + ((size_t)p * 4 * d + t) * 8;
    store_fr(w, v);
    uint4* z = reinterpret_cast<uint4*>(w + (size_t)2 * d * 8);
    z[0] = make_uint4(0, 0, 0, 0); z[1] = z[0];
  }
}

// interpolation tree step: out[p*d2 + t] = PL * mR + PR * mL on the spectra of adjacent blocks (one reduction for the
// two-term dot product); the m spectra repeat every `mblocks` pairs (several value vectors share one node tree).
__global__ void __launch_bounds__(256) k_pw_cross(const uint32_t* __restrict__ pspec, const uint32_t* __restrict__ mspec, uint32_t* __restrict__ out,
                                                   uint32_t d2, uint32_t mpairs, uint32_t total) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t p = i / d2, t = i - p * d2;
  const uint32_t* pl = pspec + ((size_t)(2 * p) * d2 + t) * 8;
  const uint32_t* ml = mspec + ((size_t)(2 * (p % mpairs)) * d2 + t) * 8;
  store_fr(out + (size_t)i * 8, mul_add(load_fr(pl), load_fr(ml + (size_t)d2 * 8), load_fr(pl + (size_t)d2 * 8), load_fr(ml)));
}
// leaves of the interpolation tree: out[k*total + j] = values[k*n + j] * weights[j] for j < n, 0 for the padding nodes
// (also written zero-padded to blocks of 2 into out_wide, the first level's NTT input)
__global__ void __launch_bounds__(256) k_interp_leaves(const uint32_t* __restrict__ values, const uint32_t* __restrict__ weights, uint32_t n,
                                                        uint32_t total, uint32_t nvec, uint32_t* __restrict__ out, uint32_t* __restrict__ out_wide) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total * nvec) return;
  const uint32_t k = i / total, j = i - k * total;
  Fr2 v = fe_zero<ModR, 2>();
  if (j < n) v = mul(load_fr(values + ((size_t)k * n + j) * 8), load_fr(weights + (size_t)j * 8));
  store_fr(out + (size_t)i * 8, v);
  store_fr(out_wide + (size_t)(2 * i) * 8, v);
  store_fr(out_wide + (size_t)(2 * i + 1) * 8, fe_zero<ModR, 2>());
}
constexpr uint32_t kSpmvLongRow = 512, kSpmvLongCap = 4096;
// sparse matrix (CSR, values in standard form) times a Montgomery-form vector: out[row] = sum_k val[k] * x[col[k]] (standard)
__global__ void __launch_bounds__(256) k_spmv(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const uint32_t* __restrict__ val,
                                               const uint32_t* __restrict__ x_mont, uint32_t nrows, uint32_t ncols, uint32_t* __restrict__ out,
                                               uint32_t* __restrict__ long_rows /* [0] = count, then up to kSpmvLongCap row indices */) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const uint32_t lo = rowptr[r], hi = rowptr[r + 1];
  if (hi - lo > kSpmvLongRow) {                        // one thread must not walk 10^6 entries: hand the row to k_spmv_long
    const uint32_t slot = atomicAdd(long_rows, 1u);
    if (slot < kSpmvLongCap) { long_rows[1 + slot] = r; return; }
  }                                                    // (list full: fall through and do it here, slowly but correctly)
  Fr2 acc = fe_zero<ModR, 2>();
  for (uint32_t k = lo; k < hi; ++k) {
    const uint32_t cidx = col[k];
    if (cidx >= ncols) continue;                       // validated on the host; never trust an index on the device
    acc = reduce2(add(acc, mul(load_fr(val + (size_t)k * 8), load_fr(x_mont + (size_t)cidx * 8))));
  }
  store_fr(out + (size_t)r * 8, acc);
}
// Rows longer than kSpmvLongRow, listed by k_spmv: one workgroup per row (the "one" variable of an R1CS sits in ~n
// constraints, and the trusted setup multiplies by the transposed system, where it is a row).
__global__ void __launch_bounds__(256) k_spmv_long(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const uint32_t* __restrict__ val,
                                                    const uint32_t* __restrict__ x_mont, uint32_t ncols, uint32_t* __restrict__ out,
                                                    const uint32_t* __restrict__ long_rows) {
  wave_priority<GS_PRIO_POLY>();
  __shared__ uint32_t sh[NL * 256];
  const uint32_t count = min(long_rows[0], kSpmvLongCap);
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const uint32_t r = long_rows[1 + i];
    const uint32_t lo = rowptr[r], hi = rowptr[r + 1];
    Fr2 acc = fe_zero<ModR, 2>();
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 256) {
      const uint32_t cidx = col[k];
      if (cidx >= ncols) continue;
      acc = reduce2(add(acc, mul(load_fr(val + (size_t)k * 8), load_fr(x_mont + (size_t)cidx * 8))));
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) sh[l * 256 + threadIdx.x] = acc.l[l];
    __syncthreads();
    for (uint32_t stride = 128; stride > 0; stride >>= 1) {
      if (threadIdx.x < stride) {
        Fr2 a, b;
#pragma unroll
        for (int l = 0; l < NL; ++l) { a.l[l] = sh[l * 256 + threadIdx.x]; b.l[l] = sh[l * 256 + threadIdx.x + stride]; }
        const Fr2 t = reduce2(add(a, b));
#pragma unroll
        for (int l = 0; l < NL; ++l) sh[l * 256 + threadIdx.x] = t.l[l];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      Fr2 t;
#pragma unroll
      for (int l = 0; l < NL; ++l) t.l[l] = sh[l * 256];
      store_fr(out + (size_t)r * 8, t);
    }
    __syncthreads();
  }
}

// ---- H(x) straight from the constraint values (the fast route of gs_groth16_prove_witness) -------------------------
// CombinePolynomials + DivisorPolynomial (r1csqap.go:191-216) interpolate A, B, C (three O(n log^2 n) tree interpolations),
// multiply them (size 2n) and divide by Z.  For a SATISFYING witness P = A B - C is an exact multiple of Z, so H = P / Z has
// only n coefficients and is fixed by n values: H(n + k) = (A(n + k) B(n + k) - C(n + k)) / Z(n + k), k = 1..n.  The values of
// A at the nodes n+1 .. 2n follow from its values at 1 .. n by ONE cyclic convolution of size 2n, because the nodes are equally
// spaced:  A(n + k) = M(n + k) * sum_j a_j w_j / (n + k - j)  with the barycentric weights w_j = 1 / M'(j).  Then one
// interpolation (instead of three) gives G(y) = H(y + n) and a Taylor shift (one more convolution) gives H.
// All tables below are per (n, deg Z), built once on the device from factorials.

// prefix products of an arithmetic sequence, tile by tile: value(i) = mode 0: i + 1 (-> (i + 1)!),  mode 1: top - i
constexpr int kPpBlock = 256, kPpPerThread = 8, kPpTile = kPpBlock * kPpPerThread;
GS_HD Fr2 pp_value(int mode, uint32_t top, uint32_t i) {
  uint32_t w[8] = {mode == 0 ? i + 1u : top - i, 0, 0, 0, 0, 0, 0, 0};
  return to_mont(unpack32<ModR>(w));
}
__global__ void __launch_bounds__(kPpBlock) k_pp_tiles(int mode, uint32_t top, uint32_t n, uint32_t* __restrict__ out, uint32_t* __restrict__ tile_prod) {
  __shared__ uint32_t sh[kPpBlock * NL];
  const uint32_t base = blockIdx.x * kPpTile + threadIdx.x * kPpPerThread;
  Fr2 v[kPpPerThread];
  Fr2 run = relax<2>(fe_one<ModR>());
#pragma unroll
  for (int j = 0; j < kPpPerThread; ++j) {
    if (base + j < n) run = mul(run, pp_value(mode, top, base + j));
    v[j] = run;
  }
  for (int l = 0; l < NL; ++l) sh[threadIdx.x * NL + l] = run.l[l];
  __syncthreads();
  Fr2 incl = run;
  for (int off = 1; off < kPpBlock; off <<= 1) {                   // Hillis-Steele inclusive scan of the thread products
    Fr2 o = relax<2>(fe_one<ModR>());
    const bool has = threadIdx.x >= (uint32_t)off;
    if (has) for (int l = 0; l < NL; ++l) o.l[l] = sh[(threadIdx.x - off) * NL + l];
    __syncthreads();
    if (has) { incl = mul(incl, o); for (int l = 0; l < NL; ++l) sh[threadIdx.x * NL + l] = incl.l[l]; }
    __syncthreads();
  }
  Fr2 excl = relax<2>(fe_one<ModR>());
  if (threadIdx.x > 0) for (int l = 0; l < NL; ++l) excl.l[l] = sh[(threadIdx.x - 1) * NL + l];
#pragma unroll
  for (int j = 0; j < kPpPerThread; ++j)
    if (base + j < n) store_fr(out + (size_t)(base + j) * 8, mul(v[j], excl));
  if (threadIdx.x == kPpBlock - 1) store_fr(tile_prod + (size_t)blockIdx.x * 8, incl);
}
// exclusive prefix products of the tile products, in place (one workgroup; ntiles <= 1024 * 64)
__global__ void __launch_bounds__(1024) k_pp_tile_scan(uint32_t* __restrict__ tile_prod, uint32_t ntiles) {
  __shared__ uint32_t sh[1024 * NL];
  const uint32_t per = (ntiles + 1023u) / 1024u, b0 = threadIdx.x * per;
  Fr2 run = relax<2>(fe_one<ModR>());
  for (uint32_t j = 0; j < per; ++j) if (b0 + j < ntiles) run = mul(run, load_fr(tile_prod + (size_t)(b0 + j) * 8));
  for (int l = 0; l < NL; ++l) sh[threadIdx.x * NL + l] = run.l[l];
  __syncthreads();
  Fr2 incl = run;
  for (int off = 1; off < 1024; off <<= 1) {
    Fr2 o = relax<2>(fe_one<ModR>());
    const bool has = threadIdx.x >= (uint32_t)off;
    if (has) for (int l = 0; l < NL; ++l) o.l[l] = sh[(threadIdx.x - off) * NL + l];
    __syncthreads();
    if (has) { incl = mul(incl, o); for (int l = 0; l < NL; ++l) sh[threadIdx.x * NL + l] = incl.l[l]; }
    __syncthreads();
  }
  Fr2 excl = relax<2>(fe_one<ModR>());
  if (threadIdx.x > 0) for (int l = 0; l < NL; ++l) excl.l[l] = sh[(threadIdx.x - 1) * NL + l];
  for (uint32_t j = 0; j < per; ++j)
    if (b0 + j < ntiles) {
      const Fr2 t = mul(excl, load_fr(tile_prod + (size_t)(b0 + j) * 8));
      store_fr(tile_prod + (size_t)(b0 + j) * 8, excl);
      excl = t;
    }
}
__global__ void __launch_bounds__(256) k_pp_apply(uint32_t* __restrict__ out, const uint32_t* __restrict__ tile_prefix, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || i < (uint32_t)kPpTile) return;
  store_fr(out + (size_t)i * 8, mul(load_fr(out + (size_t)i * 8), load_fr(tile_prefix + (size_t)(i / kPpTile) * 8)));
}

// fact[i] = i!, invfact[i] = 1 / i! for i <= top (Montgomery), from fwd[i] = (i + 1)! and rev[t] = top! / (top - t - 1)!:
//   invfact[i] = inv_top * (top! / i!) = inv_top * rev[top - i - 1]   (invfact[top] = inv_top)
__global__ void __launch_bounds__(256) k_fact_tables(const uint32_t* __restrict__ fwd, const uint32_t* __restrict__ rev, FrConst inv_top, uint32_t top,
                                                      uint32_t* __restrict__ fact, uint32_t* __restrict__ invfact) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > top) return;
  if (i == 0) store_fr_canon(fact, canon(relax<2>(fe_one<ModR>())));
  else store_fr_canon(fact + (size_t)i * 8, canon(reduce2(load_fr(fwd + (size_t)(i - 1) * 8))));
  if (i == top) store_fr_canon(invfact + (size_t)i * 8, canon(from_const(inv_top)));
  else store_fr_canon(invfact + (size_t)i * 8, canon(mul(from_const(inv_top), load_fr(rev + (size_t)(top - i - 1) * 8))));
}
// barycentric weights of the nodes 1..n: w_j = (-1)^(n - j) / ((j - 1)! (n - j)!)        [r1csqap.go:130-136 without its int overflow]
__global__ void __launch_bounds__(256) k_bary_weights(const uint32_t* __restrict__ invfact, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (j > n) return;
  const Fr2 v = mul(load_fr(invfact + (size_t)(j - 1) * 8), load_fr(invfact + (size_t)(n - j) * 8));
  if ((n - j) & 1u) store_fr_canon(out + (size_t)(j - 1) * 8, canon(neg(v)));
  else store_fr_canon(out + (size_t)(j - 1) * 8, canon(v));
}
// the tables of the direct H route, all Montgomery:
//   inv_seq[t] = 1 / (t + 1), t < 2n - 1 (padded with zeros to N)          the convolution kernel of the node extension
//   t1[k-1] = R * M(n+k)^2 / (N^2 Z(n+k)),  t2[k-1] = M(n+k) / (N Z(n+k)) = invfact[k-1] fact[n+k-1-dz] / N      (k = 1..n)
//   shift_q[t] = (-n)^t / t!, t < n (padded to N)                          the Taylor-shift kernel
__global__ void __launch_bounds__(256) k_hx_tables(const uint32_t* __restrict__ fact, const uint32_t* __restrict__ invfact, uint32_t n, uint32_t dz, uint32_t N,
                                                    FrConst inv_N, FrConst r2, uint32_t* __restrict__ inv_seq, uint32_t* __restrict__ t1, uint32_t* __restrict__ t2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (i < 2 * n - 1) store_fr_canon(inv_seq + (size_t)i * 8, canon(mul(load_fr(invfact + (size_t)(i + 1) * 8), load_fr(fact + (size_t)i * 8))));
  else store_fr_canon(inv_seq + (size_t)i * 8, canon(fe_zero<ModR, 2>()));
  if (i < n) {
    const uint32_t k = i + 1;
    const Fr2 b = mul(mul(load_fr(invfact + (size_t)(k - 1) * 8), load_fr(fact + (size_t)(n + k - 1 - dz) * 8)), from_const(inv_N));     // t2
    const Fr2 mk = mul(load_fr(fact + (size_t)(n + k - 1) * 8), load_fr(invfact + (size_t)(k - 1) * 8));                                   // M(n + k)
    store_fr_canon(t2 + (size_t)i * 8, canon(b));
    store_fr_canon(t1 + (size_t)i * 8, canon(mul(mul(mul(b, mk), from_const(inv_N)), from_const(r2))));
  }
}
__global__ void __launch_bounds__(256) k_hx_shift_table(const uint32_t* __restrict__ invfact, const uint32_t* __restrict__ negn_pow, uint32_t n, uint32_t N,
                                                         uint32_t* __restrict__ shift_q) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (i < n) store_fr_canon(shift_q + (size_t)i * 8, canon(mul(to_mont(load_fr(negn_pow + (size_t)i * 8)), load_fr(invfact + (size_t)i * 8))));
  else store_fr_canon(shift_q + (size_t)i * 8, canon(fe_zero<ModR, 2>()));
}
// u[v*N + j] = vals[v*n + j] * w[j] for j < n, 0 up to N (three vectors at once); standard form in, standard out
__global__ void __launch_bounds__(256) k_hx_weigh(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ weights, uint32_t n, uint32_t N, uint32_t nvec,
                                                   uint32_t* __restrict__ u) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * nvec) return;
  const uint32_t v = i / N, j = i - v * N;
  if (j < n) store_fr(u + (size_t)i * 8, mul(load_fr(vals + ((size_t)v * n + j) * 8), load_fr(weights + (size_t)j * 8)));
  else store_fr(u + (size_t)i * 8, fe_zero<ModR, 2>());
}
// spectra times the cached kernel spectrum, for nvec vectors of N
__global__ void __launch_bounds__(256) k_pw_mul_bcast(uint32_t* __restrict__ x, const uint32_t* __restrict__ spec, uint32_t N, uint32_t nvec) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * nvec) return;
  store_fr(x + (size_t)i * 8, mul(load_fr(x + (size_t)i * 8), load_fr(spec + (size_t)(i % N) * 8)));
}
// hv[k-1] = convA convB t1 - convC t2 at the nodes n + k (conv_X = x[X * N + n + k - 2], the middle of the cyclic convolutions)
__global__ void __launch_bounds__(256) k_hx_values(const uint32_t* __restrict__ conv, const uint32_t* __restrict__ t1, const uint32_t* __restrict__ t2, uint32_t n,
                                                    uint32_t N, uint32_t* __restrict__ hv) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t at = (size_t)(n - 1 + i);
  const Fr2 ab = mul(load_fr(conv + at * 8), load_fr(conv + ((size_t)N + at) * 8));
  const Fr2 x = mul(ab, load_fr(t1 + (size_t)i * 8));
  const Fr2 y = mul(load_fr(conv + ((size_t)2 * N + at) * 8), load_fr(t2 + (size_t)i * 8));
  store_fr_canon(hv + (size_t)i * 8, canon(reduce2(sub(x, y))));      // canonical: these values are MSM scalars on the evaluation-basis route
}
// a_j b_j == c_j at every root j of Z (j = 1..dz)?  bad[0] counts the violations
__global__ void __launch_bounds__(256) k_r1cs_check(const uint32_t* __restrict__ vals, uint32_t n, uint32_t dz, const FrConst r2, uint32_t* __restrict__ bad) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= dz || j >= n) return;
  const Fr2 ab = mul(mul(load_fr(vals + (size_t)j * 8), load_fr(vals + ((size_t)n + j) * 8)), from_const(r2));      // standard a b
  if (!is_zero(sub(ab, load_fr(vals + ((size_t)2 * n + j) * 8)))) atomicAdd(bad, 1u);
}
// Taylor shift, step 1: p[i'] = g[n-1-i'] (n-1-i')! for i' < n, zero padded to N
__global__ void __launch_bounds__(256) k_hx_shift_in(const uint32_t* __restrict__ g, const uint32_t* __restrict__ fact, uint32_t n, uint32_t N, uint32_t* __restrict__ p) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (i < n) store_fr(p + (size_t)i * 8, mul(load_fr(g + (size_t)(n - 1 - i) * 8), load_fr(fact + (size_t)(n - 1 - i) * 8)));
  else store_fr(p + (size_t)i * 8, fe_zero<ModR, 2>());
}
// step 2: h[j] = conv[n-1-j] / (j! N), canonical standard form
__global__ void __launch_bounds__(256) k_hx_shift_out(const uint32_t* __restrict__ conv, const uint32_t* __restrict__ invfact, FrConst inv_N, uint32_t n, uint32_t nh,
                                                       uint32_t* __restrict__ h) {
  wave_priority<GS_PRIO_POLY>();
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nh) return;
  store_fr_canon(h + (size_t)j * 8, canon(mul(mul(load_fr(conv + (size_t)(n - 1 - j) * 8), load_fr(invfact + (size_t)j * 8)), from_const(inv_N))));
}

// ---- trusted setup helpers (groth16.go:94-222 on a sparse R1CS) ---------------------------------------------------
// Lagrange basis at tau over the nodes 1..n:  L_j(tau) = M(tau) * w_j / (tau - j),  w_j = 1 / M'(j)   (Montgomery out)
__global__ void __launch_bounds__(256) k_lagrange_at(const uint32_t* __restrict__ weights, uint32_t n, FrConst tau, FrConst mtau,
                                                      uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t w[8] = {j + 1u, 0, 0, 0, 0, 0, 0, 0};
  const Fr2 d = reduce2(sub(from_const(tau), to_mont(unpack32<ModR>(w))));         // tau - (j + 1)
  const Fr2 v = mul(mul(from_const(mtau), load_fr(weights + (size_t)j * 8)), inv(d));
  store_fr_canon(out + (size_t)j * 8, canon(v));
}
// per variable i: cd[i] = (kbeta at_i + kalpha bt_i + ct_i) / kdelta for i > npublic, 0 otherwise   (BACDelta scalars,
// groth16.go:181-200) and ic[i] = (...) / kgamma for i <= npublic (Vk.IC, :202-219).  at/bt/ct standard form; the
// constants are Montgomery, so every product stays in standard form.
__global__ void __launch_bounds__(256) k_setup_scalars(const uint32_t* __restrict__ at, const uint32_t* __restrict__ bt, const uint32_t* __restrict__ ct,
                                                        uint32_t m, uint32_t npublic, FrConst kalpha, FrConst kbeta, FrConst inv_delta,
                                                        FrConst inv_gamma, FrConst one_m, uint32_t* __restrict__ cd, uint32_t* __restrict__ ic) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Fr2 t = mul_add(load_fr(at + (size_t)i * 8), from_const(kbeta), load_fr(bt + (size_t)i * 8), from_const(kalpha));
  const Fr2 u = reduce2(add(t, mul(load_fr(ct + (size_t)i * 8), from_const(one_m))));
  if (i > npublic) {
    store_fr_canon(cd + (size_t)i * 8, canon(mul(u, from_const(inv_delta))));
  } else {
    store_fr_canon(cd + (size_t)i * 8, canon(fe_zero<ModR, 2>()));
    store_fr_canon(ic + (size_t)i * 8, canon(mul(u, from_const(inv_gamma))));
  }
}
// Pinocchio key scalars (snark.go:178-216), per variable i, from at/bt/ct (standard form; constants Montgomery):
//   sa = rhoA at, sb = rhoB bt, sc = rhoC ct, sap = ka sa, sbp = kb sb, scp = kc sc, skp = kbeta (sa + sb + sc)
struct PinoConsts { FrConst rhoa, rhob, rhoc, ka, kb, kc, kbeta; };
__global__ void __launch_bounds__(256) k_pinocchio_scalars(const uint32_t* __restrict__ at, const uint32_t* __restrict__ bt, const uint32_t* __restrict__ ct,
                                                            uint32_t m, PinoConsts k, uint32_t* __restrict__ sa, uint32_t* __restrict__ sb,
                                                            uint32_t* __restrict__ sc, uint32_t* __restrict__ sap, uint32_t* __restrict__ sbp,
                                                            uint32_t* __restrict__ scp, uint32_t* __restrict__ skp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const size_t o = (size_t)i * 8;
  const Fr2 a = mul(load_fr(at + o), from_const(k.rhoa));
  const Fr2 b = mul(load_fr(bt + o), from_const(k.rhob));
  const Fr2 c = mul(load_fr(ct + o), from_const(k.rhoc));
  store_fr_canon(sa + o, canon(a));
  store_fr_canon(sb + o, canon(b));
  store_fr_canon(sc + o, canon(c));
  store_fr_canon(sap + o, canon(mul(a, from_const(k.ka))));
  store_fr_canon(sbp + o, canon(mul(b, from_const(k.kb))));
  store_fr_canon(scp + o, canon(mul(c, from_const(k.kc))));
  store_fr_canon(skp + o, canon(mul(reduce2(add(add(a, b), c)), from_const(k.kbeta))));
}

// out[i] = scale * base^i (standard form out when `scale` is standard and base Montgomery): PowersTauDelta scalars :139-149
__global__ void __launch_bounds__(256) k_scaled_powers(uint32_t* __restrict__ out, uint32_t count, FrConst base_m, FrConst scale) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fr2 b = from_const(base_m), acc = from_const(scale);
  for (uint32_t e = i; e != 0; e >>= 1) {
    if (e & 1u) acc = mul(acc, b);
    b = sqr(b);
  }
  store_fr_canon(out + (size_t)i * 8, canon(acc));
}

// ---- evaluation: sum_i v_i x^i  (r1csqap.go:118-126) ------------------------------------------------
constexpr int kEvalChunk = 64;
// partial[t] = x^(t*chunk) * sum_{i<chunk} v[t*chunk+i] x^i      (v standard form, x Montgomery -> standard)
__global__ void __launch_bounds__(256) k_eval_chunks(const uint32_t* __restrict__ v, uint32_t n, FrConst xm, uint32_t* __restrict__ partial, uint32_t nchunks) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks) return;
  const Fr2 x = from_const(xm);
  const uint32_t beg = t * kEvalChunk, end = min(n, beg + kEvalChunk);
  Fr2 acc = fe_zero<ModR, 2>();
  for (uint32_t i = end; i > beg; --i) acc = reduce2(add(mul(acc, x), load_fr(v + (size_t)(i - 1) * 8)));
  Fr2 base = x, p = relax<2>(fe_one<ModR>());
  for (uint32_t e = beg; e != 0; e >>= 1) { if (e & 1u) p = mul(p, base); base = sqr(base); }
  // acc is standard form, p is Montgomery: product is standard form
  store_fr(partial + (size_t)t * 8, mul(acc, p));
}
// single block: out[0] = sum partial[i]
__global__ void __launch_bounds__(256) k_sum_block(const uint32_t* __restrict__ partial, uint32_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t sh[256 * NL];
  Fr2 acc = fe_zero<ModR, 2>();
  for (uint32_t i = threadIdx.x; i < n; i += 256) acc = reduce2(add(acc, load_fr(partial + (size_t)i * 8)));
#pragma unroll
  for (int k = 0; k < NL; ++k) sh[threadIdx.x * NL + k] = acc.l[k];
  __syncthreads();
  for (int half = 128; half >= 1; half >>= 1) {
    if ((int)threadIdx.x < half) {
      Fr2 o;
#pragma unroll
      for (int k = 0; k < NL; ++k) o.l[k] = sh[(threadIdx.x + half) * NL + k];
      acc = reduce2(add(acc, o));
#pragma unroll
      for (int k = 0; k < NL; ++k) sh[threadIdx.x * NL + k] = acc.l[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_fr_canon(out, canon(acc));
}

}  // namespace gs
