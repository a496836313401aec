// BN128 prime-field arithmetic for gfx950: 9 x 29-bit limbs, Montgomery form with R = 2^261.
//
// Replaces (device side) the reference's fields/fq.go:32-98 (`Fq.Add/Sub/Neg/Mul/Square/
// Inverse` on math/big) for both moduli it is used with: q (bn128/bn128.go:85) and
// r (groth16/groth16.go:82).
//
// Why 29-bit limbs and not 8x32 / 4x64: measured on MI355X (tools/ubench_valu.hip,
// profiles/r01_ubench_valu.txt) every carry-producing/consuming VALU op (v_add_co_u32,
// v_addc_co_u32, v_lshl_add_u64) issues at HALF rate -- the same 4 cycles per wave as the
// 32x32+64 multiply-add v_mad_u64_u32 -- so a saturated-limb Montgomery product spends more
// issue slots on carries than on multiplies.  With 29-bit limbs a whole column of the
// product (9 a_i*b_j + 9 m_i*p_j terms of < 2^58) is summed inside the 64-bit accumulator
// of v_mad_u64_u32 with NO carry instructions: 162 mads + ~35 shift/mask ops per product.
// Additions are 9 independent full-rate v_add_u32 plus a carry-save "nearly normal" fix-up.
//
// Invariants ("nearly normal"): limbs 0..7 < 2^29 + 8, limb 8 (top) < 2^29; the VALUE of an
// Fe<M,B> is < B*p (B is a compile-time bound, so every formula is overflow-checked by the
// type system: mul needs Ba*Bb <= 160 and returns a value < 2p).  Values are only made
// canonical ([0,p)) at the C-ABI boundary.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "bn128_constants.h"

#define GS_HD __host__ __device__ __forceinline__
// Column accumulation order.  Each `acc += (u64)x * y` is meant to be ONE v_mad_u64_u32 whose addend is the running sum (the
// carry of the previous column included).  Left alone, the compiler re-associates the sum -- products into a fresh
// accumulator for instruction-level parallelism, then a separate 64-bit add of the carry per column (17 v_lshl_add_u64 per
// product).  A wave cannot issue dependent or independent multiply-adds faster than one per ~11 cycles anyway
// (tools/ubench_valu.hip) and three waves share a SIMD, so that parallelism buys nothing and the adds cost issue slots.
// GS_PIN makes the running sum opaque after every step so the chain stays a chain.
#ifndef GS_CHAIN
#define GS_CHAIN 0
#endif
#if GS_CHAIN && defined(__HIP_DEVICE_COMPILE__)
#define GS_PIN(acc) asm("" : "+v"(acc))
#else
#define GS_PIN(acc) ((void)0)
#endif

namespace gs {

// Issue priority of a kernel's waves (s_setprio, 0 = default .. 3): the arbiter of a SIMD prefers the higher-priority wave whenever
// both have an instruction ready.  The accumulation kernels keep every SIMD's multiplier busy; a latency-bound kernel
// that shares the SIMD (plan, NTT pass) issues little but, at equal priority, waits behind the accumulation wave for every slot --
// k_digits 0.025 ms alone, 0.46 ms beside an accumulation (profiles/r04_timeline_msm_g1_steady.txt).  Measured: profiles/r04_ab_wave_priority.txt.
#ifndef GS_PRIO_PLAN
#define GS_PRIO_PLAN 0
#endif
#ifndef GS_PRIO_POLY
#define GS_PRIO_POLY 0
#endif
#ifndef GS_PRIO_TAIL
#define GS_PRIO_TAIL 0
#endif
template <int P> __device__ __forceinline__ void wave_priority() {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (P > 0) __builtin_amdgcn_s_setprio(P);
#endif
}

constexpr int NL = 9;
constexpr int LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1u;

template <class M, int B>
struct Fe {
  static_assert(B >= 1 && B <= M::kMaxBiasK, "value bound out of supported range");
  uint32_t l[NL];
};

// ---- bound bookkeeping ---------------------------------------------------------------------
template <int BN, class M, int B>
GS_HD Fe<M, BN> relax(const Fe<M, B>& a) {      // forget precision: B -> BN >= B (free)
  static_assert(BN >= B, "relax can only loosen a bound");
  Fe<M, BN> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  return r;
}

template <class M, int B>
GS_HD Fe<M, B> fe_zero() {
  Fe<M, B> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
  return r;
}

template <class M>
GS_HD Fe<M, 1> fe_one() {                      // 1 in Montgomery form
  Fe<M, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = M::one(i);
  return r;
}

// carry-save normalisation: every limb hands its excess to the next one in parallel
// (no ripple).  Input limbs may be anything < 2^32; output is nearly normal.
template <class M, int B>
GS_HD void carry_save(Fe<M, B>& x) {
  uint32_t c[NL - 1];
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) c[i] = x.l[i] >> LB;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) x.l[i] &= LMASK;
#pragma unroll
  for (int i = 1; i < NL; ++i) x.l[i] += c[i - 1];
}

// full (rippling) normalisation: limbs 0..7 < 2^29 exactly.
template <class M, int B>
GS_HD void carry_full(Fe<M, B>& x) {
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    x.l[i + 1] += x.l[i] >> LB;
    x.l[i] &= LMASK;
  }
}

template <class M, int Ba, int Bb>
GS_HD Fe<M, Ba + Bb> add(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  Fe<M, Ba + Bb> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
  carry_save(r);
  return r;
}

template <class M, int B>
GS_HD Fe<M, 2 * B> dbl(const Fe<M, B>& a) {
  Fe<M, 2 * B> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] << 1;
  carry_save(r);
  return r;
}

// a - b + (Bb+1) p  (always >= 0 limb-wise, see tools/gen_constants.py:bias_limbs)
template <class M, int Ba, int Bb>
GS_HD Fe<M, Ba + Bb + 1> sub(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  Fe<M, Ba + Bb + 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + (M::bias(Bb + 1, i) - b.l[i]);
  carry_save(r);
  return r;
}

// The same difference with a RIPPLING carry: limb i takes the carry of limb i - 1 inside its own addition (v_add3_u32), so the
// separate carry pass of carry_save (shift, mask, add per limb: 24 instructions) shrinks to shift + mask (16) -- 34 instructions
// instead of 42, at the price of a 16-deep dependency chain that the other chains of a mixed addition cover.  Limbs 0..7 come out
// below 2^29 exactly.  Used for P and R of the accumulation kernels' mixed additions (ec.h).
template <class M, int Ba, int Bb>
GS_HD Fe<M, Ba + Bb + 1> sub_ripple(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  Fe<M, Ba + Bb + 1> r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const uint32_t t = (a.l[i] - b.l[i]) + M::bias(Bb + 1, i) + c;      // >= 0 limb-wise (bias_limbs), < 2^32
    if (i < NL - 1) { r.l[i] = t & LMASK; c = t >> LB; }
    else r.l[i] = t;
  }
  return r;
}

// (B+1) p - a
template <class M, int B>
GS_HD Fe<M, B + 1> neg(const Fe<M, B>& a) {
  Fe<M, B + 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = M::bias(B + 1, i) - a.l[i];
  carry_save(r);
  return r;
}

// ---- lazy limbs: add / sub / neg results that skip the carry pass ------------------------------------------------
// A carry pass (carry_save) is 24 instructions, a subtraction with it 42 -- and most sums and differences of a point addition
// are used exactly once, as ONE operand of a Montgomery product.  The 64-bit column accumulator has room for that: a column
// is 9 T products of nearly-normal limbs (T terms, each < 2^58 (1 + 2^-25)) + 9 reduction products (< 2^58) + the carry of the
// previous column (< 2^36), and 2^64 / (9 * 2^58) = 7.1.  Lz<M, B, W> is a field element of VALUE < B p whose limbs are only
// bounded by W * 2^29 + 16 (W = 2: an un-carried add / dbl / neg, 3: an un-carried sub): a product term with operand limb
// weights (wa, wb) counts wa * wb instead of 1, and every dot product statically checks  9 * sum_t maxa_t * maxb_t + 9 * 2^58 +
// 2^36 < 2^64  with the exact limb maxima (dot_of below).  Nothing but dot_of (and select / normalize) accepts an Lz.
template <class M, int B, int W>
struct Lz {
  static_assert(B >= 1 && B <= M::kMaxBiasK && W >= 1 && W <= 3, "lazy element out of range");
  uint32_t l[NL];
};
constexpr uint64_t kNearlyNormalMax = (1ull << LB) + 16;          // limbs 0..7 of a nearly-normal Fe stay below this
constexpr uint64_t limb_max(int w) { return (uint64_t)w * (1ull << LB) + 16; }

template <class X> struct Operand;                                 // value bound + limb weight of a product operand
template <class M, int B> struct Operand<Fe<M, B>> { using Mod = M; static constexpr int bound = B, weight = 1; };
template <class M, int B, int W> struct Operand<Lz<M, B, W>> { using Mod = M; static constexpr int bound = B, weight = W; };

template <class M, int B> GS_HD Lz<M, B, 1> as_lazy(const Fe<M, B>& a) {
  Lz<M, B, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  return r;
}
template <int WN, class M, int B, int W> GS_HD Lz<M, B, WN> widen(const Lz<M, B, W>& a) {
  static_assert(WN >= W, "widen can only loosen a limb bound");
  Lz<M, B, WN> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  return r;
}
template <class M, int B, int W> GS_HD Lz<M, B, W> select(bool c, const Lz<M, B, W>& a, const Lz<M, B, W>& b) {
  Lz<M, B, W> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
// the carry pass after all: nearly normal again
template <class M, int B, int W> GS_HD Fe<M, B> normalize(const Lz<M, B, W>& a) {
  Fe<M, B> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i];
  carry_save(r);
  return r;
}
// a + b, limbs < 2^30 + 32
template <class M, int Ba, int Bb>
GS_HD Lz<M, Ba + Bb, 2> add_lazy(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  Lz<M, Ba + Bb, 2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class M, int B>
GS_HD Lz<M, 2 * B, 2> dbl_lazy(const Fe<M, B>& a) {
  Lz<M, 2 * B, 2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] << 1;
  return r;
}
// K p - a with the TIGHT bias (tools/gen_constants.py: low limbs of the bias in [2^29 + 15, 2^30)): limbs < 2^30, K = tbias_k(B + 1)
template <class M, int B>
GS_HD Lz<M, M::tbias_k(B + 1), 2> neg_lazy(const Fe<M, B>& a) {
  Lz<M, M::tbias_k(B + 1), 2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = M::tbias(B + 1, i) - a.l[i];
  return r;
}
// a - b + K p, limbs < 3 * 2^29 + 16
template <class M, int Ba, int Bb>
GS_HD Lz<M, Ba + M::tbias_k(Bb + 1), 3> sub_lazy(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  Lz<M, Ba + M::tbias_k(Bb + 1), 3> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + (M::tbias(Bb + 1, i) - b.l[i]);
  return r;
}
// a - b - 2 c + (Bb + 2 Bc + 1) p with ONE carry pass (the X3 of every addition formula: RR - PPP - 2 Q).  Wide bias: low limbs
// >= 2^31 - 4 >= b_i + 2 c_i, and a_i + bias_i - b_i - 2 c_i < 2^32.
template <class M, int Ba, int Bb, int Bc>
GS_HD Fe<M, Ba + Bb + 2 * Bc + 1> sub_b_2c(const Fe<M, Ba>& a, const Fe<M, Bb>& b, const Fe<M, Bc>& c) {
  constexpr int K = Bb + 2 * Bc + 1;
  Fe<M, Ba + K> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + (M::wbias(K, i) - ((c.l[i] << 1) + b.l[i]));
  carry_save(r);
  return r;
}

// ---- Montgomery product (product scanning, reduction interleaved) ---------------------------
// acc is the 64-bit column accumulator: each `acc += (u64)x * y` is one v_mad_u64_u32.
template <class M>
GS_HD void mont_low_column(uint64_t& acc, uint32_t (&m)[NL], int k) {
  // add the m_i * p_{k-i} terms already known, derive m_k, clear the column, shift.
#pragma unroll
  for (int i = 0; i < NL; ++i)
    if (i < k) { acc += (uint64_t)m[i] * M::p(k - i); GS_PIN(acc); }
  m[k] = ((uint32_t)acc * M::kPinv29) & LMASK;
  acc += (uint64_t)m[k] * M::p(0); GS_PIN(acc);
  acc >>= LB;
}

template <class M>
GS_HD void mont_high_column(uint64_t& acc, const uint32_t (&m)[NL], int k, uint32_t& out) {
#pragma unroll
  for (int i = 0; i < NL; ++i)
    if (i >= k - (NL - 1)) { acc += (uint64_t)m[i] * M::p(k - i); GS_PIN(acc); }
  out = (uint32_t)acc & LMASK;
  acc >>= LB;
}

template <class M, int Ba, int Bb>
GS_HD Fe<M, 2> mul(const Fe<M, Ba>& a, const Fe<M, Bb>& b) {
  static_assert(Ba * Bb <= 160, "Montgomery product input bound exceeded (a*b must be < 169 p^2)");
  uint32_t m[NL];
  Fe<M, 2> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i <= k) { acc += (uint64_t)a.l[i] * b.l[k - i]; GS_PIN(acc); }
    mont_low_column<M>(acc, m, k);
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) { acc += (uint64_t)a.l[i] * b.l[k - i]; GS_PIN(acc); }
    mont_high_column<M>(acc, m, k, r.l[k - NL]);
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

template <class M, int Ba>
GS_HD Fe<M, 2> sqr(const Fe<M, Ba>& a) {
  static_assert(Ba * Ba <= 160, "Montgomery square input bound exceeded");
  uint32_t m[NL], a2[NL];
  Fe<M, 2> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) a2[i] = a.l[i] << 1;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j >= 0 && j < NL && i < j) { acc += (uint64_t)a2[i] * a.l[j]; GS_PIN(acc); }
    }
    if ((k & 1) == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; GS_PIN(acc); }
    if (k < NL) mont_low_column<M>(acc, m, k);
    else mont_high_column<M>(acc, m, k, r.l[k - NL]);
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// REDC(a*b + c*d): one reduction for a two-term dot product (Fq2 arithmetic).
template <class M, int Ba, int Bb, int Bc, int Bd>
GS_HD Fe<M, 2> mul_add(const Fe<M, Ba>& a, const Fe<M, Bb>& b, const Fe<M, Bc>& c, const Fe<M, Bd>& d) {
  static_assert(Ba * Bb + Bc * Bd <= 160, "dot-product input bound exceeded");
  uint32_t m[NL];
  Fe<M, 2> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j >= 0 && j < NL) {
        acc += (uint64_t)a.l[i] * b.l[j]; GS_PIN(acc);
        acc += (uint64_t)c.l[i] * d.l[j]; GS_PIN(acc);
      }
    }
    if (k < NL) mont_low_column<M>(acc, m, k);
    else mont_high_column<M>(acc, m, k, r.l[k - NL]);
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// REDC(a*b + c*d + e*f + g*h): one reduction for a four-term dot product (fused Fq2 expressions).
template <class M, int Ba, int Bb, int Bc, int Bd, int Be, int Bf, int Bg, int Bh>
GS_HD Fe<M, 2> dot4(const Fe<M, Ba>& a, const Fe<M, Bb>& b, const Fe<M, Bc>& c, const Fe<M, Bd>& d,
                    const Fe<M, Be>& e, const Fe<M, Bf>& f, const Fe<M, Bg>& g, const Fe<M, Bh>& h) {
  static_assert(Ba * Bb + Bc * Bd + Be * Bf + Bg * Bh <= 160, "dot-product input bound exceeded");
  uint32_t m[NL];
  Fe<M, 2> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = k - i;
      if (j >= 0 && j < NL) {
        acc += (uint64_t)a.l[i] * b.l[j]; GS_PIN(acc);
        acc += (uint64_t)c.l[i] * d.l[j]; GS_PIN(acc);
        acc += (uint64_t)e.l[i] * f.l[j]; GS_PIN(acc);
        acc += (uint64_t)g.l[i] * h.l[j]; GS_PIN(acc);
      }
    }
    if (k < NL) mont_low_column<M>(acc, m, k);
    else mont_high_column<M>(acc, m, k, r.l[k - NL]);
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// ---- interleaved column chains --------------------------------------------------------------------------
// A Montgomery product is ONE dependent chain of 162 v_mad_u64_u32 through its 64-bit column accumulator.  Left alone, the
// compiler breaks the chain for instruction-level parallelism -- products into a second accumulator, then a 64-bit add per
// column (17 v_lshl_add_u64 per product, ~7 % of the kernel's issue cycles: tools/ubench_valu2.hip prices the add like a
// multiply-add) -- and pinning the chain (GS_CHAIN) trades the adds for one s_nop wait state per dependent pair.  Two (or three)
// INDEPENDENT dot products computed together avoid the re-association and most of the waiting: their multiply-adds alternate in
// program order, so no instruction depends on its predecessor.  A point addition has such pairs everywhere (U2 | S2, P^2 | R^2,
// P^3 | Q, and the two coordinates of every Fq2 product).  (Round 3 claimed "no wait state is needed" here.  The ISA says otherwise --
// the compiler still pads every round of the chains with one s_nop: v_mad A; v_mad B; s_nop 0; v_mad A -- and round 4 measured it,
// tools/snop -> profiles/r04_ab_snop.txt: the padding is required (with it deleted the hardware interlocks, same results, no faster),
// and a two-chain product costs 3-4 % more than a three-chain one at 2 and at 3 waves per SIMD.)
// GS_STEP pins an accumulator after one multiply-add (device code; volatile, so the alternation survives scheduling).
#ifndef GS_PAIR
#define GS_PAIR 1
#endif
#if GS_PAIR && defined(__HIP_DEVICE_COMPILE__) && !defined(GS_NOSTEP)
#define GS_STEP(acc) asm volatile("" : "+v"(acc))
#else
#define GS_STEP(acc) ((void)0)
#endif

// one dot product of T limb-array pairs: the operands of sum_t a[t] * b[t]
template <int T>
struct Dot {
  const uint32_t* a[T];
  const uint32_t* b[T];
};
template <> struct Dot<0> {};
// Operands may be nearly-normal Fe or lazy Lz (above); both checks are static: the value bound of Montgomery's reduction
// (sum of Ba * Bb <= 160) and the 64-bit column accumulator (exact limb maxima).
constexpr bool column_fits(uint64_t sum_of_limb_products /* sum_t maxa_t * maxb_t */) {
  const unsigned __int128 col = (unsigned __int128)9 * sum_of_limb_products + (unsigned __int128)9 * (1ull << LB) * (1ull << LB) + ((unsigned __int128)1 << 36);
  return col < ((unsigned __int128)1 << 64);
}
template <class A, class Bv> constexpr uint64_t term_limbs() { return limb_max(Operand<A>::weight) * limb_max(Operand<Bv>::weight); }
template <class A, class Bv>
GS_HD Dot<1> dot_of(const A& a, const Bv& b) {
  static_assert(Operand<A>::bound * Operand<Bv>::bound <= 160, "Montgomery product input bound exceeded");
  static_assert(column_fits(term_limbs<A, Bv>()), "column accumulator would overflow: normalize an operand");
  return Dot<1>{{a.l}, {b.l}};
}
template <class A, class Bv, class Cv, class Dv>
GS_HD Dot<2> dot_of(const A& a, const Bv& b, const Cv& c, const Dv& d) {
  static_assert(Operand<A>::bound * Operand<Bv>::bound + Operand<Cv>::bound * Operand<Dv>::bound <= 160, "dot-product input bound exceeded");
  static_assert(column_fits(term_limbs<A, Bv>() + term_limbs<Cv, Dv>()), "column accumulator would overflow: normalize an operand");
  return Dot<2>{{a.l, c.l}, {b.l, d.l}};
}
template <class A, class Bv, class Cv, class Dv, class Ev, class Fv, class Gv, class Hv>
GS_HD Dot<4> dot_of(const A& a, const Bv& b, const Cv& c, const Dv& d, const Ev& e, const Fv& f, const Gv& g, const Hv& h) {
  static_assert(Operand<A>::bound * Operand<Bv>::bound + Operand<Cv>::bound * Operand<Dv>::bound + Operand<Ev>::bound * Operand<Fv>::bound +
                Operand<Gv>::bound * Operand<Hv>::bound <= 160, "dot-product input bound exceeded");
  static_assert(column_fits(term_limbs<A, Bv>() + term_limbs<Cv, Dv>() + term_limbs<Ev, Fv>() + term_limbs<Gv, Hv>()),
                "column accumulator would overflow: normalize an operand");
  return Dot<4>{{a.l, c.l, e.l, g.l}, {b.l, d.l, f.l, h.l}};
}

// r_x = REDC(x), r_y = REDC(y) [, r_z = REDC(z)]: the chains advance in lock step, one multiply-add of each in turn
// (order per column position: x_0 y_0 x_1 y_1 ... then z, so that with (TX, TY, TZ) = (2, 1, 1) no chain follows itself).
// The low and the high columns are two separate loops on purpose: with the pins (convergent inline asm) inside an
// `if (k < NL) ... else ...` of ONE loop the unroller gives up and the limbs end up in scratch memory.
#define GS_DOTS_PRODUCTS(COND)                                                                             \
  _Pragma("unroll") for (int i = 0; i < NL; ++i) {                                                         \
    if (COND) {                                                                                            \
      const int j = k - i;                                                                                 \
      if constexpr (TX > 0) { ax += (uint64_t)x.a[0][i] * x.b[0][j]; GS_STEP(ax); }                        \
      if constexpr (TY > 0) { ay += (uint64_t)y.a[0][i] * y.b[0][j]; GS_STEP(ay); }                        \
      if constexpr (TX > 1) { ax += (uint64_t)x.a[1][i] * x.b[1][j]; GS_STEP(ax); }                        \
      if constexpr (TY > 1) { ay += (uint64_t)y.a[1][i] * y.b[1][j]; GS_STEP(ay); }                        \
      if constexpr (TX > 2) { ax += (uint64_t)x.a[2][i] * x.b[2][j]; GS_STEP(ax); }                        \
      if constexpr (TY > 2) { ay += (uint64_t)y.a[2][i] * y.b[2][j]; GS_STEP(ay); }                        \
      if constexpr (TX > 3) { ax += (uint64_t)x.a[3][i] * x.b[3][j]; GS_STEP(ax); }                        \
      if constexpr (TY > 3) { ay += (uint64_t)y.a[3][i] * y.b[3][j]; GS_STEP(ay); }                        \
      if constexpr (TZ > 0) { az += (uint64_t)z->a[0][i] * z->b[0][j]; GS_STEP(az); }                      \
    }                                                                                                      \
  }
#define GS_DOTS_REDUCE(COND)                                                                               \
  _Pragma("unroll") for (int i = 0; i < NL; ++i)                                                           \
    if (COND) {                                                                                            \
      ax += (uint64_t)mx[i] * M::p(k - i); GS_STEP(ax);                                                    \
      ay += (uint64_t)my[i] * M::p(k - i); GS_STEP(ay);                                                    \
      if constexpr (TZ > 0) { az += (uint64_t)mz[i] * M::p(k - i); GS_STEP(az); }                          \
    }

template <class M, int TX, int TY, int TZ>
GS_HD void dots_interleaved(const Dot<TX>& x, const Dot<TY>& y, const Dot<TZ>* z, Fe<M, 2>& rx, Fe<M, 2>& ry, Fe<M, 2>* rz) {
  static_assert(TZ <= 1 && TX >= 1 && TY >= 1 && TX <= 4 && TY <= 4, "third chain carries one product; at most four terms per chain");
  uint32_t mx[NL], my[NL], mz[NL];
  uint64_t ax = 0, ay = 0, az = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    GS_DOTS_PRODUCTS(i <= k)
    GS_DOTS_REDUCE(i < k)
    mx[k] = ((uint32_t)ax * M::kPinv29) & LMASK;
    my[k] = ((uint32_t)ay * M::kPinv29) & LMASK;
    if constexpr (TZ > 0) mz[k] = ((uint32_t)az * M::kPinv29) & LMASK;
    ax += (uint64_t)mx[k] * M::p(0); GS_STEP(ax);
    ay += (uint64_t)my[k] * M::p(0); GS_STEP(ay);
    if constexpr (TZ > 0) { az += (uint64_t)mz[k] * M::p(0); GS_STEP(az); }
    ax >>= LB; ay >>= LB;
    if constexpr (TZ > 0) az >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    GS_DOTS_PRODUCTS(i >= k - (NL - 1))
    GS_DOTS_REDUCE(i >= k - (NL - 1))
    rx.l[k - NL] = (uint32_t)ax & LMASK;
    ry.l[k - NL] = (uint32_t)ay & LMASK;
    if constexpr (TZ > 0) rz->l[k - NL] = (uint32_t)az & LMASK;
    ax >>= LB; ay >>= LB;
    if constexpr (TZ > 0) az >>= LB;
  }
  rx.l[NL - 1] = (uint32_t)ax;
  ry.l[NL - 1] = (uint32_t)ay;
  if constexpr (TZ > 0) rz->l[NL - 1] = (uint32_t)az;
  (void)mz; (void)az;
}
#undef GS_DOTS_PRODUCTS
#undef GS_DOTS_REDUCE
template <class M, int TX, int TY>
GS_HD void dots2(const Dot<TX>& x, const Dot<TY>& y, Fe<M, 2>& rx, Fe<M, 2>& ry) {
  dots_interleaved<M, TX, TY, 0>(x, y, static_cast<const Dot<0>*>(nullptr), rx, ry, static_cast<Fe<M, 2>*>(nullptr));
}
template <class M, int TX, int TY, int TZ>
GS_HD void dots3(const Dot<TX>& x, const Dot<TY>& y, const Dot<TZ>& z, Fe<M, 2>& rx, Fe<M, 2>& ry, Fe<M, 2>& rz) {
  dots_interleaved<M, TX, TY, TZ>(x, y, &z, rx, ry, &rz);
}

// one product with lazy operands (single chain)
template <class A, class Bv>
GS_HD Fe<typename Operand<A>::Mod, 2> mul_lazy(const A& a, const Bv& b) {
  using M = typename Operand<A>::Mod;
  static_assert(Operand<A>::bound * Operand<Bv>::bound <= 160, "Montgomery product input bound exceeded");
  static_assert(column_fits(term_limbs<A, Bv>()), "column accumulator would overflow: normalize an operand");
  uint32_t m[NL];
  Fe<M, 2> r;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i <= k) { acc += (uint64_t)a.l[i] * b.l[k - i]; GS_PIN(acc); }
    mont_low_column<M>(acc, m, k);
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) { acc += (uint64_t)a.l[i] * b.l[k - i]; GS_PIN(acc); }
    mont_high_column<M>(acc, m, k, r.l[k - NL]);
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// N chains of T terms each, strictly round-robin: chain c's multiply-adds are N - 1 instructions apart (N = 4 for two Fq2
// products side by side: both coordinates of both).  Measured (tools/ubench_mulmod.hip, cycles per product per SIMD):
// compiler-scheduled 1204, two chains 1166, three chains 1100.
template <class M, int N, int T>
GS_HD void dots_uniform(const Dot<T> (&d)[N], Fe<M, 2> (&r)[N]) {
  uint32_t m[N][NL];
  uint64_t acc[N];
#pragma unroll
  for (int c = 0; c < N; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i <= k) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int c = 0; c < N; ++c) { acc[c] += (uint64_t)d[c].a[t][i] * d[c].b[t][k - i]; GS_STEP(acc[c]); }
        }
      }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i < k) {
#pragma unroll
        for (int c = 0; c < N; ++c) { acc[c] += (uint64_t)m[c][i] * M::p(k - i); GS_STEP(acc[c]); }
      }
#pragma unroll
    for (int c = 0; c < N; ++c) m[c][k] = ((uint32_t)acc[c] * M::kPinv29) & LMASK;
#pragma unroll
    for (int c = 0; c < N; ++c) { acc[c] += (uint64_t)m[c][k] * M::p(0); GS_STEP(acc[c]); }
#pragma unroll
    for (int c = 0; c < N; ++c) acc[c] >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int c = 0; c < N; ++c) { acc[c] += (uint64_t)d[c].a[t][i] * d[c].b[t][k - i]; GS_STEP(acc[c]); }
        }
      }
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) {
#pragma unroll
        for (int c = 0; c < N; ++c) { acc[c] += (uint64_t)m[c][i] * M::p(k - i); GS_STEP(acc[c]); }
      }
#pragma unroll
    for (int c = 0; c < N; ++c) { r[c].l[k - NL] = (uint32_t)acc[c] & LMASK; acc[c] >>= LB; }
  }
#pragma unroll
  for (int c = 0; c < N; ++c) r[c].l[NL - 1] = (uint32_t)acc[c];
}

// two squares together (the doubled-operand trick of sqr: 45 products each)
template <class M, int Ba, int Bb>
GS_HD void sqr2(const Fe<M, Ba>& a, const Fe<M, Bb>& b, Fe<M, 2>& ra, Fe<M, 2>& rb) {
  static_assert(Ba * Ba <= 160 && Bb * Bb <= 160, "Montgomery square input bound exceeded");
  uint32_t ma[NL], mb[NL], a2[NL], b2[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) { a2[i] = a.l[i] << 1; b2[i] = b.l[i] << 1; }
  uint64_t xa = 0, xb = 0;
#define GS_SQR_PRODUCTS                                                                                    \
  _Pragma("unroll") for (int i = 0; i < NL; ++i) {                                                         \
    const int j = k - i;                                                                                   \
    if (j >= 0 && j < NL && i < j) {                                                                       \
      xa += (uint64_t)a2[i] * a.l[j]; GS_STEP(xa);                                                         \
      xb += (uint64_t)b2[i] * b.l[j]; GS_STEP(xb);                                                         \
    }                                                                                                      \
  }                                                                                                        \
  if ((k & 1) == 0) {                                                                                      \
    xa += (uint64_t)a.l[k / 2] * a.l[k / 2]; GS_STEP(xa);                                                  \
    xb += (uint64_t)b.l[k / 2] * b.l[k / 2]; GS_STEP(xb);                                                  \
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    GS_SQR_PRODUCTS
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i < k) {
        xa += (uint64_t)ma[i] * M::p(k - i); GS_STEP(xa);
        xb += (uint64_t)mb[i] * M::p(k - i); GS_STEP(xb);
      }
    ma[k] = ((uint32_t)xa * M::kPinv29) & LMASK;
    mb[k] = ((uint32_t)xb * M::kPinv29) & LMASK;
    xa += (uint64_t)ma[k] * M::p(0); GS_STEP(xa);
    xb += (uint64_t)mb[k] * M::p(0); GS_STEP(xb);
    xa >>= LB; xb >>= LB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
    GS_SQR_PRODUCTS
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (i >= k - (NL - 1)) {
        xa += (uint64_t)ma[i] * M::p(k - i); GS_STEP(xa);
        xb += (uint64_t)mb[i] * M::p(k - i); GS_STEP(xb);
      }
    ra.l[k - NL] = (uint32_t)xa & LMASK;
    rb.l[k - NL] = (uint32_t)xb & LMASK;
    xa >>= LB; xb >>= LB;
  }
#undef GS_SQR_PRODUCTS
  ra.l[NL - 1] = (uint32_t)xa;
  rb.l[NL - 1] = (uint32_t)xb;
}

// a*b - c*d with one reduction
template <class M, int Ba, int Bb, int Bc, int Bd>
GS_HD Fe<M, 2> mul_sub(const Fe<M, Ba>& a, const Fe<M, Bb>& b, const Fe<M, Bc>& c, const Fe<M, Bd>& d) {
  return mul_add(a, b, neg(c), d);
}

// ---- reductions / comparisons ---------------------------------------------------------------
// value -> value - floor(top/(p_top+1)) * p : lands in [0, 2p).
template <class M, int B, bool kNormal = false>
GS_HD Fe<M, 2> reduce2(const Fe<M, B>& a) {     // kNormal: the limbs are fully normalised already (sub_ripple): no carry pass
  Fe<M, B> x = a;
  if constexpr (!kNormal) carry_full(x);
  // q = floor(x_top / (p_top + 1)) <= floor(x / p); x < B p  =>  q <= B-1
  const uint32_t q = x.l[NL - 1] / (M::kTopLimb + 1u);
  Fe<M, 2> r;
  int32_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    // x_i < 2^29, q*p_i < 2^35 : 64-bit signed keeps it exact
    int64_t t = (int64_t)x.l[i] - (int64_t)((uint64_t)q * M::p(i)) + carry;
    r.l[i] = (uint32_t)t & LMASK;
    carry = (int32_t)(t >> LB);
  }
  r.l[NL - 1] = (uint32_t)((int64_t)x.l[NL - 1] - (int64_t)((uint64_t)q * M::p(NL - 1)) + carry);
  return r;
}

template <class M, int B> GS_HD Fe<M, 2> reduce2_normal(const Fe<M, B>& a) { return reduce2<M, B, true>(a); }

// canonical representative in [0, p), fully normalised limbs (boundary / comparisons only)
template <class M, int B>
GS_HD Fe<M, 1> canon(const Fe<M, B>& a) {
  Fe<M, B> x = a;
  carry_full(x);
  Fe<M, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = x.l[i];
  // subtract p while >= p: at most B-1 times (B small at every call site)
  for (int it = 0; it < B; ++it) {
    uint32_t t[NL];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; ++i) {
      int32_t d = (int32_t)r.l[i] - (int32_t)M::p(i) + borrow;
      t[i] = (uint32_t)d & LMASK;
      borrow = d >> LB;           // arithmetic shift: 0 or -1
    }
    int32_t top = (int32_t)r.l[NL - 1] - (int32_t)M::p(NL - 1) + borrow;
    t[NL - 1] = (uint32_t)top;
    const bool ge = top >= 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) r.l[i] = ge ? t[i] : r.l[i];
  }
  return r;
}

// Cheap necessary condition for value == 0 (mod p) on a NEARLY-NORMAL element (limb 0 < 2^29 exactly, limbs 1..7 < 2^29 + 16: what
// every carry_save leaves), without the rippling carry pass: value = k p has k = floor(top / p_top) for the fully carried top limb,
// which is l[8] or l[8] + 1, so k is one of two candidates and limb 0 -- exact as it stands -- must equal (k p) mod 2^29 for one of
// them.  Wrong with probability 2^-28 per candidate on a random element; is_zero then decides exactly.  (The zero test of P in
// every mixed addition took a full carry pass + the multiple check: ~30 instructions; this is ~9.)
template <class M, int B>
GS_HD bool maybe_zero(const Fe<M, B>& a) {
  const uint32_t k = a.l[NL - 1] / M::kTopLimb;
  const uint32_t c0 = (uint32_t)(k * M::p(0)) & LMASK, c1 = (c0 + M::p(0)) & LMASK;
  const uint32_t l0 = a.l[0] & LMASK;
  return l0 == c0 || l0 == c1;
}

template <class M, int B>
GS_HD bool is_zero(const Fe<M, B>& a) {        // value == 0 (mod p), exact
  if (!maybe_zero(a)) return false;
  Fe<M, B> x = a;
  carry_full(x);
  // if x = k p then k = floor(x_top / p_top) exactly (k < p_top)
  const uint32_t k = x.l[NL - 1] / M::kTopLimb;
  if ((((uint32_t)(k * M::p(0))) & LMASK) != x.l[0]) return false;     // 2^-29 false-positive filter
  uint64_t carry = 0;
  bool eq = true;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    uint64_t t = (uint64_t)k * M::p(i) + carry;
    eq = eq && (((uint32_t)t & LMASK) == x.l[i]);
    carry = t >> LB;
  }
  eq = eq && ((uint32_t)((uint64_t)k * M::p(NL - 1) + carry) == x.l[NL - 1]);
  return eq;
}

template <class M, int Ba, int Bb>
GS_HD bool equal(const Fe<M, Ba>& a, const Fe<M, Bb>& b) { return is_zero(sub(a, b)); }

template <class M, int B>
GS_HD Fe<M, B> select(bool c, const Fe<M, B>& a, const Fe<M, B>& b) {   // c ? a : b
  Fe<M, B> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

// ---- boundary conversions --------------------------------------------------------------------
// canonical little-endian 8 x u32 (= the C ABI's 4 x u64 limbs)  <->  29-bit limbs
// (any 256-bit input is accepted: 2^256 < 6p, hence the bound)
template <class M>
GS_HD Fe<M, 6> unpack32(const uint32_t (&w)[8]) {
  Fe<M, 6> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = LB * i, wi = bit >> 5, sh = bit & 31;
    uint32_t v = w[wi] >> sh;
    if (sh > 32 - LB && wi + 1 < 8) v |= w[wi + 1] << (32 - sh);
    r.l[i] = v & LMASK;
  }
  return r;
}

template <class M>
GS_HD void pack32(const Fe<M, 1>& a /* canonical, fully normalised */, uint32_t (&w)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, li = bit / LB, sh = bit % LB;
    uint32_t v = a.l[li] >> sh;
    if (li + 1 < NL) v |= a.l[li + 1] << (LB - sh);
    if (LB - sh + LB < 32 && li + 2 < NL) v |= a.l[li + 2] << (2 * LB - sh);
    w[j] = v;
  }
}

template <class M, int B>
GS_HD Fe<M, 2> to_mont(const Fe<M, B>& a) {    // a -> a R
  Fe<M, 1> r2;
#pragma unroll
  for (int i = 0; i < NL; ++i) r2.l[i] = M::r2(i);
  return mul(a, r2);
}

template <class M, int B>
GS_HD Fe<M, 1> from_mont(const Fe<M, B>& a) {  // a R -> a, canonical
  Fe<M, 1> one;
#pragma unroll
  for (int i = 0; i < NL; ++i) one.l[i] = (i == 0) ? 1u : 0u;
  return canon(mul(a, one));
}

// a^(p-2): Fermat inversion (reference: fq.go:66-67 ModInverse).  0 -> 0.
template <class M, int B>
GS_HD Fe<M, 2> inv(const Fe<M, B>& a) {
  static_assert(B <= 12, "inv input bound");
  Fe<M, 2> base = mul(a, fe_one<M>());
  Fe<M, 2> acc = relax<2>(fe_one<M>());
  // exponent p-2, scanned MSB first from the canonical 32-bit words
  for (int bit = M::kBits - 1; bit >= 0; --bit) {
    acc = sqr(acc);
    uint32_t wv = M::p32(bit >> 5);
    if ((bit >> 5) == 0) wv -= 2u;             // p is odd and p32(0) >= 2: no borrow
    if ((wv >> (bit & 31)) & 1u) acc = mul(acc, base);
  }
  return acc;
}

using FqE = Fe<ModQ, 2>;
using FrE = Fe<ModR, 2>;

}  // namespace gs
