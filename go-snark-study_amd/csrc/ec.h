// BN128 G1 / G2 group arithmetic for the MSM kernels, generic over the coordinate field
// (FqTag: G1 over Fq, Fq2Tag: G2 over Fq2).
//
// Replaces the reference's bn128/g1.go:32-170 and bn128/g2.go:32-200 (Jacobian add-2007-bl /
// dbl-2009-l / MSB-first double-and-add on math/big).  Device representation: buckets and
// partial sums are XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is 8M+2S instead of
// 11M+5S for Jacobian+Jacobian, bases are affine.  Unlike the reference's Add (no P==Q branch,
// g1.go:32-89) these formulas are COMPLETE: doubling and P + (-P) are handled.
// Parity with the reference is therefore defined on the affine normal form (g1.go:157-170).
#pragma once
#include "fp29.h"
#include "fq2.h"

namespace gs {

template <class T> struct BoundOf;
template <class M, int B> struct BoundOf<Fe<M, B>> { static constexpr int v = B; };
template <int B> struct BoundOf<Fq2e<B>> { static constexpr int v = B; };

struct FqTag {
  template <int B> using E = Fe<ModQ, B>;
  static constexpr int kWords = 8;                       // canonical storage: 8 x u32 per element
  static constexpr bool mul_ok(int a, int b) { return a * b <= 160; }
  static constexpr bool sqr_ok(int a) { return a * a <= 160; }
  static constexpr bool mul_sub_ok(int a, int b, int c, int d) { return a * b + (c + 1) * d <= 160; }
  static GS_HD E<1> one() { return fe_one<ModQ>(); }
  template <int B> static GS_HD E<B> zero() { return fe_zero<ModQ, B>(); }
  template <int B> static GS_HD bool limbs_all_zero(const E<B>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) o |= a.l[i];
    return o == 0;
  }
};

struct Fq2Tag {
  template <int B> using E = Fq2e<B>;
  static constexpr int kWords = 16;
  static constexpr bool mul_ok(int a, int b) { return a * b + a * (b + 1) <= 160 && 2 * a * b <= 160; }
  static constexpr bool sqr_ok(int a) { return (2 * a) * (2 * a + 1) <= 160; }
  static constexpr bool mul_sub_ok(int a, int b, int c, int d) { return a * b + (a + 1) * b + (c + 1) * d + (c + 1) * d <= 160; }
  static GS_HD E<1> one() { return fq2_one(); }
  template <int B> static GS_HD E<B> zero() { return fq2_zero<B>(); }
  template <int B> static GS_HD bool limbs_all_zero(const E<B>& a) {
    return FqTag::limbs_all_zero(a.c0) && FqTag::limbs_all_zero(a.c1);
  }
};

// bound-aware product / square: insert reduce2 at compile time where an operand is too lazy
template <class T, class A, class Bv>
GS_HD auto smul(const A& a, const Bv& b) {
  constexpr int Ba = BoundOf<A>::v, Bb = BoundOf<Bv>::v;
  if constexpr (T::mul_ok(Ba, Bb)) return mul(a, b);
  else if constexpr (Ba >= Bb && T::mul_ok(2, Bb)) return mul(reduce2(a), b);
  else if constexpr (T::mul_ok(Ba, 2)) return mul(a, reduce2(b));
  else return mul(reduce2(a), reduce2(b));
}
// bound-aware a*b - c*d with a single reduction per coordinate (reduce2 inserted where the dot product would overflow)
template <class T, class A, class Bv, class Cv, class Dv>
GS_HD auto smul_sub(const A& a, const Bv& b, const Cv& c, const Dv& d) {
  constexpr int Ba = BoundOf<A>::v, Bb = BoundOf<Bv>::v, Bc = BoundOf<Cv>::v, Bd = BoundOf<Dv>::v;
  if constexpr (T::mul_sub_ok(Ba, Bb, Bc, Bd)) return mul_sub(a, b, c, d);
  else if constexpr (T::mul_sub_ok(2, Bb, Bc, Bd)) return mul_sub(reduce2(a), b, c, d);
  else if constexpr (T::mul_sub_ok(2, 2, Bc, Bd)) return mul_sub(reduce2(a), reduce2(b), c, d);
  else return mul_sub(reduce2(a), reduce2(b), reduce2(c), reduce2(d));
}
template <class T, class A>
GS_HD auto ssqr(const A& a) {
  if constexpr (T::sqr_ok(BoundOf<A>::v)) return sqr(a);
  else return sqr(reduce2(a));
}

// ---- point types ------------------------------------------------------------------------------
template <class T>
struct Affine {                       // Montgomery form, canonical; infinity = (0, 0) (not on y^2 = x^3 + b)
  typename T::template E<1> x, y;
};

template <class T>
struct Xyzz {                         // infinity <=> all limbs of zz are zero
  typename T::template E<9> x;
  typename T::template E<5> y;
  typename T::template E<2> zz, zzz;
};

template <class T> GS_HD bool is_inf(const Affine<T>& p) { return T::limbs_all_zero(p.x) && T::limbs_all_zero(p.y); }
template <class T> GS_HD bool is_inf(const Xyzz<T>& p) { return T::limbs_all_zero(p.zz); }

template <class T>
GS_HD Xyzz<T> xyzz_inf() {
  Xyzz<T> r;
  r.x = T::template zero<9>(); r.y = T::template zero<5>();
  r.zz = T::template zero<2>(); r.zzz = T::template zero<2>();
  return r;
}

// acc = infinity in the G1 mixed addition: on the device the zeros come out of an opaque one-instruction asm INSIDE the rare
// branch.  With a plain `acc = xyzz_inf()` the compiler materialises the 36 zero phi inputs of the merged accumulator as `v_mov v, 0`
// in front of the branches of the cancellation test (72 static instructions); they turned out to sit behind the cheap first-level
// filter, so the DYNAMIC count is unchanged (SQ_INSTS_VALU: 2292 per addition before and after) -- but the register allocation of
// the common path improves: G1 accumulations 5.11 -> 4.98 ms per proof in one run (profiles/r02_ab_opaque_infinity.txt).  The Fq2
// instance and the general addition of the tails keep the plain form: there the same change moved spills around and LOST
// (G2 3.78 -> 3.94 ms, tails 7.6 -> 8.4 ms).
#ifndef GS_OPAQUE_INF
#define GS_OPAQUE_INF 1
#endif

// Which steps of the G2 mixed addition run two Fq2 products at once (four interleaved column chains) instead of one after the
// other (two chains): bit 0 U2|S2, bit 1 P^2|R^2, bit 2 P^3|Q, bit 3 ZZ3|ZZZ3.  Four chains fill more issue slots but hold more
// registers in a kernel that sits at 256 VGPRs and spills: all four steps (15) leave 112 B per lane in scratch, 9 leaves 36 B and
// measured 1-1.5 % faster (profiles/r02_ab_g2_chain_mask.txt); P^3|Q is the step that costs the registers.
#ifndef GS_G2_MASK
#define GS_G2_MASK 9
#endif

template <class M, int B> GS_HD void fill_limbs(Fe<M, B>& a, uint32_t z) {
#pragma unroll
  for (int i = 0; i < NL; ++i) a.l[i] = z;
}
template <int B> GS_HD void fill_limbs(Fq2e<B>& a, uint32_t z) { fill_limbs(a.c0, z); fill_limbs(a.c1, z); }
template <class T>
GS_HD void xyzz_set_inf(Xyzz<T>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && GS_OPAQUE_INF
  uint32_t z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  fill_limbs(a.x, z); fill_limbs(a.y, z); fill_limbs(a.zz, z); fill_limbs(a.zzz, z);
#else
  a = xyzz_inf<T>();
#endif
}

template <class T>
GS_HD Xyzz<T> xyzz_from_affine(const Affine<T>& a) {
  if (is_inf(a)) return xyzz_inf<T>();
  Xyzz<T> r;
  r.x = relax<9>(a.x); r.y = relax<5>(a.y);
  r.zz = relax<2>(T::one()); r.zzz = relax<2>(T::one());
  return r;
}

// 2 * (x, y) for a finite affine point, y given with bound 2       [dbl-2008-s-1 with ZZ = ZZZ = 1]
template <class T>
GS_HD Xyzz<T> xyzz_dbl_affine(const typename T::template E<1>& x, const typename T::template E<2>& y) {
  auto U = dbl(y);                                      // 4
  auto V = ssqr<T>(U);
  auto W = smul<T>(U, V);
  auto S = smul<T>(x, V);
  auto xx = ssqr<T>(x);
  auto M = add(dbl(xx), xx);                            // 6
  auto MM = ssqr<T>(M);
  Xyzz<T> r;
  auto X3 = sub(MM, dbl(S));                            // 2 + 4 + 1 = 7
  r.x = relax<9>(X3);
  auto t = smul<T>(M, sub(S, X3));                      // 6 x 10
  r.y = sub(t, smul<T>(W, y));                          // 5
  r.zz = V; r.zzz = W;
  return r;
}

// The G1 mixed addition (the instruction stream of k_bucket_accumulate<G1>: 92 % of a proof's device time).  Independent products
// share their issue slots (fp29.h, interleaved column chains): U2 | S2, P^2 | R^2, P^3 | Q, Y3 | ZZ3 | ZZZ3; and every sum or
// difference that is used once, as one factor of one product, skips its carry pass (fp29.h, Lz: -y2, Q - X3, -Y1), X3 takes one
// pass instead of three: 2292 -> ~2130 VALU instructions per addition.
GS_HD void xyzz_madd_g1(Xyzz<FqTag>& acc, const Affine<FqTag>& b, bool negate) {
  using T = FqTag;
  const auto y2 = select(negate, neg_lazy(b.y), widen<2>(as_lazy(relax<2>(b.y))));   // -(x,y) = (x, 2p - y): limbs < 2^30, value < 2p
  if (is_inf(acc)) {
    acc.x = relax<9>(b.x); acc.y = relax<5>(normalize(y2));
    acc.zz = relax<2>(T::one()); acc.zzz = relax<2>(T::one());
    return;
  }
  Fe<ModQ, 2> U2, S2;
  dots2<ModQ>(dot_of(b.x, acc.zz), dot_of(y2, acc.zzz), U2, S2);
  auto P = sub_ripple(U2, acc.x);                       // 2 + 9 + 1 = 12
  auto R = sub_ripple(S2, acc.y);                       // 2 + 5 + 1 = 8
  if (is_zero(P)) {
    if (is_zero(R)) acc = xyzz_dbl_affine<T>(b.x, normalize(y2));
    else xyzz_set_inf(acc);
    return;
  }
  Fe<ModQ, 2> PP, RR, PPP, Q, Y3, ZZ3, ZZZ3;
  sqr2(P, R, PP, RR);
  dots2<ModQ>(dot_of(P, PP), dot_of(acc.x, PP), PPP, Q);
  const auto X3 = sub_b_2c(RR, PPP, Q);                 // RR - PPP - 2 Q: 2 + 2 + 4 + 1 = 9, one carry pass
  const auto D = sub_lazy(Q, X3);                       // 2 + 10 = 12, limbs < 3 * 2^29
  const auto ny = neg_lazy(acc.y);                      // 6, limbs < 2^30
  dots3<ModQ>(dot_of(R, D, ny, PPP), dot_of(acc.zz, PP), dot_of(acc.zzz, PPP), Y3, ZZ3, ZZZ3);
  acc.zz = ZZ3; acc.zzz = ZZZ3;
  acc.x = X3; acc.y = relax<5>(Y3);
}

// acc += +-(x2, y2)   [madd-2008-s: 8M + 2S], complete.  negate: add -(x2, y2) instead.
template <class T>
GS_HD void xyzz_madd(Xyzz<T>& acc, const Affine<T>& b, bool negate = false) {
  if (is_inf(b)) return;
  if constexpr (GS_PAIR != 0 && T::kWords == 8) { xyzz_madd_g1(acc, b, negate); return; }
  const auto y2 = select(negate, neg(b.y), relax<2>(b.y));   // -(x,y) = (x, 2p - y)
  if (is_inf(acc)) {
    acc.x = relax<9>(b.x); acc.y = relax<5>(y2);
    acc.zz = relax<2>(T::one()); acc.zzz = relax<2>(T::one());
    return;
  }
  typename T::template E<2> U2, S2;
  if constexpr (GS_PAIR != 0 && T::kWords == 8) dots2<ModQ>(dot_of(b.x, acc.zz), dot_of(y2, acc.zzz), U2, S2);
  else if constexpr (GS_PAIR != 0 && (GS_G2_MASK & 1) != 0) mul2(b.x, acc.zz, y2, acc.zzz, U2, S2);
  else if constexpr (GS_PAIR != 0) { U2 = mul(b.x, acc.zz); S2 = mul(y2, acc.zzz); }
  else { U2 = smul<T>(b.x, acc.zz); S2 = smul<T>(y2, acc.zzz); }
  // (GS_PAIR: the accumulation kernels' instance takes the rippling difference, 8 instructions shorter per coordinate and fully
  // normalised -- which the Fq2 instance's reduce2 below would otherwise re-establish with a carry pass of its own)
  typename T::template E<12> P;
  typename T::template E<8> R;
  if constexpr (GS_PAIR != 0) { P = sub_ripple(U2, acc.x); R = sub_ripple(S2, acc.y); }
  else { P = sub(U2, acc.x); R = sub(S2, acc.y); }      // 2 + 9 + 1 = 12,  2 + 5 + 1 = 8
  if (is_zero(P)) {
    if (is_zero(R)) acc = xyzz_dbl_affine<T>(b.x, y2);
    else if constexpr (T::kWords == 8) xyzz_set_inf(acc);
    else acc = xyzz_inf<T>();
    return;
  }
  if constexpr (GS_PAIR != 0 && T::kWords == 8) {
    // G1: independent products share their issue slots (fp29.h, interleaved column chains): P^2 | R^2, P^3 | Q,
    // Y3 | ZZ3 | ZZZ3.
    typename T::template E<2> PP, RR, PPP, Q, Y3, ZZ3, ZZZ3;
    sqr2(P, R, PP, RR);
    dots2<ModQ>(dot_of(P, PP), dot_of(acc.x, PP), PPP, Q);
    auto X3 = sub(RR, add(PPP, dbl(Q)));                // 2 + 6 + 1 = 9
    const auto D = sub(Q, X3);                          // 2 + 9 + 1 = 12
    const auto ny = neg(acc.y);                         // 6
    dots3<ModQ>(dot_of(R, D, ny, PPP), dot_of(acc.zz, PP), dot_of(acc.zzz, PPP), Y3, ZZ3, ZZZ3);
    acc.zz = ZZ3; acc.zzz = ZZZ3;
    acc.x = X3; acc.y = relax<5>(Y3);
  } else if constexpr (GS_PAIR != 0) {
    // G2: two Fq2 products at a time = four chains (both coordinates of both): P^2 | R^2, P^3 | Q, ZZ3 | ZZZ3
    typename T::template E<2> PP, RR, PPP, Q, ZZ3, ZZZ3;
    const auto Pr = reduce2_normal(P), Rr = reduce2_normal(R);   // the Fq2 square takes (2a)(2a + 1) <= 160
    if constexpr ((GS_G2_MASK & 2) != 0) sqr2(Pr, Rr, PP, RR);
    else { PP = sqr(Pr); RR = sqr(Rr); }
    if constexpr ((GS_G2_MASK & 4) != 0) mul2(Pr, PP, acc.x, PP, PPP, Q);
    else { PPP = mul(Pr, PP); Q = mul(acc.x, PP); }
    auto X3 = sub_b_2c(RR, PPP, Q);                     // RR - PPP - 2 Q, one carry pass per coordinate: 2 + 2 + 4 + 1 = 9
    auto Y3 = mul_sub(Rr, sub(Q, X3), acc.y, PPP);      // (2, 12, 5, 2): two four-term chains
    if constexpr ((GS_G2_MASK & 8) != 0) mul2(acc.zz, PP, acc.zzz, PPP, ZZ3, ZZZ3);
    else { ZZ3 = mul(acc.zz, PP); ZZZ3 = mul(acc.zzz, PPP); }
    acc.zz = ZZ3; acc.zzz = ZZZ3;
    acc.x = X3; acc.y = relax<5>(Y3);
  } else {
    auto PP = ssqr<T>(P);
    auto PPP = smul<T>(P, PP);
    auto Q = smul<T>(acc.x, PP);
    auto RR = ssqr<T>(R);
    auto X3 = sub(RR, add(PPP, dbl(Q)));                // 2 + 6 + 1 = 9
    auto Y3 = smul_sub<T>(R, sub(Q, X3), acc.y, PPP);   // R (Q - X3) - Y1 PPP, one reduction per coordinate
    acc.zz = smul<T>(acc.zz, PP);
    acc.zzz = smul<T>(acc.zzz, PPP);
    acc.x = X3; acc.y = relax<5>(Y3);
  }
}

// The accumulator of k_bucket_accumulate<G2>: an Xyzz whose y is typed with bound 2 instead of 5.  Every y the accumulation loop
// produces IS below 2p (first point: +-y2; mixed addition: one Montgomery reduction per coordinate), only the doubling of the rare
// P == Q case leaves 5 -- and with y < 2p the difference R = S2 - Y1 stays below 5p, inside what the Fq2 square (2a)(2a + 1) <= 160 and
// the four-term Y3 take, so R needs no reduce2 (two coordinates x ~60 instructions per addition; P keeps its reduction: X1 < 9p).
template <class T>
struct XyzzAcc {
  typename T::template E<9> x;
  typename T::template E<2> y, zz, zzz;
};
template <class T> GS_HD bool is_inf(const XyzzAcc<T>& p) { return T::limbs_all_zero(p.zz); }
// Constants assigned to the accumulator inside a RARE branch (zeros of a reset, the limbs of `one` at a bucket's first point) are
// phi inputs of the merged accumulator, and the compiler materialises them as v_mov in FRONT of the branch: 72 of them per
// iteration of the G2 loop.  Passing each limb through an empty volatile asm inside the branch pins the v_mov there.
#ifndef GS_G2_OPAQUE
#define GS_G2_OPAQUE 1
#endif
template <class M, int B> GS_HD void pin_in_branch(Fe<M, B>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && GS_G2_OPAQUE
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(a.l[i]));
#else
  (void)a;
#endif
}
template <int B> GS_HD void pin_in_branch(Fq2e<B>& a) { pin_in_branch(a.c0); pin_in_branch(a.c1); }
template <class T> GS_HD XyzzAcc<T> xyzz_acc_inf() {
  XyzzAcc<T> r;
  r.x = T::template zero<9>(); r.y = T::template zero<2>(); r.zz = T::template zero<2>(); r.zzz = T::template zero<2>();
  pin_in_branch(r.x); pin_in_branch(r.y); pin_in_branch(r.zz); pin_in_branch(r.zzz);
  return r;
}
template <class T> GS_HD Xyzz<T> to_xyzz(const XyzzAcc<T>& a) {
  Xyzz<T> r;
  r.x = a.x; r.y = relax<5>(a.y); r.zz = a.zz; r.zzz = a.zzz;
  return r;
}
// acc += +-(x2, y2), the G2 steps of xyzz_madd below on the tighter accumulator
template <class T>
GS_HD void xyzz_madd(XyzzAcc<T>& acc, const Affine<T>& b, bool negate = false) {
  static_assert(GS_PAIR != 0 && T::kWords != 8, "the tight accumulator is the G2 accumulation kernel's");
  if (is_inf(b)) return;
  const auto y2 = select(negate, neg(b.y), relax<2>(b.y));
  if (is_inf(acc)) {
    acc.x = relax<9>(b.x); acc.y = y2;
    acc.zz = relax<2>(T::one()); acc.zzz = relax<2>(T::one());
    pin_in_branch(acc.zz); pin_in_branch(acc.zzz);
    return;
  }
  typename T::template E<2> U2, S2;
  if constexpr ((GS_G2_MASK & 1) != 0) mul2(b.x, acc.zz, y2, acc.zzz, U2, S2);
  else { U2 = mul(b.x, acc.zz); S2 = mul(y2, acc.zzz); }
  const auto P = sub_ripple(U2, acc.x);                 // 2 + 9 + 1 = 12
  const auto R = sub_ripple(S2, acc.y);                 // 2 + 2 + 1 = 5
  if (is_zero(P)) {
    if (is_zero(R)) {
      const Xyzz<T> d = xyzz_dbl_affine<T>(b.x, y2);
      acc.x = d.x; acc.y = reduce2(d.y); acc.zz = d.zz; acc.zzz = d.zzz;
    } else acc = xyzz_acc_inf<T>();
    return;
  }
  typename T::template E<2> PP, RR, PPP, Q, ZZ3, ZZZ3;
  const auto Pr = reduce2_normal(P);                    // the Fq2 square takes (2a)(2a + 1) <= 160: P < 12p does not fit, R < 5p does
  if constexpr ((GS_G2_MASK & 2) != 0) sqr2(Pr, R, PP, RR);
  else { PP = sqr(Pr); RR = sqr(R); }
  if constexpr ((GS_G2_MASK & 4) != 0) mul2(Pr, PP, acc.x, PP, PPP, Q);
  else { PPP = mul(Pr, PP); Q = mul(acc.x, PP); }
  const auto X3 = sub_b_2c(RR, PPP, Q);                 // RR - PPP - 2 Q, one carry pass per coordinate: 2 + 2 + 4 + 1 = 9
  const auto Y3 = mul_sub(R, sub(Q, X3), acc.y, PPP);   // (5, 12, 2, 2): 60 + 72 + 6 + 6 = 144 <= 160
  if constexpr ((GS_G2_MASK & 8) != 0) mul2(acc.zz, PP, acc.zzz, PPP, ZZ3, ZZZ3);
  else { ZZ3 = mul(acc.zz, PP); ZZZ3 = mul(acc.zzz, PPP); }
  acc.zz = ZZ3; acc.zzz = ZZZ3;
  acc.x = X3; acc.y = Y3;
}

// 2 * acc   [dbl-2008-s-1: 6M + 4S... a = 0]
template <class T>
GS_HD void xyzz_dbl(Xyzz<T>& acc) {
  if (is_inf(acc)) return;
  // (ordered for short live ranges -- V, W and y die as early as the formulas allow: the reduction tails run at <= 256 VGPRs)
  auto U = dbl(acc.y);                                  // 10
  auto V = ssqr<T>(U);
  auto W = smul<T>(U, V);
  auto S = smul<T>(acc.x, V);
  acc.zz = smul<T>(V, acc.zz);
  const auto Wy = smul<T>(W, acc.y);
  acc.zzz = smul<T>(W, acc.zzz);
  auto xx = ssqr<T>(acc.x);
  auto M = add(dbl(xx), xx);                            // 6
  auto MM = ssqr<T>(M);
  auto X3 = sub(MM, dbl(S));                            // 7
  auto t = smul<T>(M, sub(S, X3));
  acc.y = sub(t, Wy);                                   // 5
  acc.x = relax<9>(X3);
}

// ---- Jacobian doublings for the window-table builder (round 6) ------------------------------------------------------------------
// A table row is 2^(c j) P: 14 x 17 doublings of ONE point, never an addition -- and a Jacobian doubling on an a = 0 curve is 3M + 4S
// (A = X^2, B = Y^2, C = B^2, D = 4 X B, E = 3 A, X3 = E^2 - 2 D, Y3 = E (D - X3) - 8 C, Z3 = 2 Y Z: dbl-2009-l with its (X + B)^2 trick
// undone, a product is cheaper here than the squaring plus the bound growth of the two subtractions) = 990 multiply-adds against the 1350 of
// the XYZZ doubling (6M + 3S) the accumulators use.  Z = 0 <=> infinity.  The bounds are kept where the next doubling squares them
// without a reduction of its own; smul / ssqr insert one wherever a type says so (Fq2 squares need bound <= 6).
template <class T>
struct Jac {
  typename T::template E<7> x;
  typename T::template E<5> y;
  typename T::template E<4> z;
};
template <class T> GS_HD bool is_inf(const Jac<T>& p) { return T::limbs_all_zero(p.z); }
template <class T>
GS_HD Jac<T> jac_from_affine(const Affine<T>& a) {      // a finite
  Jac<T> r;
  r.x = relax<7>(a.x); r.y = relax<5>(a.y); r.z = relax<4>(T::one());
  return r;
}
template <class T>
GS_HD void jac_dbl(Jac<T>& p) {
  const auto A = ssqr<T>(p.x);
  const auto B = ssqr<T>(p.y);
  const auto C = ssqr<T>(B);
  const auto D = reduce2(dbl(dbl(smul<T>(p.x, B))));    // 4 X Y^2, back to bound 2: X3 and D - X3 stay small
  const auto E = add(dbl(A), A);                        // 6
  const auto F = ssqr<T>(E);
  const auto X3 = sub(F, dbl(D));                       // 2 + 4 + 1 = 7
  p.z = dbl(smul<T>(p.y, p.z));                         // 4 (the old y)
  const auto t = smul<T>(E, sub(D, X3));                // 6 x 10
  p.y = relax<5>(reduce2(sub(t, dbl(dbl(dbl(C))))));    // 2 + 16 + 1 = 19 -> 2
  p.x = X3;
}
// (X, Y, Z) -> affine with 1 / Z given
template <class T, class EI>
GS_HD Affine<T> jac_to_affine_with_inverse(const Jac<T>& p, const EI& zinv) {
  const auto i2 = ssqr<T>(zinv);
  const auto i3 = smul<T>(i2, zinv);
  Affine<T> r;
  r.x = canon(smul<T>(p.x, i2));
  r.y = canon(smul<T>(p.y, i3));
  return r;
}

// acc += b   [add-2008-s: 12M + 2S], complete
template <class T>
GS_HD void xyzz_add(Xyzz<T>& acc, const Xyzz<T>& b) {
  if (is_inf(b)) return;
  if (is_inf(acc)) { acc = b; return; }
  auto U1 = smul<T>(acc.x, b.zz);
  auto U2 = smul<T>(b.x, acc.zz);
  auto S1 = smul<T>(acc.y, b.zzz);
  auto S2 = smul<T>(b.y, acc.zzz);
  auto P = sub(U2, U1);                                 // 5
  auto R = sub(S2, S1);                                 // 5
  if (is_zero(P)) {
    if (is_zero(R)) xyzz_dbl(acc);
    else acc = xyzz_inf<T>();
    return;
  }
  auto PP = ssqr<T>(P);
  auto PPP = smul<T>(P, PP);
  auto Q = smul<T>(U1, PP);
  auto RR = ssqr<T>(R);
  auto X3 = sub(RR, add(PPP, dbl(Q)));                  // 9
  auto Y3 = relax<5>(smul_sub<T>(R, sub(Q, X3), S1, PPP));
  acc.zz = smul<T>(smul<T>(acc.zz, b.zz), PP);
  acc.zzz = smul<T>(smul<T>(acc.zzz, b.zzz), PPP);
  acc.x = X3; acc.y = Y3;
}

// acc += B for a point B that lives in MEMORY (HBM or LDS) as the raw limbs store_xyzz writes [x | y | zz | zzz], cw words per
// coordinate.  Same formulas as xyzz_add, ordered so that each coordinate of B is loaded where it is used and is dead two products
// later, with a compiler barrier in front of every load (the scheduler may not hoist the loads to the top and keep 72 / 144 more
// registers alive): the tail kernels of the MSM (bucket combine, reduction trees) hold ONE point in registers plus the temporaries
// of the formula -- round 3's two-operand form needed 317-388 VGPRs for G2, one wave per SIMD.
// The doubling case (B == acc as points) doubles B's copy in memory, so acc's own coordinates need not survive to the test.
#if defined(__HIP_DEVICE_COMPILE__)
#define GS_MEM_FENCE() asm volatile("" ::: "memory")
#else
#define GS_MEM_FENCE() do {} while (0)
#endif
template <class T, int B>
GS_HD typename T::template E<B> load_coord(const uint32_t* p) {
  typename T::template E<B> e;
  if constexpr (T::kWords == 8) {
#pragma unroll
    for (int i = 0; i < NL; ++i) e.l[i] = p[i];
  } else {
#pragma unroll
    for (int i = 0; i < NL; ++i) { e.c0.l[i] = p[i]; e.c1.l[i] = p[NL + i]; }
  }
  return e;
}
template <class T>
GS_HD Xyzz<T> load_point(const uint32_t* b) {
  constexpr int cw = (T::kWords == 8 ? 1 : 2) * NL;
  Xyzz<T> r;
  r.x = load_coord<T, 9>(b); r.y = load_coord<T, 5>(b + cw); r.zz = load_coord<T, 2>(b + 2 * cw); r.zzz = load_coord<T, 2>(b + 3 * cw);
  return r;
}
template <class T>
GS_HD void xyzz_add_mem(Xyzz<T>& acc, const uint32_t* b) {
  constexpr int cw = (T::kWords == 8 ? 1 : 2) * NL;
  // G1 has the registers to request all four coordinates at once (one memory latency per addition, as round 3's two-operand form);
  // G2 keeps ONE coordinate in flight ahead of the products that consume the previous one (the latency hides behind ~2 products).
  constexpr bool kAllAtOnce = T::kWords == 8;
  const auto bzz = load_coord<T, 2>(b + 2 * cw);
  auto bx = load_coord<T, 9>(b);
  typename T::template E<2> bzzz;
  typename T::template E<5> by;
  if constexpr (kAllAtOnce) { bzzz = load_coord<T, 2>(b + 3 * cw); by = load_coord<T, 5>(b + cw); }
  if (T::limbs_all_zero(bzz)) return;                   // B = infinity
  if (is_inf(acc)) { acc = load_point<T>(b); return; }
  const auto U1 = smul<T>(acc.x, bzz);
  const auto Tz = smul<T>(acc.zz, bzz);                 // ZZ1 ZZ2
  if constexpr (!kAllAtOnce) { GS_MEM_FENCE(); bzzz = load_coord<T, 2>(b + 3 * cw); }
  const auto U2 = smul<T>(bx, acc.zz);
  const auto P = sub(U2, U1);                           // 5
  if constexpr (!kAllAtOnce) { GS_MEM_FENCE(); by = load_coord<T, 5>(b + cw); }
  const auto S1 = smul<T>(acc.y, bzzz);
  const auto Vz = smul<T>(acc.zzz, bzzz);               // ZZZ1 ZZZ2
  if constexpr (!kAllAtOnce) GS_MEM_FENCE();
  const auto S2 = smul<T>(by, acc.zzz);
  const auto R = sub(S2, S1);                           // 5
  if (is_zero(P)) {
    if (is_zero(R)) { acc = load_point<T>(b); xyzz_dbl(acc); }
    else acc = xyzz_inf<T>();
    return;
  }
  const auto PP = ssqr<T>(P);
  const auto Q = smul<T>(U1, PP);
  const auto PPP = smul<T>(P, PP);
  acc.zz = smul<T>(Tz, PP);
  acc.zzz = smul<T>(Vz, PPP);
  const auto RR = ssqr<T>(R);
  const auto X3 = sub(RR, add(PPP, dbl(Q)));            // 9
  acc.y = relax<5>(smul_sub<T>(R, sub(Q, X3), S1, PPP));
  acc.x = X3;
}

template <class T>
GS_HD Xyzz<T> xyzz_neg(const Xyzz<T>& a) {
  Xyzz<T> r = a;
  if (!is_inf(a)) r.y = relax<5>(neg(reduce2(a.y)));    // p-bounded 3 -> fits 5
  return r;
}

// k * p for a 256-bit scalar given as 8 little-endian u32 words (prover tail, groth16.go:253-275) with a fixed 4-bit
// window: 64 windows of 4 doublings + 1 table addition
template <class T>
GS_HD Xyzz<T> xyzz_mul_words_w4(const Xyzz<T>& p, const uint32_t (&k)[8]) {
  Xyzz<T> tab[16];
  tab[0] = xyzz_inf<T>();
  tab[1] = p;
  for (int i = 2; i < 16; ++i) { tab[i] = tab[i - 1]; xyzz_add(tab[i], p); }
  Xyzz<T> r = xyzz_inf<T>();
  for (int nib = 63; nib >= 0; --nib) {
    for (int d = 0; d < 4; ++d) xyzz_dbl(r);
    const uint32_t v = (k[nib >> 3] >> ((nib & 7) * 4)) & 15u;
    if (v) xyzz_add(r, tab[v]);
  }
  return r;
}

// XYZZ -> affine (canonical Montgomery): one field inversion           [cf. g1.go:157-170]
template <class T>
GS_HD Affine<T> xyzz_to_affine(const Xyzz<T>& p) {
  Affine<T> r;
  if (is_inf(p)) { r.x = T::template zero<1>(); r.y = T::template zero<1>(); return r; }
  auto i3 = inv(p.zzz);                                 // 1/ZZZ
  auto zi = smul<T>(p.zz, i3);                          // ZZ/ZZZ ; (ZZ/ZZZ)^2 = 1/ZZ
  auto i2 = ssqr<T>(zi);
  auto x = smul<T>(p.x, i2);
  auto y = smul<T>(p.y, i3);
  r.x = canon(x); r.y = canon(y);
  return r;
}

// Jacobian (X, Y, Z) in Montgomery form -> affine: x = X/Z^2, y = Y/Z^3 (pk upload; the
// reference keeps pk points Jacobian with Z != 1, e.g. groth16.go:139-175)
template <class T, class EX>
GS_HD Affine<T> jacobian_to_affine(const EX& X, const EX& Y, const EX& Z) {
  Affine<T> r;
  if (is_zero(Z)) { r.x = T::template zero<1>(); r.y = T::template zero<1>(); return r; }
  auto zi = inv(Z);
  auto zi2 = ssqr<T>(zi);
  auto x = smul<T>(X, zi2);
  auto y = smul<T>(Y, smul<T>(zi2, zi));
  r.x = canon(x); r.y = canon(y);
  return r;
}

// y^2 == x^3 + b on E (b = 3) / on the twist E' (b = 3 / (9 + u)); infinity passes.  Uploads check it: a key point off its
// curve makes every later sum meaningless, silently (the reference never checks either, bn128/g1.go has no IsOnCurve).
template <class T> GS_HD typename T::template E<1> curve_b();
template <> GS_HD Fe<ModQ, 1> curve_b<FqTag>() {
  Fe<ModQ, 1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = Gen::b1(i);
  return r;
}
template <> GS_HD Fq2e<1> curve_b<Fq2Tag>() {
  Fq2e<1> r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { r.c0.l[i] = Gen::b2c0(i); r.c1.l[i] = Gen::b2c1(i); }
  return r;
}
template <class T>
GS_HD bool on_curve(const Affine<T>& a) {
  if (is_inf(a)) return true;
  const auto lhs = ssqr<T>(a.y);
  const auto rhs = add(smul<T>(ssqr<T>(a.x), a.x), curve_b<T>());
  return is_zero(sub(lhs, rhs));
}

using G1Affine = Affine<FqTag>;
using G2Affine = Affine<Fq2Tag>;
using G1Xyzz = Xyzz<FqTag>;
using G2Xyzz = Xyzz<Fq2Tag>;

}  // namespace gs
