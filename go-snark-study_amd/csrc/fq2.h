// Fq2 = Fq[u]/(u^2+1) on top of fp29.h.  Device replacement for the reference's
// fields/fq2.go:37-133 (`Fq2.Add/Sub/Neg/Mul/Square/Inverse`).  The reference stores the
// non-residue as q-1 and multiplies by it generically (bn128/bn128.go:86, fq2.go:32-34);
// here u^2 = -1 is applied as a subtraction, and each coordinate of a product is ONE
// Montgomery reduction of a two-term dot product (mul_add), i.e. 2 reductions per Fq2 product
// instead of Karatsuba's 3 products + 5 add/subs.
#pragma once
#include "fp29.h"

namespace gs {

template <int B>
struct Fq2e {
  Fe<ModQ, B> c0, c1;      // c0 + c1 * u
};

// single-coordinate alias with the same template shape, so curve code is generic over both
template <int B>
using Fq1e = Fe<ModQ, B>;

template <int BN, int B> GS_HD Fq2e<BN> relax(const Fq2e<B>& a) { return {relax<BN>(a.c0), relax<BN>(a.c1)}; }
template <int Ba, int Bb> GS_HD Fq2e<Ba + Bb> add(const Fq2e<Ba>& a, const Fq2e<Bb>& b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <int Ba, int Bb> GS_HD Fq2e<Ba + Bb + 1> sub(const Fq2e<Ba>& a, const Fq2e<Bb>& b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
template <int B> GS_HD Fq2e<2 * B> dbl(const Fq2e<B>& a) { return {dbl(a.c0), dbl(a.c1)}; }
template <int B> GS_HD Fq2e<B + 1> neg(const Fq2e<B>& a) { return {neg(a.c0), neg(a.c1)}; }
template <int B> GS_HD Fq2e<2> reduce2(const Fq2e<B>& a) { return {reduce2(a.c0), reduce2(a.c1)}; }
// the rippling difference (fp29.h: limbs come out fully normalised) and the reduction that relies on it
template <int Ba, int Bb> GS_HD Fq2e<Ba + Bb + 1> sub_ripple(const Fq2e<Ba>& a, const Fq2e<Bb>& b) { return {sub_ripple(a.c0, b.c0), sub_ripple(a.c1, b.c1)}; }
template <int B> GS_HD Fq2e<2> reduce2_normal(const Fq2e<B>& a) { return {reduce2_normal(a.c0), reduce2_normal(a.c1)}; }
// a - b - 2 c with one carry pass per coordinate (fp29.h)
template <int Ba, int Bb, int Bc> GS_HD Fq2e<Ba + Bb + 2 * Bc + 1> sub_b_2c(const Fq2e<Ba>& a, const Fq2e<Bb>& b, const Fq2e<Bc>& c) {
  return {sub_b_2c(a.c0, b.c0, c.c0), sub_b_2c(a.c1, b.c1, c.c1)};
}
template <int B> GS_HD Fq2e<1> canon(const Fq2e<B>& a) { return {canon(a.c0), canon(a.c1)}; }
template <int B> GS_HD bool is_zero(const Fq2e<B>& a) { return is_zero(a.c0) && is_zero(a.c1); }
template <int B> GS_HD Fq2e<B> select(bool c, const Fq2e<B>& a, const Fq2e<B>& b) { return {select(c, a.c0, b.c0), select(c, a.c1, b.c1)}; }

// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u        [fq2.go:63-76]
// The two coordinates are independent dot products: their column chains run interleaved (fp29.h, dots2).
template <int Ba, int Bb>
GS_HD Fq2e<2> mul(const Fq2e<Ba>& a, const Fq2e<Bb>& b) {
#if GS_PAIR
  const auto nb1 = neg_lazy(b.c1);                     // used once, as one factor of one term: no carry pass (fp29.h, Lz)
  Fq2e<2> r;
  dots2<ModQ>(dot_of(a.c0, b.c0, a.c1, nb1), dot_of(a.c0, b.c1, a.c1, b.c0), r.c0, r.c1);
  return r;
#else
  return {mul_add(a.c0, b.c0, a.c1, neg(b.c1)), mul_add(a.c0, b.c1, a.c1, b.c0)};
#endif
}

// a*b - c*d with ONE reduction per coordinate (four-term dot products): 2 x 405 mads instead of 2 x 486
template <int Ba, int Bb, int Bc, int Bd>
GS_HD Fq2e<2> mul_sub(const Fq2e<Ba>& a, const Fq2e<Bb>& b, const Fq2e<Bc>& c, const Fq2e<Bd>& d) {
#if GS_PAIR
  const auto na1 = neg_lazy(a.c1);
  const auto nc0 = neg_lazy(c.c0);
  const auto nc1 = neg_lazy(c.c1);
#else
  const auto na1 = neg(a.c1);
  const auto nc0 = neg(c.c0);
  const auto nc1 = neg(c.c1);
#endif
  // re: a0 b0 - a1 b1 - c0 d0 + c1 d1      im: a0 b1 + a1 b0 - c0 d1 - c1 d0
#if GS_PAIR
  Fq2e<2> r;
  dots2<ModQ>(dot_of(a.c0, b.c0, na1, b.c1, nc0, d.c0, c.c1, d.c1), dot_of(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0), r.c0, r.c1);
  return r;
#else
  return {dot4(a.c0, b.c0, na1, b.c1, nc0, d.c0, c.c1, d.c1), dot4(a.c0, b.c1, a.c1, b.c0, nc0, d.c1, nc1, d.c0)};
#endif
}

// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u                       [fq2.go:118-133]
template <int B>
GS_HD Fq2e<2> sqr(const Fq2e<B>& a) {
#if GS_PAIR
  const auto s = add_lazy(a.c0, a.c1);                 // limb weights 2 x 3 + the reduction = 7 of the column's 7.1 (fp29.h)
  const auto d = sub_lazy(a.c0, a.c1);
  const auto t = dbl_lazy(a.c0);
  Fq2e<2> r;
  dots2<ModQ>(dot_of(s, d), dot_of(t, a.c1), r.c0, r.c1);
  return r;
#else
  return {mul(add(a.c0, a.c1), sub(a.c0, a.c1)), mul(dbl(a.c0), a.c1)};
#endif
}

// Two independent Fq2 products / squares side by side: four interleaved column chains (fp29.h, dots_uniform).
template <int Ba, int Bb, int Bc, int Bd>
GS_HD void mul2(const Fq2e<Ba>& a, const Fq2e<Bb>& b, const Fq2e<Bc>& c, const Fq2e<Bd>& d, Fq2e<2>& ab, Fq2e<2>& cd) {
  const auto nb1 = neg_lazy(b.c1);
  const auto nd1 = neg_lazy(d.c1);
  const Dot<2> ch[4] = {dot_of(a.c0, b.c0, a.c1, nb1), dot_of(a.c0, b.c1, a.c1, b.c0), dot_of(c.c0, d.c0, c.c1, nd1), dot_of(c.c0, d.c1, c.c1, d.c0)};
  Fe<ModQ, 2> r[4];
  dots_uniform<ModQ, 4, 2>(ch, r);
  ab.c0 = r[0]; ab.c1 = r[1]; cd.c0 = r[2]; cd.c1 = r[3];
}
template <int Ba, int Bb>
GS_HD void sqr2(const Fq2e<Ba>& a, const Fq2e<Bb>& b, Fq2e<2>& aa, Fq2e<2>& bb) {
  const auto sa = add_lazy(a.c0, a.c1);
  const auto sb = add_lazy(b.c0, b.c1);
  const auto da = sub_lazy(a.c0, a.c1);
  const auto db = sub_lazy(b.c0, b.c1);
  const auto ta = dbl_lazy(a.c0);
  const auto tb = dbl_lazy(b.c0);
  const Dot<1> ch[4] = {dot_of(sa, da), dot_of(ta, a.c1), dot_of(sb, db), dot_of(tb, b.c1)};
  Fe<ModQ, 2> r[4];
  dots_uniform<ModQ, 4, 1>(ch, r);
  aa.c0 = r[0]; aa.c1 = r[1]; bb.c0 = r[2]; bb.c1 = r[3];
}

// 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2)                        [fq2.go:99-110]
template <int B>
GS_HD Fq2e<2> inv(const Fq2e<B>& a) {
  Fe<ModQ, 2> n = mul_add(a.c0, a.c0, a.c1, a.c1);
  Fe<ModQ, 2> ni = inv(n);
  return {mul(a.c0, ni), mul(neg(a.c1), ni)};
}

template <int B> GS_HD Fq2e<B> fq2_zero() { return {fe_zero<ModQ, B>(), fe_zero<ModQ, B>()}; }
GS_HD Fq2e<1> fq2_one() { return {fe_one<ModQ>(), fe_zero<ModQ, 1>()}; }

}  // namespace gs
