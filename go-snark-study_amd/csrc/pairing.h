// BN254 (alt_bn128) optimal ate pairing for the verifier, host only ("multi-pairing with shared final
// exponentiation, proper cyclotomic final exp instead of one 2790-bit Fq12.Exp -- CPU C++ first").
//
// What the reference computes (bn128/bn128.go:179-421): Pairing(P, Q) = MillerLoop(...)^((q^12 - 1)/r) with the tower
//   Fq2 = Fq[u]/(u^2 + 1), Fq6 = Fq2[v]/(v^3 - (9 + u)), Fq12 = Fq6[w]/(w^2 - v)        (bn128.go:86-97)
// and a verifier that compares products of such values (groth16.go:281-305, snark.go:292-368).
//
// What is here, derived from the definitions rather than from that code:
//   * Fq as 4 x 64-bit Montgomery words (the verifier is O(1) work per proof: a few thousand Fq products, latency
//     bound; it runs on a host core next to the host tail of the prover and needs no device),
//   * a multi-Miller loop over k pairs on affine twist coordinates, one shared squaring per bit and ONE batched
//     inversion per step for all pairs,
//   * lines in the sparse form y_P + (-lambda x_P) w + (lambda x_T - y_T) w^3 (they differ from the reference's by a
//     factor in Fq4, which the final exponentiation kills),
//   * final exponentiation = easy part (q^6 - 1)(q^2 + 1) by conjugation/Frobenius/one inversion, hard part
//     (q^4 - q^2 + 1)/r by the BN addition chain over x = 4965661367192848881 (three 63-bit exponentiations);
//     tools/gen_constants.py checks that this chain's exponent IS (q^4 - q^2 + 1)/r, so the value equals the
//     reference's Fq12.Exp(f, FinalExp) bit for bit (tests compare it with the oracle's restatement).
#pragma once
#include <stdint.h>

#include <cstring>
#include <vector>

namespace gs {
namespace pairing {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------ Fq
static const uint64_t kP[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t kX = 0x44e992b44a6909f1ULL;                          // BN parameter x
static const uint64_t kLoop[2] = {0x9d797039be763ba8ULL, 0x1ULL};          // 6x + 2 (65 bits), bn128.go:122
static const uint64_t kPm1Div6[4] = {0x34b017592414d4e1ULL, 0xee9591c2e6bda1c2ULL, 0xf40d60f3c0403964ULL, 0x0810b7bdd032f006ULL};

struct Fp { uint64_t v[4]; };

inline bool fp_is_zero(const Fp& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline bool fp_eq(const Fp& a, const Fp& b) { return a.v[0] == b.v[0] && a.v[1] == b.v[1] && a.v[2] == b.v[2] && a.v[3] == b.v[3]; }
inline bool geq_p(const uint64_t* a) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > kP[i]) return true;
    if (a[i] < kP[i]) return false;
  }
  return true;
}
inline void sub_p(uint64_t* a) {
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - kP[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
inline Fp fp_add(const Fp& a, const Fp& b) {
  Fp r;
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (uint64_t)c;
    c >>= 64;
  }
  if (c || geq_p(r.v)) sub_p(r.v);       // p < 2^254: no carry out in practice
  return r;
}
inline Fp fp_sub(const Fp& a, const Fp& b) {
  Fp r;
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)br;
    r.v[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (u128)r.v[i] + kP[i];
      r.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
inline Fp fp_neg(const Fp& a) {
  if (fp_is_zero(a)) return a;
  Fp p;
  memcpy(p.v, kP, sizeof kP);
  return fp_sub(p, a);
}
inline Fp fp_dbl(const Fp& a) { return fp_add(a, a); }

struct FpConsts {
  uint64_t n0;     // -p^-1 mod 2^64
  Fp r2;           // 2^512 mod p
  Fp one;          // 2^256 mod p
  Fp r3;           // 2^768 mod p
};
inline const FpConsts& fpc() {
  static const FpConsts c = [] {
    FpConsts k;
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - kP[0] * inv;       // Newton: p^-1 mod 2^64
    k.n0 = 0 - inv;
    Fp t = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; ++i) t = fp_add(t, t);
    k.one = t;
    for (int i = 0; i < 256; ++i) t = fp_add(t, t);
    k.r2 = t;
    for (int i = 0; i < 256; ++i) t = fp_add(t, t);
    k.r3 = t;
    return k;
  }();
  return c;
}

// Montgomery product a b 2^-256 mod p (CIOS, 4 words)
inline Fp fp_mul(const Fp& a, const Fp& b) {
  const uint64_t n0 = fpc().n0;
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a.v[j] * b.v[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * n0;
    c = (u128)m * kP[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * kP[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  Fp r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_p(r.v)) sub_p(r.v);
  return r;
}
inline Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }
inline Fp fp_from_std(const uint64_t w[4]) {        // any 256-bit value -> Montgomery residue
  Fp a = {{w[0], w[1], w[2], w[3]}};
  return fp_mul(a, fpc().r2);                        // (a)(R^2) R^-1 = a R  (fp_mul accepts a < 2^256)
}
inline void fp_to_std(const Fp& a, uint64_t w[4]) {
  Fp one = {{1, 0, 0, 0}};
  Fp r = fp_mul(a, one);
  memcpy(w, r.v, 32);
}
inline Fp fp_small(uint64_t k) {
  uint64_t w[4] = {k, 0, 0, 0};
  return fp_from_std(w);
}
template <int N>
inline Fp fp_pow(const Fp& a, const uint64_t (&e)[N]) {
  Fp r = fpc().one;
  for (int i = N * 64 - 1; i >= 0; --i) {
    r = fp_sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = fp_mul(r, a);
  }
  return r;
}
inline bool limbs_is_one(const uint64_t* a) { return a[0] == 1 && (a[1] | a[2] | a[3]) == 0; }
inline bool limbs_geq(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > b[i]) return true;
    if (a[i] < b[i]) return false;
  }
  return true;
}
inline void limbs_sub(uint64_t* a, const uint64_t* b) {      // a -= b, a >= b
  u128 br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - (uint64_t)br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
inline void limbs_shr1(uint64_t* a) {
  for (int i = 0; i < 3; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
  a[3] >>= 1;
}
inline void half_mod_p(Fp& x) {                              // x / 2 mod p (x < p < 2^254: x + p fits 4 words)
  if (x.v[0] & 1) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
      c += (u128)x.v[i] + kP[i];
      x.v[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  limbs_shr1(x.v);
}
// Inverse by the binary extended Euclid (a few hundred shift/subtract rounds on 4 words instead of the ~380 Montgomery
// products of a^(p-2)); the multi-Miller loop does one inversion per step, so this is its critical path.  0 -> 0.
inline Fp fp_inv(const Fp& a) {
  if (fp_is_zero(a)) return a;
  uint64_t u[4], v[4];
  memcpy(u, a.v, 32);
  memcpy(v, kP, 32);
  Fp x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  while (!limbs_is_one(u) && !limbs_is_one(v)) {
    while (!(u[0] & 1)) { limbs_shr1(u); half_mod_p(x1); }
    while (!(v[0] & 1)) { limbs_shr1(v); half_mod_p(x2); }
    if (limbs_geq(u, v)) { limbs_sub(u, v); x1 = fp_sub(x1, x2); }
    else { limbs_sub(v, u); x2 = fp_sub(x2, x1); }
  }
  // (a R)^-1 as a plain integer -> Montgomery form of a^-1: times R^3, one Montgomery product
  return fp_mul(limbs_is_one(u) ? x1 : x2, fpc().r3);
}

// ------------------------------------------------------------------------------------------------ Fq2
struct Fp2 { Fp c0, c1; };
inline Fp2 f2_zero() { return Fp2{Fp{{0, 0, 0, 0}}, Fp{{0, 0, 0, 0}}}; }
inline Fp2 f2_one() { return Fp2{fpc().one, Fp{{0, 0, 0, 0}}}; }
inline bool f2_is_zero(const Fp2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
inline bool f2_eq(const Fp2& a, const Fp2& b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
inline Fp2 f2_add(const Fp2& a, const Fp2& b) { return Fp2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
inline Fp2 f2_sub(const Fp2& a, const Fp2& b) { return Fp2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
inline Fp2 f2_neg(const Fp2& a) { return Fp2{fp_neg(a.c0), fp_neg(a.c1)}; }
inline Fp2 f2_dbl(const Fp2& a) { return f2_add(a, a); }
inline Fp2 f2_conj(const Fp2& a) { return Fp2{a.c0, fp_neg(a.c1)}; }
inline Fp2 f2_mul(const Fp2& a, const Fp2& b) {      // u^2 = -1
  Fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
  Fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
  return Fp2{fp_sub(t0, t1), fp_sub(fp_sub(s, t0), t1)};
}
inline Fp2 f2_sqr(const Fp2& a) {
  Fp t = fp_mul(a.c0, a.c1);
  return Fp2{fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_dbl(t)};
}
inline Fp2 f2_mul_fp(const Fp2& a, const Fp& k) { return Fp2{fp_mul(a.c0, k), fp_mul(a.c1, k)}; }
inline Fp2 f2_mul_xi(const Fp2& a) {                 // (9 + u)(a0 + a1 u) = (9 a0 - a1) + (9 a1 + a0) u
  Fp a0_8 = fp_dbl(fp_dbl(fp_dbl(a.c0))), a1_8 = fp_dbl(fp_dbl(fp_dbl(a.c1)));
  return Fp2{fp_sub(fp_add(a0_8, a.c0), a.c1), fp_add(fp_add(a1_8, a.c1), a.c0)};
}
inline Fp2 f2_inv(const Fp2& a) {
  Fp n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
  return Fp2{fp_mul(a.c0, n), fp_neg(fp_mul(a.c1, n))};
}
template <int N>
inline Fp2 f2_pow(const Fp2& a, const uint64_t (&e)[N]) {
  Fp2 r = f2_one();
  for (int i = N * 64 - 1; i >= 0; --i) {
    r = f2_sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = f2_mul(r, a);
  }
  return r;
}
// all entries must be non-zero; Montgomery's trick: one Fq inversion for the whole batch
inline void f2_batch_inv(std::vector<Fp2>& v) {
  const size_t n = v.size();
  if (!n) return;
  std::vector<Fp2> pre(n);
  Fp2 acc = f2_one();
  for (size_t i = 0; i < n; ++i) {
    pre[i] = acc;
    acc = f2_mul(acc, v[i]);
  }
  Fp2 inv = f2_inv(acc);
  for (size_t i = n; i-- > 0;) {
    Fp2 t = f2_mul(inv, pre[i]);
    inv = f2_mul(inv, v[i]);
    v[i] = t;
  }
}

// ------------------------------------------------------------------------------------------------ Fq6 = Fq2[v]/(v^3 - xi)
struct Fp6 { Fp2 c0, c1, c2; };
inline Fp6 f6_zero() { return Fp6{f2_zero(), f2_zero(), f2_zero()}; }
inline Fp6 f6_one() { return Fp6{f2_one(), f2_zero(), f2_zero()}; }
inline Fp6 f6_add(const Fp6& a, const Fp6& b) { return Fp6{f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; }
inline Fp6 f6_sub(const Fp6& a, const Fp6& b) { return Fp6{f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; }
inline Fp6 f6_neg(const Fp6& a) { return Fp6{f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; }
inline bool f6_eq(const Fp6& a, const Fp6& b) { return f2_eq(a.c0, b.c0) && f2_eq(a.c1, b.c1) && f2_eq(a.c2, b.c2); }
inline Fp6 f6_mul(const Fp6& a, const Fp6& b) {
  Fp2 t0 = f2_mul(a.c0, b.c0), t1 = f2_mul(a.c1, b.c1), t2 = f2_mul(a.c2, b.c2);
  Fp2 c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), t1), t2)));
  Fp2 c1 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), t0), t1), f2_mul_xi(t2));
  Fp2 c2 = f2_add(f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), t0), t2), t1);
  return Fp6{c0, c1, c2};
}
// a * (b0 + b1 v): the shape of a line's w-part
inline Fp6 f6_mul_01(const Fp6& a, const Fp2& b0, const Fp2& b1) {
  Fp2 t0 = f2_mul(a.c0, b0), t1 = f2_mul(a.c1, b1);
  Fp2 c0 = f2_add(t0, f2_mul_xi(f2_mul(a.c2, b1)));
  Fp2 c1 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b0, b1)), t0), t1);
  Fp2 c2 = f2_add(f2_mul(a.c2, b0), t1);
  return Fp6{c0, c1, c2};
}
inline Fp6 f6_mul_fp(const Fp6& a, const Fp& k) { return Fp6{f2_mul_fp(a.c0, k), f2_mul_fp(a.c1, k), f2_mul_fp(a.c2, k)}; }
inline Fp6 f6_mul_v(const Fp6& a) { return Fp6{f2_mul_xi(a.c2), a.c0, a.c1}; }
inline Fp6 f6_inv(const Fp6& a) {
  Fp2 A = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
  Fp2 B = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
  Fp2 C = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
  Fp2 F = f2_add(f2_mul(a.c0, A), f2_mul_xi(f2_add(f2_mul(a.c2, B), f2_mul(a.c1, C))));
  Fp2 Fi = f2_inv(F);
  return Fp6{f2_mul(A, Fi), f2_mul(B, Fi), f2_mul(C, Fi)};
}

// ------------------------------------------------------------------------------------------------ Fq12 = Fq6[w]/(w^2 - v)
struct Fp12 { Fp6 a0, a1; };
inline Fp12 f12_one() { return Fp12{f6_one(), f6_zero()}; }
inline bool f12_eq(const Fp12& a, const Fp12& b) { return f6_eq(a.a0, b.a0) && f6_eq(a.a1, b.a1); }
inline Fp12 f12_mul(const Fp12& a, const Fp12& b) {
  Fp6 t0 = f6_mul(a.a0, b.a0), t1 = f6_mul(a.a1, b.a1);
  Fp6 c1 = f6_sub(f6_sub(f6_mul(f6_add(a.a0, a.a1), f6_add(b.a0, b.a1)), t0), t1);
  return Fp12{f6_add(t0, f6_mul_v(t1)), c1};
}
inline Fp12 f12_sqr(const Fp12& a) {                 // complex squaring over Fq6
  Fp6 t = f6_mul(a.a0, a.a1);
  Fp6 c0 = f6_sub(f6_sub(f6_mul(f6_add(a.a0, a.a1), f6_add(a.a0, f6_mul_v(a.a1))), t), f6_mul_v(t));
  return Fp12{c0, f6_add(t, t)};
}
inline Fp12 f12_conj(const Fp12& a) { return Fp12{a.a0, f6_neg(a.a1)}; }     // = a^(q^6)
inline Fp12 f12_inv(const Fp12& a) {
  Fp6 d = f6_inv(f6_sub(f6_mul(a.a0, a.a0), f6_mul_v(f6_mul(a.a1, a.a1))));
  return Fp12{f6_mul(a.a0, d), f6_neg(f6_mul(a.a1, d))};
}
// f * (y + (l1 + l3 v) w),  y in Fq: one line of the Miller loop
inline Fp12 f12_mul_line(const Fp12& f, const Fp& y, const Fp2& l1, const Fp2& l3) {
  Fp6 c0 = f6_add(f6_mul_fp(f.a0, y), f6_mul_v(f6_mul_01(f.a1, l1, l3)));
  Fp6 c1 = f6_add(f6_mul_fp(f.a1, y), f6_mul_01(f.a0, l1, l3));
  return Fp12{c0, c1};
}

struct TowerConsts {
  Fp2 g1[6], g2[6], g3[6];     // gamma_k^i, gamma_1 = xi^((q-1)/6) = w^(q-1), gamma_2 = w^(q^2-1), gamma_3 = w^(q^3-1)
  Fp2 b_twist;                 // 3 / xi
  Fp three;
};
inline const TowerConsts& twc() {
  static const TowerConsts c = [] {
    TowerConsts k;
    Fp2 xi{fp_small(9), fp_small(1)};
    Fp2 gam1 = f2_pow(xi, kPm1Div6);
    Fp2 gam2 = f2_mul(gam1, f2_conj(gam1));          // gamma_1^(q+1): the norm, in Fq
    Fp2 gam3 = f2_mul(gam1, gam2);                   // gamma_1^(q^2+q+1) = gamma_1^2 conj(gamma_1)
    k.g1[0] = k.g2[0] = k.g3[0] = f2_one();
    for (int i = 1; i < 6; ++i) {
      k.g1[i] = f2_mul(k.g1[i - 1], gam1);
      k.g2[i] = f2_mul(k.g2[i - 1], gam2);
      k.g3[i] = f2_mul(k.g3[i - 1], gam3);
    }
    k.three = fp_small(3);
    k.b_twist = f2_mul_fp(f2_inv(xi), k.three);
    return k;
  }();
  return c;
}
// f = sum_i c_i w^i with (a0 = c0, c2, c4; a1 = c1, c3, c5);  f^(q^k) = sum_i frob_k(c_i) gamma_k^i w^i
inline Fp12 f12_frob(const Fp12& f, int k) {
  const TowerConsts& t = twc();
  const Fp2* g = k == 1 ? t.g1 : (k == 2 ? t.g2 : t.g3);
  const bool cj = (k & 1) != 0;
  auto m = [&](const Fp2& c, int i) { return f2_mul(cj ? f2_conj(c) : c, g[i]); };
  return Fp12{Fp6{m(f.a0.c0, 0), m(f.a0.c1, 2), m(f.a0.c2, 4)}, Fp6{m(f.a1.c0, 1), m(f.a1.c1, 3), m(f.a1.c2, 5)}};
}
// Squaring of a UNITARY element (f^(q^6+1) = 1, true after the easy part of the final exponentiation) by the
// Granger-Scott formulas: write f = A + B w + C w^2 over Fq4 = Fq2[s]/(s^2 - xi), s = w^3, with
// A = (c0, c3), B = (c1, c4), C = (c2, c5) in the w-power coefficients; then
//   f^2 = (3 A^2 - 2 conj(A)) + (3 s C^2 + 2 conj(B)) w + (3 B^2 - 2 conj(C)) w^2        (9 Fq2 squarings)
inline void f4_sqr(const Fp2& x0, const Fp2& x1, Fp2& r0, Fp2& r1) {
  Fp2 t0 = f2_sqr(x0), t1 = f2_sqr(x1);
  r1 = f2_sub(f2_sub(f2_sqr(f2_add(x0, x1)), t0), t1);
  r0 = f2_add(t0, f2_mul_xi(t1));
}
inline Fp12 f12_cyclotomic_sqr(const Fp12& f) {
  const Fp2 &c0 = f.a0.c0, &c1 = f.a1.c0, &c2 = f.a0.c1, &c3 = f.a1.c1, &c4 = f.a0.c2, &c5 = f.a1.c2;
  Fp2 t0, t1, u0, u1, v0, v1;
  f4_sqr(c0, c3, t0, t1);      // A^2
  f4_sqr(c2, c5, u0, u1);      // C^2
  f4_sqr(c1, c4, v0, v1);      // B^2
  auto three_minus_two = [](const Fp2& t, const Fp2& c) { Fp2 d = f2_sub(t, c); return f2_add(f2_dbl(d), t); };   // 3t - 2c
  auto three_plus_two = [](const Fp2& t, const Fp2& c) { Fp2 d = f2_add(t, c); return f2_add(f2_dbl(d), t); };    // 3t + 2c
  Fp2 n0 = three_minus_two(t0, c0), n3 = three_plus_two(t1, c3);
  Fp2 n1 = three_plus_two(f2_mul_xi(u1), c1), n4 = three_minus_two(u0, c4);
  Fp2 n2 = three_minus_two(v0, c2), n5 = three_plus_two(v1, c5);
  return Fp12{Fp6{n0, n2, n4}, Fp6{n1, n3, n5}};
}
inline Fp12 f12_pow_x(const Fp12& a) {               // a^x for unitary a, x = 63 bits
  Fp12 r = a;
  for (int i = 61; i >= 0; --i) {                    // bit 62 is the top set bit of kX
    r = f12_cyclotomic_sqr(r);
    if ((kX >> i) & 1) r = f12_mul(r, a);
  }
  return r;
}
// f^((q^12 - 1)/r)
inline Fp12 final_exponentiation(const Fp12& f) {
  // easy part: f^((q^6 - 1)(q^2 + 1)); afterwards the element is unitary, inverse = conjugate
  Fp12 t = f12_mul(f12_conj(f), f12_inv(f));
  t = f12_mul(f12_frob(t, 2), t);
  // hard part (q^4 - q^2 + 1)/r = (q + q^2 + q^3) - 2 + 6 x^2 q^2 - 12 x q - 18 (x + x^2 q) - 30 x^2 - 36 (x^3 + x^3 q)
  Fp12 fx = f12_pow_x(t), fx2 = f12_pow_x(fx), fx3 = f12_pow_x(fx2);
  Fp12 y0 = f12_mul(f12_mul(f12_frob(t, 1), f12_frob(t, 2)), f12_frob(t, 3));
  Fp12 y1 = f12_conj(t);
  Fp12 y2 = f12_frob(fx2, 2);
  Fp12 y3 = f12_conj(f12_frob(fx, 1));
  Fp12 y4 = f12_conj(f12_mul(fx, f12_frob(fx2, 1)));
  Fp12 y5 = f12_conj(fx2);
  Fp12 y6 = f12_conj(f12_mul(fx3, f12_frob(fx3, 1)));
  // y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36 by a vector addition chain
  Fp12 t0 = f12_cyclotomic_sqr(y6);
  t0 = f12_mul(t0, y4);
  t0 = f12_mul(t0, y5);
  Fp12 t1 = f12_mul(y3, y5);
  t1 = f12_mul(t1, t0);
  t0 = f12_mul(t0, y2);
  t1 = f12_cyclotomic_sqr(t1);
  t1 = f12_mul(t1, t0);
  t1 = f12_cyclotomic_sqr(t1);
  t0 = f12_mul(t1, y1);
  t1 = f12_mul(t1, y0);
  t0 = f12_cyclotomic_sqr(t0);
  return f12_mul(t0, t1);
}

// ------------------------------------------------------------------------------------------------ curve points
struct G1Aff { Fp x, y; bool inf; };
struct G2Aff { Fp2 x, y; bool inf; };

inline bool g1_on_curve(const G1Aff& p) {
  if (p.inf) return true;
  return fp_eq(fp_sqr(p.y), fp_add(fp_mul(fp_sqr(p.x), p.x), twc().three));
}
inline bool g2_on_curve(const G2Aff& q) {
  if (q.inf) return true;
  return f2_eq(f2_sqr(q.y), f2_add(f2_mul(f2_sqr(q.x), q.x), twc().b_twist));
}
// Jacobian triple in standard form (12 words) -> affine, g1.go:157-170 (Z = 0 -> infinity, g1.go:28-30)
inline G1Aff g1_from_jacobian_std(const uint64_t* w) {
  Fp X = fp_from_std(w), Y = fp_from_std(w + 4), Z = fp_from_std(w + 8);
  if (fp_is_zero(Z)) return G1Aff{X, Y, true};
  if (fp_eq(Z, fpc().one)) return G1Aff{X, Y, false};
  Fp zi = fp_inv(Z), zi2 = fp_sqr(zi);
  return G1Aff{fp_mul(X, zi2), fp_mul(Y, fp_mul(zi2, zi)), false};
}
inline G2Aff g2_from_jacobian_std(const uint64_t* w) {
  Fp2 X{fp_from_std(w), fp_from_std(w + 4)}, Y{fp_from_std(w + 8), fp_from_std(w + 12)}, Z{fp_from_std(w + 16), fp_from_std(w + 20)};
  if (f2_is_zero(Z)) return G2Aff{X, Y, true};
  if (f2_eq(Z, f2_one())) return G2Aff{X, Y, false};
  Fp2 zi = f2_inv(Z), zi2 = f2_sqr(zi);
  return G2Aff{f2_mul(X, zi2), f2_mul(Y, f2_mul(zi2, zi)), false};
}

// small complete Jacobian arithmetic on G1 for the handful of additions the verifiers do (IC accumulation)
struct G1Jac { Fp x, y, z; };
inline G1Jac g1j_inf() { return G1Jac{fpc().one, fpc().one, Fp{{0, 0, 0, 0}}}; }
inline G1Jac g1j_from(const G1Aff& p) { return p.inf ? g1j_inf() : G1Jac{p.x, p.y, fpc().one}; }
inline G1Jac g1j_dbl(const G1Jac& p) {
  if (fp_is_zero(p.z) || fp_is_zero(p.y)) return g1j_inf();
  Fp a = fp_sqr(p.x), b = fp_sqr(p.y), c = fp_sqr(b);
  Fp d = fp_dbl(fp_sub(fp_sub(fp_sqr(fp_add(p.x, b)), a), c));
  Fp e = fp_add(fp_dbl(a), a), f = fp_sqr(e);
  Fp x3 = fp_sub(f, fp_dbl(d));
  Fp c8 = fp_dbl(fp_dbl(fp_dbl(c)));
  Fp y3 = fp_sub(fp_mul(e, fp_sub(d, x3)), c8);
  Fp z3 = fp_dbl(fp_mul(p.y, p.z));
  return G1Jac{x3, y3, z3};
}
inline G1Jac g1j_add(const G1Jac& p, const G1Jac& q) {
  if (fp_is_zero(p.z)) return q;
  if (fp_is_zero(q.z)) return p;
  Fp z1z1 = fp_sqr(p.z), z2z2 = fp_sqr(q.z);
  Fp u1 = fp_mul(p.x, z2z2), u2 = fp_mul(q.x, z1z1);
  Fp s1 = fp_mul(p.y, fp_mul(q.z, z2z2)), s2 = fp_mul(q.y, fp_mul(p.z, z1z1));
  if (fp_eq(u1, u2)) return fp_eq(s1, s2) ? g1j_dbl(p) : g1j_inf();
  Fp h = fp_sub(u2, u1), r = fp_sub(s2, s1);
  Fp hh = fp_sqr(h), hhh = fp_mul(h, hh), v = fp_mul(u1, hh);
  Fp x3 = fp_sub(fp_sub(fp_sqr(r), hhh), fp_dbl(v));
  Fp y3 = fp_sub(fp_mul(r, fp_sub(v, x3)), fp_mul(s1, hhh));
  Fp z3 = fp_mul(fp_mul(p.z, q.z), h);
  return G1Jac{x3, y3, z3};
}
inline G1Jac g1j_mul(const G1Jac& p, const uint64_t k[4]) {   // any 256-bit k (no reduction needed: the group law does it)
  G1Jac r = g1j_inf();
  for (int i = 255; i >= 0; --i) {
    r = g1j_dbl(r);
    if ((k[i / 64] >> (i % 64)) & 1) r = g1j_add(r, p);
  }
  return r;
}
inline G1Aff g1j_affine(const G1Jac& p) {
  if (fp_is_zero(p.z)) return G1Aff{p.x, p.y, true};
  Fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
  return G1Aff{fp_mul(p.x, zi2), fp_mul(p.y, fp_mul(zi2, zi)), false};
}
inline G1Aff g1_neg(const G1Aff& p) { return G1Aff{p.x, fp_neg(p.y), p.inf}; }

// G2 membership: E'(Fq2) has a large cofactor (2q - r), so a point on the twist need not have order r, and a pairing
// "check" on such a point proves nothing.  On the order-r subgroup the twisted Frobenius psi acts as multiplication by
// q = t - 1 = 6 x^2 (mod r); psi(Q) == [6 x^2] Q singles that subgroup out (127-bit scalar instead of [r] Q).
struct G2Jac { Fp2 x, y, z; };
inline G2Jac g2j_dbl(const G2Jac& p) {
  if (f2_is_zero(p.z) || f2_is_zero(p.y)) return G2Jac{f2_one(), f2_one(), f2_zero()};
  Fp2 a = f2_sqr(p.x), b = f2_sqr(p.y), c = f2_sqr(b);
  Fp2 d = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, b)), a), c));
  Fp2 e = f2_add(f2_dbl(a), a), f = f2_sqr(e);
  Fp2 x3 = f2_sub(f, f2_dbl(d));
  Fp2 y3 = f2_sub(f2_mul(e, f2_sub(d, x3)), f2_dbl(f2_dbl(f2_dbl(c))));
  return G2Jac{x3, y3, f2_dbl(f2_mul(p.y, p.z))};
}
inline G2Jac g2j_add_affine(const G2Jac& p, const G2Aff& q) {       // q finite
  if (f2_is_zero(p.z)) return G2Jac{q.x, q.y, f2_one()};
  Fp2 z1z1 = f2_sqr(p.z);
  Fp2 u2 = f2_mul(q.x, z1z1), s2 = f2_mul(q.y, f2_mul(p.z, z1z1));
  if (f2_eq(p.x, u2)) return f2_eq(p.y, s2) ? g2j_dbl(p) : G2Jac{f2_one(), f2_one(), f2_zero()};
  Fp2 h = f2_sub(u2, p.x), r = f2_sub(s2, p.y);
  Fp2 hh = f2_sqr(h), hhh = f2_mul(h, hh), v = f2_mul(p.x, hh);
  Fp2 x3 = f2_sub(f2_sub(f2_sqr(r), hhh), f2_dbl(v));
  Fp2 y3 = f2_sub(f2_mul(r, f2_sub(v, x3)), f2_mul(p.y, hhh));
  return G2Jac{x3, y3, f2_mul(p.z, h)};
}
inline bool g2_in_subgroup(const G2Aff& q) {
  if (q.inf) return true;
  const u128 k = (u128)6 * kX * kX;                                  // 6 x^2 < 2^128
  G2Jac acc{f2_one(), f2_one(), f2_zero()};
  for (int i = 127; i >= 0; --i) {
    acc = g2j_dbl(acc);
    if ((k >> i) & 1) acc = g2j_add_affine(acc, q);
  }
  if (f2_is_zero(acc.z)) return false;
  const TowerConsts& t = twc();
  const Fp2 px = f2_mul(f2_conj(q.x), t.g1[2]), py = f2_mul(f2_conj(q.y), t.g1[3]);   // psi(Q)
  const Fp2 zz = f2_sqr(acc.z);
  return f2_eq(f2_mul(px, zz), acc.x) && f2_eq(f2_mul(py, f2_mul(zz, acc.z)), acc.y);
}

// ------------------------------------------------------------------------------------------------ multi-Miller loop
// prod_i f_{6x+2, Q_i}(P_i) * (the two Frobenius lines), pairs with an infinite member contribute 1.
// Returns false when a degenerate step shows a Q_i is not a point of order r (vertical line inside the loop).
inline bool multi_miller_loop(const std::vector<G1Aff>& ps, const std::vector<G2Aff>& qs, Fp12& out) {
  struct Pair { Fp xp, yp; G2Aff q; Fp2 tx, ty; };
  std::vector<Pair> pr;
  for (size_t i = 0; i < ps.size(); ++i) {
    if (ps[i].inf || qs[i].inf) continue;
    pr.push_back(Pair{ps[i].x, ps[i].y, qs[i], qs[i].x, qs[i].y});
  }
  Fp12 f = f12_one();
  const size_t k = pr.size();
  std::vector<Fp2> den(k);
  // one chord/tangent step for every pair: den <- 1/denominators (batched), then lines and point updates
  auto step = [&](bool dbl, const std::vector<G2Aff>* other) -> bool {
    for (size_t i = 0; i < k; ++i) {
      den[i] = dbl ? f2_dbl(pr[i].ty) : f2_sub((*other)[i].x, pr[i].tx);
      if (f2_is_zero(den[i])) return false;
    }
    f2_batch_inv(den);
    for (size_t i = 0; i < k; ++i) {
      Pair& a = pr[i];
      Fp2 lam, x3;
      if (dbl) {
        Fp2 xx = f2_sqr(a.tx);
        lam = f2_mul(f2_add(f2_dbl(xx), xx), den[i]);
        x3 = f2_sub(f2_sqr(lam), f2_dbl(a.tx));
      } else {
        lam = f2_mul(f2_sub((*other)[i].y, a.ty), den[i]);
        x3 = f2_sub(f2_sub(f2_sqr(lam), a.tx), (*other)[i].x);
      }
      // l(P) = y_P - lambda x_P w + (lambda x_T - y_T) w^3
      f = f12_mul_line(f, a.yp, f2_neg(f2_mul_fp(lam, a.xp)), f2_sub(f2_mul(lam, a.tx), a.ty));
      Fp2 y3 = f2_sub(f2_mul(lam, f2_sub(a.tx, x3)), a.ty);
      a.tx = x3;
      a.ty = y3;
    }
    return true;
  };
  std::vector<G2Aff> qv(k);
  for (size_t i = 0; i < k; ++i) qv[i] = pr[i].q;
  for (int b = 63; b >= 0; --b) {                    // bit 64 of 6x+2 is the leading one
    f = f12_sqr(f);
    if (!step(true, nullptr)) return false;
    if ((kLoop[0] >> b) & 1)
      if (!step(false, &qv)) return false;
  }
  // Q1 = pi(Q), -Q2 = -pi^2(Q) in twist coordinates: (conj(x) g1^2, conj(y) g1^3), (x g2^2, -y g2^3)
  const TowerConsts& t = twc();
  std::vector<G2Aff> q1(k), q2n(k);
  for (size_t i = 0; i < k; ++i) {
    q1[i] = G2Aff{f2_mul(f2_conj(pr[i].q.x), t.g1[2]), f2_mul(f2_conj(pr[i].q.y), t.g1[3]), false};
    q2n[i] = G2Aff{f2_mul(pr[i].q.x, t.g2[2]), f2_neg(f2_mul(pr[i].q.y, t.g2[3])), false};
  }
  if (!step(false, &q1)) return false;
  if (!step(false, &q2n)) return false;
  out = f;
  return true;
}

}  // namespace pairing
}  // namespace gs
