// Packed HBM formats of points / scalars and their (un)packing, shared by kernels and host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ec.h"

namespace gs {

constexpr uint32_t kSignBit = 0x80000000u;      // bucket entries: sign | window (5 bits) | term index (26 bits)
constexpr int kWindowShift = 26;
constexpr uint32_t kIndexMask = (1u << kWindowShift) - 1u;

// ---- packed memory formats ---------------------------------------------------------------------
// canonical Montgomery coordinate <-> 8 words
GS_HD Fe<ModQ, 1> load_fq(const uint32_t* p) {
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = p[i];
  const Fe<ModQ, 6> u = unpack32<ModQ>(w);
  Fe<ModQ, 1> r;                                  // stored values are canonical by construction
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = u.l[i];
  return r;
}
GS_HD void store_fq(uint32_t* p, const Fe<ModQ, 1>& a) {
  uint32_t w[8];
  pack32<ModQ>(a, w);
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = w[i];
}

template <class T> struct PointIO;
template <> struct PointIO<FqTag> {
  static constexpr int kAffineWords = 16;      // x[8] y[8]
  static constexpr int kXyzzWords = 4 * NL;    // x y zz zzz, raw limbs
  static GS_HD Affine<FqTag> load_affine(const uint32_t* p) { return {load_fq(p), load_fq(p + 8)}; }
  static GS_HD void store_affine(uint32_t* p, const Affine<FqTag>& a) { store_fq(p, a.x); store_fq(p + 8, a.y); }
  template <int B> static GS_HD void load_limbs(const uint32_t* p, Fe<ModQ, B>& e) {
#pragma unroll
    for (int i = 0; i < NL; ++i) e.l[i] = p[i];
  }
  template <int B> static GS_HD void store_limbs(uint32_t* p, const Fe<ModQ, B>& e) {
#pragma unroll
    for (int i = 0; i < NL; ++i) p[i] = e.l[i];
  }
  // standard-form words -> Montgomery element
  static GS_HD Fe<ModQ, 2> load_std(const uint32_t* p) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = p[i];
    return to_mont(unpack32<ModQ>(w));
  }
  static GS_HD void store_std(uint32_t* p, const Fe<ModQ, 1>& mont_canon) {
    uint32_t w[8];
    pack32<ModQ>(from_mont(mont_canon), w);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = w[i];
  }
  static constexpr int kCoordWords = 8;
};
template <> struct PointIO<Fq2Tag> {
  static constexpr int kAffineWords = 32;      // x0 x1 y0 y1
  static constexpr int kXyzzWords = 8 * NL;
  static GS_HD Affine<Fq2Tag> load_affine(const uint32_t* p) {
    return {{load_fq(p), load_fq(p + 8)}, {load_fq(p + 16), load_fq(p + 24)}};
  }
  static GS_HD void store_affine(uint32_t* p, const Affine<Fq2Tag>& a) {
    store_fq(p, a.x.c0); store_fq(p + 8, a.x.c1); store_fq(p + 16, a.y.c0); store_fq(p + 24, a.y.c1);
  }
  template <int B> static GS_HD void load_limbs(const uint32_t* p, Fq2e<B>& e) {
    PointIO<FqTag>::load_limbs(p, e.c0); PointIO<FqTag>::load_limbs(p + NL, e.c1);
  }
  template <int B> static GS_HD void store_limbs(uint32_t* p, const Fq2e<B>& e) {
    PointIO<FqTag>::store_limbs(p, e.c0); PointIO<FqTag>::store_limbs(p + NL, e.c1);
  }
  static GS_HD Fq2e<2> load_std(const uint32_t* p) { return {PointIO<FqTag>::load_std(p), PointIO<FqTag>::load_std(p + 8)}; }
  static GS_HD void store_std(uint32_t* p, const Fq2e<1>& m) {
    PointIO<FqTag>::store_std(p, m.c0); PointIO<FqTag>::store_std(p + 8, m.c1);
  }
  static constexpr int kCoordWords = 16;
};

// packed affine point as raw 16-byte words (software-pipelined gathers: load now, unpack after the wait)
template <class T> struct RawAffine { uint4 q[PointIO<T>::kAffineWords / 4]; };
template <class T>
GS_HD RawAffine<T> load_raw_affine(const uint32_t* p) {
  RawAffine<T> r;
  const uint4* s4 = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < PointIO<T>::kAffineWords / 4; ++i) r.q[i] = s4[i];
  return r;
}
template <class T>
GS_HD Affine<T> unpack_affine(const RawAffine<T>& r) {
  uint32_t w[PointIO<T>::kAffineWords];
#pragma unroll
  for (int i = 0; i < PointIO<T>::kAffineWords / 4; ++i) { w[4 * i] = r.q[i].x; w[4 * i + 1] = r.q[i].y; w[4 * i + 2] = r.q[i].z; w[4 * i + 3] = r.q[i].w; }
  return PointIO<T>::load_affine(w);
}

template <class T>
GS_HD Xyzz<T> load_xyzz(const uint32_t* p) {
  constexpr int cw = PointIO<T>::kXyzzWords / 4;
  Xyzz<T> r;
  PointIO<T>::load_limbs(p, r.x); PointIO<T>::load_limbs(p + cw, r.y);
  PointIO<T>::load_limbs(p + 2 * cw, r.zz); PointIO<T>::load_limbs(p + 3 * cw, r.zzz);
  return r;
}
template <class T>
GS_HD void store_xyzz(uint32_t* p, const Xyzz<T>& a) {
  constexpr int cw = PointIO<T>::kXyzzWords / 4;
  PointIO<T>::store_limbs(p, a.x); PointIO<T>::store_limbs(p + cw, a.y);
  PointIO<T>::store_limbs(p + 2 * cw, a.zz); PointIO<T>::store_limbs(p + 3 * cw, a.zzz);
}

// ---- scalar handling ------------------------------------------------------------------------------
// reduce a 256-bit scalar below r (inputs may be any 256-bit value; 2^256 < 6 r)
GS_HD void scalar_canon(uint32_t (&k)[8]) {
  for (int it = 0; it < 5; ++it) {
    uint32_t t[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t d = (uint64_t)k[i] - ModR::p32(i) - borrow;
      t[i] = (uint32_t)d;
      borrow = (d >> 32) & 1u;
    }
    const bool ge = borrow == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = ge ? t[i] : k[i];
  }
}

GS_HD uint32_t scalar_bits(const uint32_t (&k)[8], int pos, int c) {   // bits [pos, pos+c), c <= 24
  const int wi = pos >> 5, sh = pos & 31;
  uint64_t v = (wi < 8) ? k[wi] : 0u;
  if (wi + 1 < 8) v |= (uint64_t)k[wi + 1] << 32;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

}  // namespace gs
