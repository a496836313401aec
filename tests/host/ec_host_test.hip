// Host-side unit-test driver for go-snark-study_amd/csrc/ec.h (same __host__ __device__ source the
// kernels use), driven by tests/test_ec_host.py against oracle/ref_py.py in affine form.
// Input: first line "<g1|g2> <op> <n>", then n lines "X.. Y.. Z.. k sign" (hex; g1: 3 coords,
// g2: 6 coords c0 c1 per coordinate).  Output: affine coordinates hex or "inf".
//   op = lincomb : sum_i mul_words(P_i, k_i)  (xyzz_dbl / xyzz_add)
//   op = maddsum : sum_i (sign ? -P_i : P_i)  (xyzz_madd)
//   op = small   : sum_i mul_u32(P_i, k_i & 0xffffffff)
//   op = jacdbl  : sum_i 2^(k_i & 0xffff) P_i, every power by repeated jac_dbl (the window-table builder's doubling, round 6), one inversion each
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include "../../go-snark-study_amd/csrc/ec.h"

using namespace gs;

static void parse_hex(const std::string& s, uint32_t (&w)[8]) {
  memset(w, 0, sizeof(w));
  int n = (int)s.size();
  for (int i = 0; i < n; ++i) {
    char c = s[n - 1 - i];
    uint32_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
    if (i / 8 < 8) w[i / 8] |= v << (4 * (i % 8));
  }
}
static std::string hex_of(const Fe<ModQ, 1>& mont) {
  uint32_t w[8];
  pack32<ModQ>(from_mont(mont), w);
  char buf[80];
  snprintf(buf, sizeof buf, "%08x%08x%08x%08x%08x%08x%08x%08x", w[7], w[6], w[5], w[4], w[3], w[2], w[1], w[0]);
  return buf;
}
static Fe<ModQ, 2> rd(std::istringstream& ss) {
  std::string h; ss >> h;
  uint32_t w[8]; parse_hex(h, w);
  return to_mont(unpack32<ModQ>(w));
}
static void rd_elem(std::istringstream& ss, Fe<ModQ, 2>& e) { e = rd(ss); }
static void rd_elem(std::istringstream& ss, Fq2e<2>& e) { e.c0 = rd(ss); e.c1 = rd(ss); }
static std::string show(const Fe<ModQ, 1>& e) { return hex_of(e); }
static std::string show(const Fq2e<1>& e) { return hex_of(e.c0) + " " + hex_of(e.c1); }

// the accumulation kernel's own accumulator: G2 takes the tight one (ec.h XyzzAcc), G1 the plain Xyzz
template <class T> struct TightOf { using type = Xyzz<T>; static type inf() { return xyzz_inf<T>(); } static Xyzz<T> plain(const type& a) { return a; } };
template <> struct TightOf<Fq2Tag> {
  using type = XyzzAcc<Fq2Tag>;
  static type inf() { return xyzz_acc_inf<Fq2Tag>(); }
  static Xyzz<Fq2Tag> plain(const type& a) { return to_xyzz(a); }
};

template <class T>
static void run(const std::string& op, int n) {
  Xyzz<T> acc = xyzz_inf<T>();
  typename TightOf<T>::type tight = TightOf<T>::inf();
  for (int i = 0; i < n; ++i) {
    std::string line; std::getline(std::cin, line);
    std::istringstream ss(line);
    typename T::template E<2> X, Y, Z;
    rd_elem(ss, X); rd_elem(ss, Y); rd_elem(ss, Z);
    std::string kh; int sign; ss >> kh >> sign;
    uint32_t k[8]; parse_hex(kh, k);
    Affine<T> a = jacobian_to_affine<T>(X, Y, Z);
    if (op == "lincomb") {
      Xyzz<T> t = xyzz_mul_words_w4(xyzz_from_affine(a), k);
      if (sign) t = xyzz_neg(t);
      xyzz_add(acc, t);
    } else if (op == "maddsum") {
      xyzz_madd(acc, a, sign != 0);
    } else if (op == "maddacc") {          // the same chain through the kernel's accumulator type
      xyzz_madd(tight, a, sign != 0);
    } else if (op == "jacdbl") {
      if (!is_inf(a)) {
        Jac<T> j = jac_from_affine<T>(a);
        for (uint32_t d = 0; d < (k[0] & 0xffffu); ++d) jac_dbl(j);
        if (!is_inf(j)) xyzz_madd(acc, jac_to_affine_with_inverse<T>(j, inv(reduce2(j.z))), sign != 0);
      }
    } else if (op == "small") {
      const uint32_t small[8] = {k[0], 0, 0, 0, 0, 0, 0, 0};
      xyzz_add(acc, xyzz_mul_words_w4(xyzz_from_affine(a), small));
    }
  }
  if (op == "maddacc") acc = TightOf<T>::plain(tight);
  Affine<T> r = xyzz_to_affine(acc);
  if (is_inf(r)) std::cout << "inf\n";
  else std::cout << show(r.x) << " " << show(r.y) << "\n";
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream ss(line);
    std::string g, op; int n;
    ss >> g >> op >> n;
    if (g == "g1") run<FqTag>(op, n); else run<Fq2Tag>(op, n);
  }
  return 0;
}
