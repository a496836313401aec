// Host-side unit-test driver for go-snark-study_amd/csrc/fp29.h.  The SAME __host__ __device__
// functions the kernels use are instantiated for the CPU here (compiled by hipcc, run without a
// GPU) and driven from pytest (tests/test_field_host.py) against Python big integers.
// Protocol: one request per line on stdin:  <field q|r> <op> <hex a> <hex b> <hex c> <hex d>
// -> one line on stdout with the hex result (standard, non-Montgomery form).
#include <cstdio>
#include <cstring>
#include <string>
#include <iostream>
#include <sstream>
#include "../../go-snark-study_amd/csrc/fp29.h"

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
static std::string to_hex(const uint32_t (&w)[8]) {
  char buf[80];
  snprintf(buf, sizeof buf, "%08x%08x%08x%08x%08x%08x%08x%08x", w[7], w[6], w[5], w[4], w[3], w[2], w[1], w[0]);
  return buf;
}

template <class M>
static std::string run(const std::string& op, const std::string& ha, const std::string& hb,
                       const std::string& hc, const std::string& hd) {
  uint32_t wa[8], wb[8], wc[8], wd[8], wo[8];
  parse_hex(ha, wa); parse_hex(hb, wb); parse_hex(hc, wc); parse_hex(hd, wd);
  Fe<M, 2> a = to_mont(unpack32<M>(wa)), b = to_mont(unpack32<M>(wb));
  Fe<M, 2> c = to_mont(unpack32<M>(wc)), d = to_mont(unpack32<M>(wd));
  Fe<M, 1> out;
  if (op == "mul") out = from_mont(mul(a, b));
  else if (op == "sqr") out = from_mont(sqr(a));
  else if (op == "add") out = from_mont(add(a, b));
  else if (op == "sub") out = from_mont(sub(a, b));
  else if (op == "neg") out = from_mont(neg(a));
  else if (op == "subr") {       // rippling difference: same value as sub, limbs 0..7 below 2^29 exactly; reduce2 without its carry pass on top
    auto t = sub_ripple(add(add(a, b), c), d);
    bool normal = true;
    for (int i = 0; i < NL - 1; ++i) normal = normal && t.l[i] < (1u << LB);
    if (!normal) return std::string("limbs");
    out = from_mont(reduce2_normal(t));
  }
  else if (op == "dbl") out = from_mont(dbl(a));
  else if (op == "inv") out = from_mont(inv(a));
  else if (op == "muladd") out = from_mont(mul_add(a, b, c, d));
  else if (op == "roundtrip") out = from_mont(a);
  else if (op == "reduce2") out = from_mont(reduce2(add(add(a, b), c)));       // bound 6 -> 2
  else if (op == "iszero") {   // (a - b) == 0 ?  and is (a+b-c) zero?
    bool z1 = is_zero(sub(a, b)), z2 = is_zero(sub(add(a, b), c));
    return std::string(z1 ? "1" : "0") + (z2 ? "1" : "0");
  } else if (op == "iszero4") {   // 4a - b and 4a - b + c - d: multiples k p with k up to 9 when they vanish (the prefilter's two candidates)
    auto f = dbl(dbl(a));                                    // bound 8
    bool z1 = is_zero(sub(f, b)), z2 = is_zero(sub(add(sub(f, b), c), d)), m1 = maybe_zero(sub(f, b));
    return std::string(z1 ? "1" : "0") + (z2 ? "1" : "0") + (m1 || !z1 ? "1" : "0");
  } else if (op == "chain") {
    // ((a+b) - c) * ((a - b) + 2d)  with lazily bounded operands, squared, minus a*d
    auto t1 = sub(add(a, b), c);                 // bound 2+2+2+1 = 7
    auto t2 = add(sub(a, b), dbl(d));            // bound 5 + 4 = 9
    auto t3 = mul(t1, t2);                       // 63 <= 160
    auto t4 = sqr(add(t3, t3));                  // bound 4 -> 16
    out = from_mont(sub(t4, mul(a, d)));
  } else if (op == "lazy12") {
    // heavy lazy operand: 12p-bounded times 13p-bounded
    auto x = add(add(add(a, b), add(c, d)), add(a, c));    // 12
    auto y = sub(x, neg(relax<3>(b)));                     // 12 + 4 + 1 = 17 -> too big for mul with 12; use 9
    auto z = add(add(a, b), add(c, d));                    // 8
    (void)y;
    out = from_mont(mul(x, add(z, relax<5>(d))));           // 12 * 13 = 156 <= 160
  } else return "ERR";
  pack32<M>(out, wo);
  return to_hex(wo);
}

int main() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream ss(line);
    std::string f, op, a, b, c, d;
    ss >> f >> op >> a >> b >> c >> d;
    std::string r = (f == "q") ? run<ModQ>(op, a, b, c, d) : run<ModR>(op, a, b, c, d);
    std::cout << r << "\n";
  }
  return 0;
}
