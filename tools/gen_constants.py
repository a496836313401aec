#!/usr/bin/env python3
"""Generates go-snark-study_amd/csrc/bn128_constants.h: BN128 moduli (bn128/bn128.go:40-50 in the
reference) in the 9 x 29-bit limb Montgomery representation (R = 2^261) used by the kernels,
plus generator points and NTT roots.  Run: python3 tools/gen_constants.py"""
import os

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617
W, N = 29, 9
MASK = (1 << W) - 1
RBITS = W * N
MONT_R = 1 << RBITS


def limbs(x):
    out = []
    for _ in range(N):
        out.append(x & MASK)
        x >>= W
    assert x == 0
    return out


def words8(x):
    return [(x >> (32 * i)) & 0xffffffff for i in range(8)]


def arr(name, vals, n=N):
    body = " ".join("case %d: return 0x%08xu;" % (i, v) for i, v in enumerate(vals))
    return ("  static __host__ __device__ __forceinline__ constexpr uint32_t %s(int i) {\n"
            "    switch (i) { %s default: return 0u; }\n  }\n" % (name, body))


def bias_limbs(kp):
    """K*p written with every low limb >= 2^30 so that limb-wise (a + bias - b) never underflows
    for nearly-normal b (limbs < 2^29 + 2^4) with value(b) <= (K-1) p."""
    l = limbs(kp)
    d = [l[0] + (1 << 30)] + [l[i] + (1 << 30) - 2 for i in range(1, N - 1)] + [l[N - 1] - 2]
    assert sum(v << (W * i) for i, v in enumerate(d)) == kp
    assert all(0 <= v < (1 << 32) for v in d) and d[-1] >= 0
    return d


def tight_bias_limbs(kp):
    """K*p with every low limb in [2^29 + 15, 2^30): limb-wise (a + bias - b) never underflows for nearly-normal b (limbs < 2^29 + 16)
    and stays below 3 * 2^29 + 16 -- the bias of the LAZY subtraction (fp29.h, sub_lazy / neg_lazy), whose result skips the carry
    pass and goes straight into a product.  Needs every low limb of K*p >= 16 (None otherwise: the caller takes the next K)."""
    l = limbs(kp)
    if any(v < 16 for v in l[:N - 1]) or l[N - 1] < 1:
        return None
    d = [l[0] + (1 << 29)] + [l[i] + (1 << 29) - 1 for i in range(1, N - 1)] + [l[N - 1] - 1]
    assert sum(v << (W * i) for i, v in enumerate(d)) == kp
    assert all((1 << 29) + 15 <= v < (1 << 30) for v in d[:N - 1])
    return d


def wide_bias_limbs(kp):
    """K*p with every low limb >= 2^31 - 4: a + bias - b - 2c never underflows for nearly-normal b, c (fp29.h, sub_b_2c)."""
    l = limbs(kp)
    d = [l[0] + (1 << 31)] + [l[i] + (1 << 31) - 4 for i in range(1, N - 1)] + [l[N - 1] - 4]
    assert sum(v << (W * i) for i, v in enumerate(d)) == kp
    assert all(0 <= v < (1 << 31) + (1 << 29) for v in d) and d[-1] >= 0
    return d


def field(name, p, extra=""):
    pinv = (-pow(p, -1, 1 << W)) % (1 << W)
    s = "struct %s {\n" % name
    s += "  static constexpr int kBits = %d;\n" % p.bit_length()
    s += "  static constexpr uint32_t kPinv29 = 0x%08xu;   // -p^-1 mod 2^29\n" % pinv
    s += "  static constexpr uint32_t kTopLimb = 0x%08xu;  // p >> 232\n" % (p >> (W * (N - 1)))
    s += arr("p", limbs(p))
    s += arr("p32", words8(p), 8)
    s += arr("one", limbs(MONT_R % p))            # 1 in Montgomery form
    s += arr("r2", limbs(MONT_R * MONT_R % p))     # to-Montgomery multiplier
    # bias tables for sub<K>: K in 1..MAXK
    maxk = 40
    s += "  static constexpr int kMaxBiasK = %d;\n" % maxk
    s += "  static __host__ __device__ __forceinline__ constexpr uint32_t bias(int k, int i) {\n    switch (k * 16 + i) {\n"
    for k in range(1, maxk + 1):
        d = bias_limbs(k * p)
        s += "      " + " ".join("case %d: return 0x%08xu;" % (k * 16 + i, v) for i, v in enumerate(d)) + "\n"
    s += "      default: return 0u;\n    }\n  }\n"
    # tight bias (lazy subtraction): entry k holds K' p for the smallest K' >= k whose low limbs are all >= 16
    tk = []
    s += "  static __host__ __device__ __forceinline__ constexpr uint32_t tbias(int k, int i) {\n    switch (k * 16 + i) {\n"
    for k in range(1, maxk + 1):
        kk = k
        while tight_bias_limbs(kk * p) is None:
            kk += 1
        tk.append(kk)
        d = tight_bias_limbs(kk * p)
        s += "      " + " ".join("case %d: return 0x%08xu;" % (k * 16 + i, v) for i, v in enumerate(d)) + "\n"
    s += "      default: return 0u;\n    }\n  }\n"
    s += "  static __host__ __device__ __forceinline__ constexpr int tbias_k(int k) {      // the multiple of p entry k of tbias holds\n    switch (k) { "
    s += " ".join("case %d: return %d;" % (k, kk) for k, kk in zip(range(1, maxk + 1), tk)) + " default: return 0; }\n  }\n"
    s += "  static __host__ __device__ __forceinline__ constexpr uint32_t wbias(int k, int i) {\n    switch (k * 16 + i) {\n"
    for k in range(1, maxk + 1):
        d = wide_bias_limbs(k * p)
        s += "      " + " ".join("case %d: return 0x%08xu;" % (k * 16 + i, v) for i, v in enumerate(d)) + "\n"
    s += "      default: return 0u;\n    }\n  }\n"
    s += extra
    s += "};\n\n"
    return s


def pairing_constants():
    """The derived constants of csrc/pairing.h (host verifier) and the identity its final exponentiation rests on:
    the BN addition chain over x computes EXACTLY (q^4 - q^2 + 1)/r, so easy part x hard part = (q^12 - 1)/r = the
    reference's FinalExp (bn128/bn128.go:169)."""
    x = 4965661367192848881
    assert Q == 36 * x ** 4 + 36 * x ** 3 + 24 * x ** 2 + 6 * x + 1 and R_ORDER == 36 * x ** 4 + 36 * x ** 3 + 18 * x ** 2 + 6 * x + 1
    hard = (Q ** 4 - Q ** 2 + 1) // R_ORDER
    assert (Q ** 4 - Q ** 2 + 1) % R_ORDER == 0
    chain = ((Q + Q ** 2 + Q ** 3) + 2 * (-1) + 6 * (x * x * Q * Q) + 12 * (-x * Q) + 18 * (-(x + x * x * Q)) + 30 * (-x * x)
             + 36 * (-(x ** 3 + x ** 3 * Q)))
    assert chain == hard, "the y0..y6 chain of pairing.h must be the exact hard part"
    assert (Q ** 6 - 1) * (Q ** 2 + 1) * hard == (Q ** 12 - 1) // R_ORDER
    assert (Q - 1) % 6 == 0
    w64 = lambda v, n: [(v >> (64 * i)) & (2 ** 64 - 1) for i in range(n)]   # noqa: E731
    return {"kP": w64(Q, 4), "kX": [x], "kLoop": w64(6 * x + 2, 2), "kPm1Div6": w64((Q - 1) // 6, 4)}


def main():
    out = "// GENERATED by tools/gen_constants.py -- do not edit.\n"
    out += "// BN128 constants (reference: bn128/bn128.go:40-83) in 9 x 29-bit limbs, Montgomery R = 2^261.\n"
    out += "#pragma once\n#include <stdint.h>\n#include <hip/hip_runtime.h>\n\nnamespace gs {\n\n"
    out += field("ModQ", Q)
    # Fr: NTT roots.  2-adicity 28, generator 5
    omega = pow(5, (R_ORDER - 1) >> 28, R_ORDER)
    assert pow(omega, 1 << 28, R_ORDER) == 1 and pow(omega, 1 << 27, R_ORDER) != 1
    extra = "  static constexpr int kTwoAdicity = 28;\n"
    extra += arr("omega28_mont", limbs(omega * MONT_R % R_ORDER))
    extra += arr("omega28_inv_mont", limbs(pow(omega, -1, R_ORDER) * MONT_R % R_ORDER))
    out += field("ModR", R_ORDER, extra)
    # generators (affine, Montgomery form)
    g2x = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634)
    g2y = (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531)
    out += "struct Gen {\n"
    out += arr("g1x", limbs(1 * MONT_R % Q)) + arr("g1y", limbs(2 * MONT_R % Q))
    out += arr("g2x0", limbs(g2x[0] * MONT_R % Q)) + arr("g2x1", limbs(g2x[1] * MONT_R % Q))
    out += arr("g2y0", limbs(g2y[0] * MONT_R % Q)) + arr("g2y1", limbs(g2y[1] * MONT_R % Q))
    # curve coefficients (Montgomery form): E: y^2 = x^3 + 3; twist E': y^2 = x^3 + 3/(9 + u)   (bn128.go:131-136)
    inv_norm = pow(9 * 9 + 1, -1, Q)                       # 1/(9 + u) = (9 - u)/82
    tb0, tb1 = 3 * 9 * inv_norm % Q, (-3 * inv_norm) % Q
    assert ((tb0 * 9 - tb1) % Q, (tb0 + 9 * tb1) % Q) == (3, 0)
    assert (g2y[0] ** 2 - g2y[1] ** 2) % Q == (g2x[0] ** 3 - 3 * g2x[0] * g2x[1] ** 2 + tb0) % Q     # the generator is on the twist
    out += arr("b1", limbs(3 * MONT_R % Q)) + arr("b2c0", limbs(tb0 * MONT_R % Q)) + arr("b2c1", limbs(tb1 * MONT_R % Q))
    out += "};\n\n}  // namespace gs\n"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "go-snark-study_amd", "csrc", "bn128_constants.h")
    with open(path, "w") as f:
        f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    for _k, _v in pairing_constants().items():
        print("pairing.h %s = {%s}" % (_k, ", ".join("0x%016xULL" % w for w in _v)))
    main()
