"""CPU ORACLE (test infrastructure, NOT product code) -- Python-int restatement of the reference's
pairing and verifiers: fields/fq6.go, fields/fq12.go, bn128/bn128.go:104-421,
groth16/groth16.go:281-305, snark.go:292-368.

Only tests/ may import this.  It follows the reference's operation order (projective doubling /
mixed-addition steps with precomputed line coefficients, mulBy024, one Fq12.Exp by FinalExp), so
its Fq12 value is what `Bn.Pairing` returns; the product's verifier computes the same value by a
different route (affine steps, batched inversion, Frobenius-based final exponentiation).

Pinned: VerifyProof below returns what the reference's own compiled wasm verifier returned on the
recorded instances (tests/golden/wasm_*: accept for public input 35, reject for 34), see
tests/test_verifier.py.  One pairing takes ~1 s in CPython (the 2790-bit Fq12.Exp).

Derived constants are computed here from their definitions (TwistMulByQX = xi^((q-1)/3),
TwistMulByQY = xi^((q-1)/2), FinalExp = (q^12 - 1)/r) and were checked once against the decimal
literals of bn128.go:143-169."""
from oracle.ref_py import FQ, FQ2, G1, G2, G2_GEN, Q, R


class Fq6:
    """fields/fq6.go : Fq2[v]/(v^3 - NonResidue)"""

    def __init__(self, f, non_residue):
        self.F = f
        self.NonResidue = non_residue

    def Zero(self):
        return (self.F.Zero(), self.F.Zero(), self.F.Zero())

    def One(self):
        return (self.F.One(), self.F.Zero(), self.F.Zero())

    def mulByNonResidue(self, a):      # fq6.go:33-35
        return self.F.Mul(self.NonResidue, a)

    def Add(self, a, b):               # fq6.go:38-44
        return tuple(self.F.Add(x, y) for x, y in zip(a, b))

    def Sub(self, a, b):               # fq6.go:51-57
        return tuple(self.F.Sub(x, y) for x, y in zip(a, b))

    def Neg(self, a):                  # fq6.go:60-62
        return self.Sub(self.Zero(), a)

    def Mul(self, a, b):               # fq6.go:65-95
        F = self.F
        v0, v1, v2 = F.Mul(a[0], b[0]), F.Mul(a[1], b[1]), F.Mul(a[2], b[2])
        return (
            F.Add(v0, self.mulByNonResidue(F.Sub(F.Mul(F.Add(a[1], a[2]), F.Add(b[1], b[2])), F.Add(v1, v2)))),
            F.Add(F.Sub(F.Mul(F.Add(a[0], a[1]), F.Add(b[0], b[1])), F.Add(v0, v1)), self.mulByNonResidue(v2)),
            F.Add(F.Sub(F.Mul(F.Add(a[0], a[2]), F.Add(b[0], b[2])), F.Add(v0, v2)), v1),
        )

    def Inverse(self, a):              # fq6.go:116-140
        F = self.F
        t0, t1, t2 = F.Square(a[0]), F.Square(a[1]), F.Square(a[2])
        t3, t4, t5 = F.Mul(a[0], a[1]), F.Mul(a[0], a[2]), F.Mul(a[1], a[2])
        c0 = F.Sub(t0, self.mulByNonResidue(t5))
        c1 = F.Sub(self.mulByNonResidue(t2), t3)
        c2 = F.Sub(t1, t4)
        t6 = F.Inverse(F.Add(F.Mul(a[0], c0), self.mulByNonResidue(F.Add(F.Mul(a[2], c1), F.Mul(a[1], c2)))))
        return (F.Mul(t6, c0), F.Mul(t6, c1), F.Mul(t6, c2))

    def Square(self, a):               # fq6.go:148-173
        F = self.F
        s0 = F.Square(a[0])
        ab = F.Mul(a[0], a[1])
        s1 = F.Add(ab, ab)
        s2 = F.Square(F.Add(F.Sub(a[0], a[1]), a[2]))
        bc = F.Mul(a[1], a[2])
        s3 = F.Add(bc, bc)
        s4 = F.Square(a[2])
        return (F.Add(s0, self.mulByNonResidue(s3)), F.Add(s1, self.mulByNonResidue(s4)),
                F.Sub(F.Add(F.Add(s1, s2), s3), F.Add(s0, s4)))

    def Equal(self, a, b):             # fq6.go:182-184
        return all(self.F.Equal(x, y) for x, y in zip(a, b))


class Fq12:
    """fields/fq12.go : Fq6[w]/(w^2 - v)"""

    def __init__(self, f, fq2, non_residue):
        self.F, self.Fq2, self.NonResidue = f, fq2, non_residue

    def Zero(self):
        return (self.F.Zero(), self.F.Zero())

    def One(self):
        return (self.F.One(), self.F.Zero())

    def mulByNonResidue(self, a):      # fq12.go:37-43
        return (self.Fq2.Mul(self.NonResidue, a[2]), a[0], a[1])

    def Mul(self, a, b):               # fq12.go:72-84
        F = self.F
        v0, v1 = F.Mul(a[0], b[0]), F.Mul(a[1], b[1])
        return (F.Add(v0, self.mulByNonResidue(v1)), F.Sub(F.Mul(F.Add(a[0], a[1]), F.Add(b[0], b[1])), F.Add(v0, v1)))

    def Inverse(self, a):              # fq12.go:105-114
        F = self.F
        t2 = F.Sub(F.Square(a[0]), self.mulByNonResidue(F.Square(a[1])))
        t3 = F.Inverse(t2)
        return (F.Mul(a[0], t3), F.Neg(F.Mul(a[1], t3)))

    def Square(self, a):               # fq12.go:122-137
        F = self.F
        ab = F.Mul(a[0], a[1])
        return (F.Sub(F.Mul(F.Add(a[0], a[1]), F.Add(a[0], self.mulByNonResidue(a[1]))), F.Add(ab, self.mulByNonResidue(ab))),
                F.Add(ab, ab))

    def Exp(self, base, e):            # fq12.go:139-156 (LSB first)
        res, rem, exp = self.One(), e, base
        while rem:
            if rem & 1:
                res = self.Mul(res, exp)
            exp = self.Square(exp)
            rem >>= 1
        return res

    def Equal(self, a, b):             # fq12.go:163-165
        return self.F.Equal(a[0], b[0]) and self.F.Equal(a[1], b[1])


XI = (9, 1)                                                    # bn128.go:90-93 NonResidueFq6 = Twist
FQ6 = Fq6(FQ2, XI)
FQ12 = Fq12(FQ6, FQ2, XI)


def _fq2_pow(a, e):
    r = FQ2.One()
    while e:
        if e & 1:
            r = FQ2.Mul(r, a)
        a = FQ2.Square(a)
        e >>= 1
    return r


def _fq2_mul_scalar(p, e):             # fq2.go:78-96: double-and-add of p, e times = p * e
    return (FQ.Mul(p[0], e), FQ.Mul(p[1], e))


LOOP_COUNT = 29793968203157093288                              # bn128.go:122 (6x + 2)
TWO_INV = FQ.Inverse(2)                                        # :129
TWIST_COEF_B = _fq2_mul_scalar(FQ2.Inverse(XI), 3)             # :136
FROBENIUS_C11 = Q - 1                                          # :138
TWIST_MUL_BY_Q_X = _fq2_pow(XI, (Q - 1) // 3)                  # :143-154
TWIST_MUL_BY_Q_Y = _fq2_pow(XI, (Q - 1) // 2)                  # :156-167
FINAL_EXP = (Q ** 12 - 1) // R                                 # :169


def doublingStep(cur):                 # bn128.go:262-292
    F = FQ2
    x, y, z = cur
    a = _fq2_mul_scalar(F.Mul(x, y), TWO_INV)
    b = F.Square(y)
    c = F.Square(z)
    d = F.Add(c, F.Add(c, c))
    e = F.Mul(TWIST_COEF_B, d)
    f = F.Add(e, F.Add(e, e))
    g = _fq2_mul_scalar(F.Add(b, f), TWO_INV)
    h = F.Sub(F.Square(F.Add(y, z)), F.Add(b, c))
    i = F.Sub(e, b)
    j = F.Square(x)
    e_sqr = F.Square(e)
    nxt = (F.Mul(a, F.Sub(b, f)), F.Sub(F.Sub(F.Square(g), e_sqr), F.Add(e_sqr, e_sqr)), F.Mul(b, h))
    return (F.Mul(i, XI), F.Neg(h), F.Add(j, F.Add(j, j))), nxt      # (Ell0, EllVW, EllVV)


def mixedAdditionStep(base, cur):      # bn128.go:294-330
    F = FQ2
    x1, y1, z1 = cur
    x2, y2 = base[0], base[1]
    d = F.Sub(x1, F.Mul(x2, z1))
    e = F.Sub(y1, F.Mul(y2, z1))
    f = F.Square(d)
    g = F.Square(e)
    h = F.Mul(d, f)
    i = F.Mul(x1, f)
    j = F.Sub(F.Add(h, F.Mul(z1, g)), F.Add(i, i))
    nxt = (F.Mul(d, j), F.Sub(F.Mul(e, F.Sub(i, j)), F.Mul(h, y1)), F.Mul(z1, h))
    return (F.Mul(XI, F.Sub(F.Mul(e, x2), F.Mul(d, y2))), d, F.Neg(e)), nxt


def g2MulByQ(p):                       # bn128.go:331-351
    fm = lambda c: (c[0], FQ.Mul(c[1], FROBENIUS_C11))   # noqa: E731
    return (FQ2.Mul(TWIST_MUL_BY_Q_X, fm(p[0])), FQ2.Mul(TWIST_MUL_BY_Q_Y, fm(p[1])), fm(p[2]))


def _affine3(group, p):
    """G.Affine in the reference returns a triple [x, y, 1] (g2.go:183-200); ref_py.Affine returns (x, y) or None."""
    a = group.Affine(p)
    if a is None:
        raise ValueError("pairing of the point at infinity: the reference divides by zero here")
    return a


def preComputeG2(p):                   # bn128.go:213-260
    q = _affine3(G2, p)
    q3 = (q[0], q[1], FQ2.One())
    r = (q[0], q[1], FQ2.One())
    coeffs = []
    for i in range(LOOP_COUNT.bit_length() - 2, -1, -1):
        c, r = doublingStep(r)
        coeffs.append(c)
        if (LOOP_COUNT >> i) & 1:
            c, r = mixedAdditionStep(q3, r)
            coeffs.append(c)
    q1a = _affine3(G2, g2MulByQ(q3))
    q1 = (q1a[0], q1a[1], FQ2.One())
    q2a = _affine3(G2, g2MulByQ(q1))
    q2 = (q2a[0], FQ2.Neg(q2a[1]), FQ2.One())
    c, r = mixedAdditionStep(q1, r)
    coeffs.append(c)
    c, r = mixedAdditionStep(q2, r)
    coeffs.append(c)
    return coeffs


def mulBy024(a, ell0, ellVW, ellVV):   # bn128.go:401-416
    z = FQ2.Zero()
    return FQ12.Mul(a, ((ell0, z, ellVV), (z, ellVW, z)))


def MillerLoop(pre1, coeffs):          # bn128.go:353-399
    px, py = pre1
    f = FQ12.One()
    idx = 0

    def use(f, c):
        return mulBy024(f, c[0], _fq2_mul_scalar(c[1], py), _fq2_mul_scalar(c[2], px))
    for i in range(LOOP_COUNT.bit_length() - 2, -1, -1):
        f = FQ12.Square(f)
        f = use(f, coeffs[idx])
        idx += 1
        if (LOOP_COUNT >> i) & 1:
            f = use(f, coeffs[idx])
            idx += 1
    f = use(f, coeffs[idx])
    f = use(f, coeffs[idx + 1])
    return f


def Pairing(p1, p2):                   # bn128.go:179-186
    pre1 = _affine3(G1, p1)
    return FQ12.Exp(MillerLoop(pre1, preComputeG2(p2)), FINAL_EXP)


def _ic_sum(ic, publicSignals):        # groth16.go:283-286 / snark.go:330-333
    acc = ic[0]
    for i, s in enumerate(publicSignals):
        acc = G1.Add(acc, G1.MulScalar(ic[i + 1], s))
    return acc


def groth16_VerifyProof(vk, proof, publicSignals):
    """groth16.go:281-305; proof = (PiA, PiB, PiC)."""
    piA, piB, piC = proof
    icPubl = _ic_sum(vk.IC, publicSignals)
    lhs = Pairing(piA, piB)
    rhs = FQ12.Mul(Pairing(vk.G1_Alpha, vk.G2_Beta), FQ12.Mul(Pairing(icPubl, vk.G2_Gamma), Pairing(piC, vk.G2_Delta)))
    return FQ12.Equal(lhs, rhs)


def snark_VerifyProof(vk, proof, publicSignals):
    """snark.go:292-368; proof = dict PiA, PiAp, PiB, PiBp, PiC, PiCp, PiH, PiKp.  -> (ok, first failing check or 0)."""
    p = proof
    if not FQ12.Equal(Pairing(p["PiA"], vk.Vka), Pairing(p["PiAp"], G2_GEN)):
        return False, 1
    if not FQ12.Equal(Pairing(vk.Vkb, p["PiB"]), Pairing(p["PiBp"], G2_GEN)):
        return False, 2
    if not FQ12.Equal(Pairing(p["PiC"], vk.Vkc), Pairing(p["PiCp"], G2_GEN)):
        return False, 3
    vkxpia = G1.Add(_ic_sum(vk.IC, publicSignals), p["PiA"])
    if not FQ12.Equal(Pairing(vkxpia, p["PiB"]), FQ12.Mul(Pairing(p["PiH"], vk.Vkz), Pairing(p["PiC"], G2_GEN))):
        return False, 4
    piApiC = G1.Add(vkxpia, p["PiC"])
    left = FQ12.Mul(Pairing(piApiC, vk.G2Kbg), Pairing(vk.G1Kbg, p["PiB"]))
    if not FQ12.Equal(left, Pairing(p["PiKp"], vk.G2Kg)):
        return False, 5
    return True, 0
