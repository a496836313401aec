"""-m gpu parity: Groth16 / Pinocchio provers and the Fr polynomial kernels through the C ABI vs
 * the reference's own compiled prover (tests/golden/wasm_*.json) -- affine normal form,
 * the oracles (ref_py / gs_oracle.c) on seeded random instances,
 * the reference's polynomial test vectors (r1csqap_test.go)."""
import random

import numpy as np
import pytest

import gosnark_amd
from gosnark_amd import capi, groth16, snark, r1csqap
import golden_util as GU
import gpu_util as U
from oracle import c_oracle as C
from oracle import ref_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True, params=["always", "never"])
def _init(request):
    """Every test of this module runs twice: on window tables (gs_set_table_policy `always`: built inside the first call that needs
    them, what rounds 1-4 did) and table-free (`never`: a bucket set per window, Horner recombination) -- the two routes of round 5
    must give the same proofs on every golden, closed form and edge case below.  (`auto`, the library's default, is a schedule of
    these two: tests/test_gpu_table_policy.py.)"""
    capi.init()
    capi.set_table_policy(request.param)
    yield request.param
    capi.set_table_policy("auto")


def jac_affine_g1(p):
    a = O.G1.Affine(p)
    return (0, 0, 0) if a is None else (a[0], a[1], 1)


def jac_affine_g2(p):
    a = O.G2.Affine(p)
    return ((0, 0), (0, 0), (0, 0)) if a is None else (a[0], a[1], (1, 0))


def mk_groth_pk(opk):
    return groth16.Pk(BACDelta=opk.BACDelta, Z=opk.Z, G1_Alpha=opk.G1_Alpha, G1_Beta=opk.G1_Beta, G1_Delta=opk.G1_Delta,
                      G1_At=opk.G1_At, G1_BACGamma=opk.G1_BACGamma, G2_Beta=opk.G2_Beta, G2_Delta=opk.G2_Delta,
                      G2_BACGamma=opk.G2_BACGamma, PowersTauDelta=opk.PowersTauDelta)


@pytest.mark.parametrize("name", ["groth_x3", "groth_rand_m9", "groth_rand_m17"])
def test_groth16_proof_equals_reference_wasm_affine(name):
    """Same pk, w, px, r, s as the reference's compiled prover -> identical affine proof elements."""
    rec = GU.load(name)
    opk = GU.groth_pk(rec["setup"])
    r, s = GU.rs_from_stream(rec["rand"])
    circ = groth16.Circuit(rec["circuit"]["NVars"], rec["circuit"]["NPublic"])
    proof = groth16.GenerateProofsWithRS(circ, mk_groth_pk(opk), rec["w"], rec["px"], r, s)
    assert proof.PiA == jac_affine_g1(GU.g1(rec["proof"]["PiA"]))
    assert proof.PiB == jac_affine_g2(GU.g2(rec["proof"]["PiB"]))
    assert proof.PiC == jac_affine_g1(GU.g1(rec["proof"]["PiC"]))


@pytest.mark.parametrize("name", ["pinocchio_x3_fixture", "pinocchio_rand_m9"])
def test_pinocchio_proof_equals_reference_wasm_affine(name):
    rec = GU.load(name)
    opk = GU.pinocchio_pk(rec["setup"])
    circ = snark.Circuit(rec["circuit"]["NVars"], rec["circuit"]["NPublic"])
    pk = snark.Pk(G1T=opk.G1T, A=opk.A, B=opk.B, C=opk.C, Kp=opk.Kp, Ap=opk.Ap, Bp=opk.Bp, Cp=opk.Cp, Z=opk.Z)
    proof = snark.GenerateProofs(circ, pk, rec["w"], rec["px"])
    for k in ("PiA", "PiAp", "PiBp", "PiC", "PiCp", "PiH", "PiKp"):
        assert getattr(proof, k) == jac_affine_g1(GU.g1(rec["proof"][k])), k
    assert proof.PiB == jac_affine_g2(GU.g2(rec["proof"]["PiB"]))


def test_groth16_random_instance_vs_python_oracle_m33():
    """Seeded instance beyond the goldens (m = 33, n = 32, inexact division)."""
    rng = random.Random(4242)
    m, n = 33, 32
    opk = O.GrothPk()
    z = [1]
    for i in range(1, m - 1):
        z = O.PF.Mul(z, [O.FR.Neg(i), 1])
    opk.Z = z
    opk.G1_Alpha, opk.G1_Beta, opk.G1_Delta = (U.rand_g1_jac(rng) for _ in range(3))
    opk.G2_Beta, opk.G2_Delta = U.rand_g2_jac(rng), U.rand_g2_jac(rng)
    opk.G1_At = [U.rand_g1_jac(rng, 0.1) for _ in range(m)]
    opk.G1_BACGamma = [U.rand_g1_jac(rng, 0.1) for _ in range(m)]
    opk.G2_BACGamma = [U.rand_g2_jac(rng, 0.1) for _ in range(m)]
    # BACDelta[0..NPublic] deliberately NOT infinity: the reference never reads them (groth16.go:248)
    opk.BACDelta = [U.rand_g1_jac(rng) for _ in range(m)]
    opk.PowersTauDelta = [U.rand_g1_jac(rng) for _ in range(len(z))]
    w = [1] + [rng.randrange(O.R) for _ in range(m - 1)]
    px = [rng.randrange(O.R) for _ in range(2 * n - 1)]
    r, s = rng.randrange(O.R), rng.randrange(O.R)
    want = O.groth16_GenerateProofs(m, 1, opk, w, px, r, s)
    got = groth16.GenerateProofsWithRS(groth16.Circuit(m, 1), mk_groth_pk(opk), w, px, r, s)
    assert got.PiA == jac_affine_g1(want[0])
    assert got.PiB == jac_affine_g2(want[1])
    assert got.PiC == jac_affine_g1(want[2])


def test_prover_rejects_shape_violations():
    rec = GU.load("groth_x3")
    opk = GU.groth_pk(rec["setup"])
    circ = groth16.Circuit(8, 1)
    pk = mk_groth_pk(opk)
    with pytest.raises(capi.GosnarkHipError):
        groth16.GenerateProofsWithRS(circ, pk, rec["w"][:-1], rec["px"], 1, 2)
    with pytest.raises(capi.GosnarkHipError):       # len(hx) would exceed len(PowersTauDelta)
        groth16.GenerateProofsWithRS(circ, pk, rec["w"], rec["px"] + [1, 2, 3], 1, 2)


# ---- polynomial field (r1csqap/r1csqap_test.go) -------------------------------------------------------
def test_poly_reference_vectors():
    PF = r1csqap.PolynomialField()
    a, b = [1, 0, 5], [3, 0, 1]
    assert PF.Mul(a, b) == [3, 0, 16, 0, 5]                       # r1csqap_test.go:59-62
    q, rem = PF.Div([3, 0, 16, 0, 5], b)                          # :64-67
    assert q == a and rem == [0, 0]
    assert PF.Add(a, b) == [4, 0, 6]                              # :69-71
    assert PF.Sub(a, b) == [O.R - 2, 0, 4]                        # :73-77
    assert PF.Eval([1, 2, 3], 5) == 86


@pytest.mark.parametrize("na,nb", [(1, 1), (2, 1), (13, 7), (64, 64), (257, 100), (1000, 999)])
def test_poly_mul_div_vs_c_oracle(na, nb):
    rng = random.Random(na * 1000 + nb)
    PF = r1csqap.PolynomialField()
    a = [rng.randrange(O.R) for _ in range(na)]
    b = [rng.randrange(O.R) for _ in range(nb)]
    b[-1] = b[-1] or 1
    assert PF.Mul(a, b) == C.poly_mul(a, b)
    q, rem = PF.Div(a, b)
    cq, cr = C.poly_div(a, b)
    assert q == cq and rem == cr
    x = rng.randrange(O.R)
    assert PF.Eval(a, x) == C.poly_eval(a, x)
    assert PF.Add(a, b) == O.PF.Add(a, b) and PF.Sub(b, a) == O.PF.Sub(b, a)


def test_poly_div_x3_px_by_z_exact():
    """groth16_test.go:78-86: px / Z has zero remainder, hx * Z == px, len(hx) = len(px) - len(Z) + 1."""
    rec = GU.load("pinocchio_x3_fixture")
    z = GU.pinocchio_pk(rec["setup"]).Z
    PF = r1csqap.PolynomialField()
    hx, rem = PF.Div(rec["px"], z)
    assert all(x == 0 for x in rem) and len(hx) == len(rec["px"]) - len(z) + 1
    assert PF.Mul(hx, z) == rec["px"]
    assert hx == O.PF.DivisorPolynomial(rec["px"], z)


def test_poly_quotient_large_roundtrip():
    """Size-independent property at 2^16: (q * b + r) / b == q with deg r < deg b."""
    n = 1 << 16
    q = U.rand_scalars_u64(n, 31)
    b = U.rand_scalars_u64(n, 32)
    lib = capi.load_library()
    prod = np.zeros((2 * n - 1, 4), dtype=np.uint64)
    capi.check(lib.gs_poly_mul(capi.ptr64(q), n, capi.ptr64(b), n, capi.ptr64(prod)))
    rsmall = U.rand_scalars_u64(n - 1, 33)
    tot = np.zeros((2 * n - 1, 4), dtype=np.uint64)
    capi.check(lib.gs_poly_add(capi.ptr64(prod), 2 * n - 1, capi.ptr64(rsmall), n - 1, capi.ptr64(tot)))
    q2 = np.zeros((n, 4), dtype=np.uint64)
    r2 = np.zeros((n - 1, 4), dtype=np.uint64)
    capi.check(lib.gs_poly_div(capi.ptr64(tot), 2 * n - 1, capi.ptr64(b), n, capi.ptr64(q2), capi.ptr64(r2)))
    assert np.array_equal(q2, q) and np.array_equal(r2, rsmall)
    # and a 2^12 slice against the C oracle's long division
    m = 1 << 12
    cq, cr = C.poly_div_u64(tot[:2 * m - 1], b[:m])
    q3 = np.zeros((m, 4), dtype=np.uint64)
    r3 = np.zeros((m - 1, 4), dtype=np.uint64)
    capi.check(lib.gs_poly_div(capi.ptr64(np.ascontiguousarray(tot[:2 * m - 1])), 2 * m - 1, capi.ptr64(np.ascontiguousarray(b[:m])), m,
                               capi.ptr64(q3), capi.ptr64(r3)))
    assert np.array_equal(q3, cq) and np.array_equal(r3, cr)


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 6, 7, 8, 9, 31, 100, 1000])
def test_zpoly_vs_reference_product(deg):
    """Z(x) = prod_{i=1}^{deg}(x - i) built as the reference does (groth16.go:122-131: repeated Mul by
    [-i, 1]) equals the device subproduct tree."""
    z = [1]
    for i in range(1, deg + 1):
        z = O.PF.Mul(z, [O.FR.Neg(i), 1])
    assert r1csqap.ZPoly(deg) == z


def test_zpoly_large_properties():
    """deg = 2^16 + 5: monic, Z(k) = 0 for sampled nodes, Z(0) = (-1)^deg deg!, Z(deg+1) = deg!."""
    deg = (1 << 16) + 5
    z = r1csqap.ZPoly(deg)
    PF = r1csqap.PolynomialField()
    assert len(z) == deg + 1 and z[-1] == 1
    for k in (1, 2, 77, deg // 2, deg - 1, deg):
        assert PF.Eval(z, k) == 0
    fact = 1
    for i in range(1, deg + 1):
        fact = fact * i % O.R
    assert z[0] == (fact if deg % 2 == 0 else O.R - fact)
    assert PF.Eval(z, deg + 1) == fact


def test_groth16_2p16_linearity_and_c_oracle_slices():
    """BASELINE config sizes cannot be replayed by the naive oracle, so check size-independent
    properties of the full prover at n = 2^16 (m = n + 1):
      * PiA - alpha - r delta, evaluated through two witnesses, is additive in w (MSM linearity),
      * with r = s = 0 and px = 0 the proof elements equal plain MSMs, which are checked against the
        C oracle's naive loop on a 2^10-term slice via the MSM entry point sharing the same bases."""
    from gosnark_amd import synth
    n = 1 << 16
    inst = synth.random_instance(n, 0xC0FFEE)
    pk = inst.device_pk()
    zero_px = capi.scalars_upload(np.zeros((2 * n - 1, 4), dtype=np.uint64))
    w1 = synth.scalars_u64(n + 1, 101)
    w2 = synth.scalars_u64(n + 1, 102)
    w12 = capi.ints_to_u64([(a + b) % O.R for a, b in zip(U.u64_rows_to_ints(w1), U.u64_rows_to_ints(w2))])
    proofs = [groth16.prove_resident(pk, capi.scalars_upload(w), zero_px, 0, 0) for w in (w1, w2, w12)]
    alpha = O.G1.Affine(inst.alpha)
    beta2 = O.G2.Affine(inst.beta2)

    def minus(G, p, q):          # affine p - q via the oracle's group law
        return G.Affine(G.Add((p[0], p[1], G.F.One()), G.Neg((q[0], q[1], G.F.One()))))

    def plus(G, p, q):
        return G.Affine(G.Add((p[0], p[1], G.F.One()), (q[0], q[1], G.F.One())))
    a = [minus(O.G1, (p.PiA[0], p.PiA[1]), alpha) for p in proofs]
    assert plus(O.G1, a[0], a[1]) == a[2]
    b = [minus(O.G2, (p.PiB[0], p.PiB[1]), beta2) for p in proofs]
    assert plus(O.G2, b[0], b[1]) == b[2]
    # MSM over At equals PiA - alpha; cross-check the engine against the naive loop on a slice
    full = capi.msm(inst.g1["at"], w1)
    assert full == a[0]
    k = 1 << 10
    pts = capi.g1_download(inst.g1["at"])[:k]
    want = C.g1_affine(C.g1_msm_naive(pts, w1[:k], threads=8))
    assert capi.msm(inst.g1["at"], w1[:k]) == want


# ---- interpolation / sparse R1CS -> P(x)  (r1csqap.go:129-210) ----------------------------------------------
def test_lagrange_reference_vectors():
    """r1csqap_test.go:107-112: interpolation through the nodes 1..n reproduces the values; and the reference's own
    (n <= 21, before its Go-int overflow) result from the C oracle."""
    PF = r1csqap.PolynomialField()
    for v in ([1], [5, 7], [0, 0, 0, 5], [3, 1, 4, 1, 5, 9, 2, 6]):
        coef = PF.LagrangeInterpolation(v)
        assert coef == C.lagrange(v)
        for j, val in enumerate(v):
            assert PF.Eval(coef, j + 1) == val % O.R


@pytest.mark.parametrize("n", [2, 3, 17, 21, 33, 100, 257])
def test_lagrange_vs_c_oracle(n):
    rng = random.Random(700 + n)
    v = [rng.randrange(O.R) for _ in range(n)]
    assert r1csqap.PolynomialField().LagrangeInterpolation(v) == C.lagrange(v)


def test_r1cs_to_qap_x3_circuit_matches_reference_flow():
    """circuit_test.go / r1csqap_test.go:114-174 flow on the x^3 + x + 5 R1CS: R1CSToQAP + CombinePolynomials (dense,
    reference semantics, python oracle) == device mirror == the sparse ComputePx path; px / Z is exact."""
    a = [[0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 1, 0, 0, 0], [5, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0]]
    b = [[0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0]]
    c = [[0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [1, 0, 0, 0, 0, 0, 0, 0]]
    w = list(O.X3_WITNESS)
    # the witness must satisfy the system for the division to be exact
    for ra, rb, rc in zip(a, b, c):
        dot = lambda row: sum(x * y for x, y in zip(row, w)) % O.R   # noqa: E731
        assert dot(ra) * dot(rb) % O.R == dot(rc)
    PF = r1csqap.PolynomialField()
    al, be, ga, z = PF.R1CSToQAP(a, b, c)
    oal, obe, oga, oz = O.PF.R1CSToQAP(a, b, c)
    assert (al, be, ga, z) == (oal, obe, oga, oz)
    ax, bx, cx, px = PF.CombinePolynomials(w, al, be, ga)
    assert (ax, bx, cx, px) == tuple(O.PF.CombinePolynomials(w, oal, obe, oga))
    rows = lambda mat: [{k: v for k, v in enumerate(row) if v} for row in mat]   # noqa: E731
    sa, sb, sc, spx = r1csqap.ComputePx(r1csqap.csr_from_rows(rows(a)), r1csqap.csr_from_rows(rows(b)), r1csqap.csr_from_rows(rows(c)),
                                        capi.ints_to_u64(w), len(w))
    assert U.u64_rows_to_ints(sa) == ax and U.u64_rows_to_ints(sb) == bx and U.u64_rows_to_ints(sc) == cx
    assert U.u64_rows_to_ints(spx) == px
    hx, rem = PF.Div(px, z)
    assert all(v == 0 for v in rem) and len(hx) == len(px) - len(z) + 1


@pytest.mark.parametrize("logn", [5, 12, 16])
def test_sqchain_px_is_exactly_divisible(logn):
    """The synthetic sqchain(n) circuit: px from the sparse system vanishes on all n nodes, i.e. px = hx * Z'
    with Z' = prod_{i=1}^{n}(x - i), and the interpolants reproduce (A w)_j at sampled nodes."""
    from gosnark_amd import synth
    n = 1 << logn
    a, b, c, w = synth.sqchain_r1cs(n, 123456789)
    ax, bx, cx, px = r1csqap.ComputePx(a, b, c, w, n + 1)
    wi = U.u64_rows_to_ints(w)
    PF = r1csqap.PolynomialField()
    axi = U.u64_rows_to_ints(ax)
    for j in (1, 2, n // 2, n - 1, n):
        want = wi[j] if j < n else 1                       # (A w)_j = s_j, last row: one
        assert PF.Eval(axi, j) == want
    zfull = capi.zpoly(n)                                  # all n nodes
    lib = capi.load_library()
    q = np.zeros((n - 1, 4), dtype=np.uint64)
    rem = np.zeros((n, 4), dtype=np.uint64)
    capi.check(lib.gs_poly_div(capi.ptr64(px), 2 * n - 1, capi.ptr64(zfull), n + 1, capi.ptr64(q), capi.ptr64(rem)))
    assert not rem.any()
    if logn <= 12:                                         # the C oracle's O(n^2) long division agrees on the quotient
        cq, cr = C.poly_div_u64(px, zfull)
        assert np.array_equal(cq, q) and not cr.any()


# ---- trusted setup on the device (groth16.go:94-222) ------------------------------------------------------------------
X3_A = [[0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 1, 0, 0, 0], [5, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0]]
X3_B = [[0, 0, 1, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0, 0]]
X3_C = [[0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 1, 0], [1, 0, 0, 0, 0, 0, 0, 0]]


def test_trusted_setup_equals_reference_on_x3_circuit():
    """gs_groth16_setup on the sparse x^3 + x + 5 R1CS == the reference's GenerateTrustedSetup (python oracle, dense QAP
    from R1CSToQAP) with the same toxic values: every proving-key array, the single points and the verification key."""
    rng = random.Random(2024)
    toxic = tuple(rng.randrange(1, O.R) for _ in range(5))
    al, be, ga, _ = O.PF.R1CSToQAP(X3_A, X3_B, X3_C)
    opk, ovk = O.groth16_GenerateTrustedSetup(8, 1, al, be, ga, toxic)
    rows = lambda mat: [{k: v for k, v in enumerate(row) if v} for row in mat]   # noqa: E731
    dpk, vk = groth16.GenerateTrustedSetupSparse(7, 8, 1, r1csqap.csr_from_rows(rows(X3_A)), r1csqap.csr_from_rows(rows(X3_B)),
                                                  r1csqap.csr_from_rows(rows(X3_C)), toxic)
    for name, ref in (("G1_At", opk.G1_At), ("G1_BACGamma", opk.G1_BACGamma), ("BACDelta", opk.BACDelta), ("PowersTauDelta", opk.PowersTauDelta)):
        assert groth16.ExportPkArray(dpk, name) == [jac_affine_g1(p) for p in ref], name
    assert groth16.ExportPkArray(dpk, "G2_BACGamma") == [jac_affine_g2(p) for p in opk.G2_BACGamma]
    assert vk.G1_Alpha == jac_affine_g1(ovk.G1_Alpha) and vk.G2_Beta == jac_affine_g2(ovk.G2_Beta)
    assert vk.G2_Gamma == jac_affine_g2(ovk.G2_Gamma) and vk.G2_Delta == jac_affine_g2(ovk.G2_Delta)
    assert vk.IC == [jac_affine_g1(p) for p in ovk.IC]
    # and a proof with that resident key equals the oracle prover on the oracle's key (alpha/beta/delta singles included)
    w = list(O.X3_WITNESS)
    _, _, _, px = O.PF.CombinePolynomials(w, al, be, ga)
    r, s = rng.randrange(O.R), rng.randrange(O.R)
    want = O.groth16_GenerateProofs(8, 1, opk, w, px, r, s)
    got = groth16.GenerateProofsWithRS(groth16.Circuit(8, 1), dpk, w, px, r, s)
    assert got.PiA == jac_affine_g1(want[0]) and got.PiB == jac_affine_g2(want[1]) and got.PiC == jac_affine_g1(want[2])


@pytest.mark.parametrize("n", [16, 1000, 1024, 3001, 1 << 16, (1 << 16) + 1, 1 << 20])
def test_full_pipeline_proof_matches_closed_form_from_toxic_values(n):
    """End to end at sizes the reference cannot replay (2^20 = BASELINE.json configs[2]): sparse R1CS -> device trusted
    setup -> px on the device -> prove.  With the toxic scalars known, groth16.go:243-275 must output PiA = a G1, PiB = b G2, PiC = c G1 for the
    closed-form (a, b, c) of synth.SqchainSetupInstance.expected_proof_scalars; the generator multiples come from the
    C oracle's MulScalar."""
    from gosnark_amd import synth
    inst = synth.sqchain_setup_instance(n, 0xBEEF00 + n % 251)
    r, s = synth.field_elems(2, 4040 + n % 251)
    proof = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    a, b, c = inst.expected_proof_scalars(r, s)
    wa = C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    wb = C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
    wc = C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    assert (proof.PiA[0], proof.PiA[1]) == wa
    assert (proof.PiB[0], proof.PiB[1]) == wb
    assert (proof.PiC[0], proof.PiC[1]) == wc
    # and the verifier (groth16.go:281-305) accepts it against the device-built vk for the right public input only
    x = capi.u64_to_ints(inst.w_host[1:2])[0]
    assert groth16.VerifyProof(inst.vk, proof, [x]) is True
    assert groth16.VerifyProof(inst.vk, proof, [(x + 1) % O.R]) is False
    assert groth16.VerifyProof(inst.vk, groth16.Proof(proof.PiC, proof.PiB, proof.PiA), [x]) is False


@pytest.mark.parametrize("shards", [1, 2, 3, 8])
def test_sharded_prove_equals_single_device_prove(shards):
    """SURVEY 8e on one device: the proof assembled from `shards` logical ranks (gs_groth16_prove_partials per shard, the
    library's host-side complete addition for the exchange step, gs_groth16_finish) equals gs_groth16_prove_resident."""
    from gosnark_amd import synth
    n = 1 << 12
    inst = synth.sqchain_setup_instance(n, 0x5A5A)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 99)
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    parts = [groth16.prove_partials(pk, inst.w, inst.px, k, shards)[0] for k in range(shards)]
    combined = [capi.sum_affine([parts[k][i] for k in range(shards)], g2=g2) for i, g2 in enumerate(groth16.SUM_IS_G2)]
    got = groth16.finish(pk, combined, r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    # world size 1 through the torch.distributed-shaped entry point
    one = groth16.prove_sharded(pk, inst.w, inst.px, r, s)
    assert (one.PiA, one.PiB, one.PiC) == (want.PiA, want.PiB, want.PiC)


def _x3_csr():
    rows = lambda mat: [{k: v for k, v in enumerate(row) if v} for row in mat]   # noqa: E731
    return tuple(r1csqap.csr_from_rows(rows(m)) for m in (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C))


def test_device_groth16_setup_equals_the_key_the_reference_verifier_accepted():
    """Golden `groth_x3`: its key came from the recorded toxic recipe and the reference's own grothVerifyProofs accepted a
    proof made with it.  gs_groth16_setup with the same toxic values must rebuild exactly that key (affine)."""
    rec = GU.load("groth_x3")
    assert [v["result"] for v in rec["verify"]] == ["true", "false"]
    toxic = tuple(int.from_bytes(bytes((i * k + 7) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13))
    a, b, c = _x3_csr()
    dpk, vk = groth16.GenerateTrustedSetupSparse(7, 8, 1, a, b, c, toxic)
    opk = GU.groth_pk(rec["setup"])
    for name, ref in (("G1_At", opk.G1_At), ("G1_BACGamma", opk.G1_BACGamma), ("BACDelta", opk.BACDelta), ("PowersTauDelta", opk.PowersTauDelta)):
        assert groth16.ExportPkArray(dpk, name) == [jac_affine_g1(p) for p in ref], name
    assert groth16.ExportPkArray(dpk, "G2_BACGamma") == [jac_affine_g2(p) for p in opk.G2_BACGamma]
    svk = rec["setup"]["Vk"]
    assert vk.IC == [jac_affine_g1(GU.g1(p)) for p in svk["IC"]]
    assert vk.G1_Alpha == jac_affine_g1(GU.g1(svk["G1"]["Alpha"])) and vk.G2_Gamma == jac_affine_g2(GU.g2(svk["G2"]["Gamma"]))
    # and the proof with the device-built key equals the reference prover's proof
    r, s = GU.rs_from_stream(rec["rand"])
    proof = groth16.GenerateProofsWithRS(groth16.Circuit(8, 1), dpk, rec["w"], rec["px"], r, s)
    assert proof.PiA == jac_affine_g1(GU.g1(rec["proof"]["PiA"])) and proof.PiC == jac_affine_g1(GU.g1(rec["proof"]["PiC"]))
    assert proof.PiB == jac_affine_g2(GU.g2(rec["proof"]["PiB"]))


def test_device_pinocchio_setup_equals_the_key_the_reference_verifier_accepted():
    """Golden `pinocchio_x3_setup` (reference VerifyProof: true / false): gs_pinocchio_setup rebuilds that key from the toxic
    recipe, and the proof made with the device-built key equals the reference prover's proof."""
    rec = GU.load("pinocchio_x3_setup")
    assert [v["result"] for v in rec["verify"]] == ["true", "false"]
    toxic = tuple(int.from_bytes(bytes((i * k + 9) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13, 17, 19, 23))
    a, b, c = _x3_csr()
    dpk, vk = snark.GenerateTrustedSetupSparse(7, 8, 1, a, b, c, toxic)
    spk, svk = rec["setup"]["Pk"], rec["setup"]["Vk"]
    for k in ("Bp", "C", "Cp", "Kp", "G1T"):
        assert snark.ExportPkArray(dpk, k) == [jac_affine_g1(GU.g1(p)) for p in spk[k]], k
    for k in ("A", "Ap"):        # resident A / Ap carry infinity for i <= NPublic (what snark.go:265 sums)
        want = [jac_affine_g1(GU.g1(p)) for p in spk[k]]
        assert snark.ExportPkArray(dpk, k) == [(0, 0, 0)] * 2 + want[2:], k
    assert snark.ExportPkArray(dpk, "B") == [jac_affine_g2(GU.g2(p)) for p in spk["B"]]
    assert vk.IC == [jac_affine_g1(GU.g1(p)) for p in svk["IC"]]
    assert vk.Vkb == jac_affine_g1(GU.g1(svk["Vkb"])) and vk.G1Kbg == jac_affine_g1(GU.g1(svk["G1Kbg"]))
    for k in ("Vka", "Vkc", "G2Kbg", "G2Kg", "Vkz"):
        assert getattr(vk, k) == jac_affine_g2(GU.g2(svk[k])), k
    proof = snark.GenerateProofs(snark.Circuit(8, 1), dpk, rec["w"], rec["px"])
    for k in ("PiA", "PiAp", "PiBp", "PiC", "PiCp", "PiH", "PiKp"):
        assert getattr(proof, k) == jac_affine_g1(GU.g1(rec["proof"][k])), k
    assert proof.PiB == jac_affine_g2(GU.g2(rec["proof"]["PiB"]))


def test_pipelined_proving_three_in_flight_equals_blocking_calls():
    """gs_groth16_prove_begin / _end: up to three proofs outstanding on disjoint workspaces; results equal the blocking entry point
    (different witnesses and randomness per proof, collected in order); a fourth begin is refused with GS_ERR_BUSY, every other
    entry point keeps working while tickets are outstanding (it queues behind their device work)."""
    from gosnark_amd import synth
    n = 1 << 12
    inst = synth.sqchain_setup_instance(n, 0x717E)
    pk = inst.device_pk()
    inst_bases_dummy()
    _, _, _, w2 = synth.sqchain_r1cs(n, 424242)
    _, _, _, px2 = r1csqap.ComputePx(*inst.r1cs, w2, n + 1)
    w2h, px2h = capi.scalars_upload(w2), capi.scalars_upload(px2)
    rs = [tuple(synth.field_elems(2, 500 + i)) for i in range(5)]
    inputs = [(inst.w, inst.px), (w2h, px2h), (inst.w, inst.px), (w2h, px2h), (inst.w, inst.px)]
    want = [groth16.prove_resident(pk, w, px, r, s) for (w, px), (r, s) in zip(inputs, rs)]
    got, tickets = [], []
    for (w, px), (r, s) in zip(inputs, rs):
        tickets.append(groth16.prove_begin(pk, w, px, r, s))
        if len(tickets) == 3:
            with pytest.raises(capi.GosnarkHipError) as busy:  # only three may be outstanding
                groth16.prove_begin(pk, w, px, r, s)
            assert busy.value.code == -6                       # GS_ERR_BUSY
            # everything else still runs meanwhile: 5 * G through the blocking MSM entry point
            five = capi.msm(inst_bases_dummy(), capi.ints_to_u64([5]))
            assert five == O.G1.Affine(O.G1.MulScalar(O.G1_GEN, 5))[:2]
            got.append(groth16.prove_end(tickets.pop(0)))
    while tickets:
        got.append(groth16.prove_end(tickets.pop(0)))
    assert [(p.PiA, p.PiB, p.PiC) for p in got] == [(p.PiA, p.PiB, p.PiC) for p in want]
    with pytest.raises(capi.GosnarkHipError):
        groth16.prove_end(123456)                                # unknown ticket


_DUMMY_BASES = []


def inst_bases_dummy():
    if not _DUMMY_BASES:
        _DUMMY_BASES.append(capi.g1_upload(capi.g1_points_to_u64([O.G1_GEN])))
    return _DUMMY_BASES[0]


@pytest.mark.parametrize("logn", [3, 9])
def test_full_pipeline_m_equals_n_plus_2(logn):
    """The other shape the reference accepts (m = n + 2, e.g. snark_test.go:280-290: n = 4, m = 6): deg Z = n, len(hx) = n - 1,
    len(PowersTauDelta) = n + 1.  Closed-form check as above, plus shape errors for m outside [n + 1, 2n + 1]."""
    from gosnark_amd import synth
    n = 1 << logn
    inst = synth.sqchain_setup_instance(n, 0xC0DE00 + logn, extra_vars=1)
    assert inst.m == n + 2
    r, s = synth.field_elems(2, 6060 + logn)
    proof = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    a, b, c = inst.expected_proof_scalars(r, s)
    assert (proof.PiA[0], proof.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    assert (proof.PiB[0], proof.PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
    assert (proof.PiC[0], proof.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    ra, rb, rc = inst.r1cs
    with pytest.raises(capi.GosnarkHipError):          # m = n: len(hx) would exceed len(PowersTauDelta)
        groth16.GenerateTrustedSetupSparse(n, n, 1, ra, rb, rc, inst.toxic)


def test_resident_r1cs_px_equals_host_buffer_path_and_feeds_the_prover():
    """gs_r1cs_upload + gs_r1cs_px (everything resident) == gs_r1cs_to_px (host buffers), the px handle can be overwritten for
    the next witness, and it feeds the prover directly: R1CS + witness -> proof without touching the host."""
    from gosnark_amd import synth
    n = 1 << 10
    inst = synth.sqchain_setup_instance(n, 0xD00D)
    a, b, c = inst.r1cs
    dev = r1csqap.DeviceR1CS(a, b, c, n + 1)
    px1 = dev.ComputePxResident(inst.w)
    assert np.array_equal(capi.scalars_download(px1), inst.px_host)
    _, _, _, w2 = synth.sqchain_r1cs(n, 777)
    _, _, _, want2 = r1csqap.ComputePx(a, b, c, w2, n + 1)
    w2h = capi.scalars_upload(w2)
    px2 = dev.ComputePxResident(w2h, px1)                       # overwrite in place
    assert px2 is px1 and np.array_equal(capi.scalars_download(px1), want2)
    r, s = synth.field_elems(2, 31)
    got = groth16.prove_resident(inst.device_pk(), w2h, px1, r, s)
    want = groth16.prove_resident(inst.device_pk(), w2h, capi.scalars_upload(want2), r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    with pytest.raises(capi.GosnarkHipError):                   # witness length must match the system
        dev.ComputePxResident(capi.scalars_upload(w2[:-1]))


def test_resident_key_round_trips_through_the_wire_formats(tmp_path):
    """a key built on the device is written as the binary limb container (gs_groth16_pk_export 0..6), mapped
    back and uploaded without becoming Python integers, and proves the reference's recorded proof; the same file, read as
    host integers and printed with the reference's *String layout, is the golden key in affine form."""
    from gosnark_amd import utils
    rec = GU.load("groth_x3")
    toxic = tuple(int.from_bytes(bytes((i * k + 7) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13))
    a, b, c = _x3_csr()
    dpk, vk = groth16.GenerateTrustedSetupSparse(7, 8, 1, a, b, c, toxic)
    path = str(tmp_path / "x3.gskey")
    utils.GrothSetupToBinary(path, groth16.Circuit(8, 1), dpk, vk)
    circ, dpk2 = utils.UploadGrothPkBinary(path)
    assert (circ.NVars, circ.NPublic) == (8, 1)
    r, s = GU.rs_from_stream(rec["rand"])
    proof = groth16.GenerateProofsWithRS(circ, dpk2, rec["w"], rec["px"], r, s)
    assert utils.GrothProofToString(proof) == {"PiA": [str(x) for x in jac_affine_g1(GU.g1(rec["proof"]["PiA"]))],
                                               "PiB": [[str(x) for x in cc] for cc in jac_affine_g2(GU.g2(rec["proof"]["PiB"]))],
                                               "PiC": [str(x) for x in jac_affine_g1(GU.g1(rec["proof"]["PiC"]))]}
    _, hpk = utils.GrothPkFromBinary(path)
    opk = GU.groth_pk(rec["setup"])
    assert hpk.Z == [z % O.R for z in opk.Z]
    assert (hpk.G1_Alpha, hpk.G1_Beta, hpk.G1_Delta) == tuple(jac_affine_g1(p) for p in (opk.G1_Alpha, opk.G1_Beta, opk.G1_Delta))
    assert (hpk.G2_Beta, hpk.G2_Delta) == (jac_affine_g2(opk.G2_Beta), jac_affine_g2(opk.G2_Delta))
    assert hpk.G1_At == [jac_affine_g1(p) for p in opk.G1_At] and hpk.G2_BACGamma == [jac_affine_g2(p) for p in opk.G2_BACGamma]
    svk = utils.GrothVkToString(utils.GrothVkFromBinary(path))
    want_vk = utils.GrothVkToString(utils.GrothVkFromString(rec["setup"]["Vk"]))
    aff = lambda d: json_affine(d)   # noqa: E731
    assert aff(svk) == aff(want_vk)


def json_affine(vk_strings):
    from gosnark_amd import utils
    vk = utils.GrothVkFromString(vk_strings)
    return ([jac_affine_g1(p) for p in vk.IC], jac_affine_g1(vk.G1_Alpha), jac_affine_g2(vk.G2_Beta), jac_affine_g2(vk.G2_Gamma),
            jac_affine_g2(vk.G2_Delta))


def test_device_pinocchio_setup_prove_verify_round_trip():
    """Pinocchio end to end on the device-built key: gs_pinocchio_setup -> gs_pinocchio_prove -> snark.VerifyProof accepts
    for the circuit's public output 35 and rejects 34 (the reference's wasm did the same with this recipe's key)."""
    rec = GU.load("pinocchio_x3_setup")
    toxic = tuple(int.from_bytes(bytes((i * k + 9) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13, 17, 19, 23))
    a, b, c = _x3_csr()
    dpk, vk = snark.GenerateTrustedSetupSparse(7, 8, 1, a, b, c, toxic)
    proof = snark.GenerateProofs(snark.Circuit(8, 1), dpk, rec["w"], rec["px"])
    assert snark.VerifyProof(vk, proof, [35]) is True
    assert snark.VerifyProof(vk, proof, [34]) is False


@pytest.mark.parametrize("n", [16, 1000, 1 << 16])
def test_pinocchio_full_pipeline_proofs_verify_at_sizes_the_reference_cannot_replay(n):
    """snark.GenerateProofs at scale: sparse R1CS -> gs_pinocchio_setup -> px on the device -> the eight proof
    elements -> snark.VerifyProof (five pairing equations against the device-built vk) accepts for the instance's public
    input and fails the divisibility equation for any other; a corrupted element fails its own knowledge-commitment check."""
    from gosnark_amd import synth
    inst = synth.sqchain_pinocchio_instance(n, 0xFACE00 + n % 251)
    proof = snark.prove_resident(inst.device_pk(), inst.w, inst.px)
    assert snark.VerifyProof(inst.vk, proof, inst.public) is True
    assert snark.VerifyProof(inst.vk, proof, [(inst.public[0] + 1) % O.R]) is False
    fields = {k: getattr(proof, k) for k in snark.Proof.FIELDS}
    fields["PiH"] = O.G1.Double(fields["PiH"])
    assert snark.VerifyProof(inst.vk, snark.Proof(**fields), inst.public) is False
    # host-buffer entry point gives the same eight elements
    w = capi.u64_to_ints(inst.w_host)
    px = capi.u64_to_ints(inst.px_host)
    if n <= 1000:
        again = snark.GenerateProofs(snark.Circuit(inst.m, 1), inst.device_pk(), w, px)
        assert all(getattr(again, k) == getattr(proof, k) for k in snark.Proof.FIELDS)


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_sharded_prove_with_key_slices(shards, tmp_path):
    """SURVEY 8e "each GPU holds 1/8 of every pk array": every logical rank holds ONLY its slice of the key -- cut from a
    resident key (gs_groth16_pk_shard), uploaded from host slices (gs_groth16_pk_create_shard) or mapped from the binary
    container -- and the proof assembled from the partial sums equals the single-device proof.  n + 1 = 4098 variables do
    not divide by 3 or 8 (ragged ranges)."""
    from gosnark_amd import synth, utils
    n = (1 << 12) + 1
    inst = synth.sqchain_setup_instance(n, 0x51CE)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 199)
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)

    def assemble(keys):
        parts = [groth16.prove_partials(keys[k], inst.w, inst.px, k, shards)[0] for k in range(shards)]
        combined = [capi.sum_affine([parts[k][i] for k in range(shards)], g2=g2) for i, g2 in enumerate(groth16.SUM_IS_G2)]
        got = groth16.finish(keys[shards - 1], combined, r, s)
        return (got.PiA, got.PiB, got.PiC)
    slices = [groth16.ShardPk(pk, k, shards) for k in range(shards)]
    assert assemble(slices) == (want.PiA, want.PiB, want.PiC)
    # a slice holds 1/shards of the arrays, and refuses everything but its own shard
    lens = [len(sl.handle) for sl in slices]
    assert sum(lens) == n + 1 and max(lens) - min(lens) <= 1
    with pytest.raises(capi.GosnarkHipError, match="holds shard"):
        groth16.prove_resident(slices[0], inst.w, inst.px, r, s)
    with pytest.raises(capi.GosnarkHipError, match="holds shard"):
        groth16.prove_partials(slices[0], inst.w, inst.px, 1, shards)
    with pytest.raises(capi.GosnarkHipError, match="itself a slice"):
        groth16.ShardPk(slices[0], 0, 2)
    # the same through the binary container: each rank maps only its slices of the file
    path = str(tmp_path / "key.gskey")
    utils.GrothSetupToBinary(path, groth16.Circuit(n + 1, 1), pk, inst.vk)
    mapped = [utils.UploadGrothPkBinary(path, shard=(k, shards))[1] for k in range(shards)]
    assert assemble(mapped) == (want.PiA, want.PiB, want.PiC)
    if shards == 2:      # and from host integers (groth16.UploadPkShard)
        circ, hpk = utils.GrothPkFromBinary(path)
        host = [groth16.UploadPkShard(hpk, circ, k, shards) for k in range(shards)]
        assert assemble(host) == (want.PiA, want.PiB, want.PiC)
        assert groth16.VerifyProof(inst.vk, groth16.Proof(*assemble(host)), capi.u64_to_ints(inst.w_host[1:2])) is True


@pytest.mark.parametrize("n", [7, 1000, 1 << 14])
def test_one_call_r1cs_to_proof_equals_the_two_step_path(n):
    """gs_groth16_prove_r1cs == gs_r1cs_px then gs_groth16_prove_resident: same px, same proof, verifier accepts."""
    from gosnark_amd import synth
    inst = synth.sqchain_setup_instance(n, 0xC0DE + n % 97)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 777 + n % 97)
    px2 = dev.ComputePxResident(inst.w)
    want = groth16.prove_resident(inst.device_pk(), inst.w, px2, r, s)
    got, px1 = groth16.prove_from_r1cs(inst.device_pk(), dev, inst.w, r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    assert np.array_equal(capi.scalars_download(px1), capi.scalars_download(px2))
    assert np.array_equal(capi.scalars_download(px1), np.asarray(inst.px_host, dtype=np.uint64).reshape(-1, 4))
    again, px1b = groth16.prove_from_r1cs(inst.device_pk(), dev, inst.w, r, s, px1)       # overwrite in place
    assert px1b is px1 and (again.PiA, again.PiB, again.PiC) == (want.PiA, want.PiB, want.PiC)
    assert groth16.VerifyProof(inst.vk, got, capi.u64_to_ints(inst.w_host[1:2])) is True


def _random_circuit(n, npub, seed):
    """A random satisfied R1CS in the only shape the reference supports (NVars = n + 1): variables [one, p_1..p_npub,
    v_1..]; constraints 1..npub: p_i * one = p_i; every later constraint multiplies two random sparse combinations of
    earlier variables into a NEW variable.  -> dense rows (dicts), witness ints."""
    rng = random.Random(seed)
    w = [1] + [rng.randrange(O.R) for _ in range(npub)]
    A, B, Cc = [], [], []
    for i in range(1, npub + 1):
        A.append({i: 1}); B.append({0: 1}); Cc.append({i: 1})
    while len(A) < n:
        def combo():
            ks = rng.sample(range(len(w)), min(len(w), rng.randint(1, 3)))
            return {k: rng.randrange(1, O.R) for k in ks}
        a, b = combo(), combo()
        va = sum(c * w[k] for k, c in a.items()) % O.R
        vb = sum(c * w[k] for k, c in b.items()) % O.R
        w.append(va * vb % O.R)
        A.append(a); B.append(b); Cc.append({len(w) - 1: 1})
    assert len(w) == n + 1
    return A, B, Cc, w


def _dense(rows, m):
    return [[row.get(k, 0) for k in range(m)] for row in rows]


def test_random_circuit_with_three_public_inputs_equals_the_oracle_and_verifies():
    """General sparse matrices and NPublic = 3 (IC accumulation over several signals, BACDelta / A / Ap zeroed for i <= 3):
    device setup == the oracle's GenerateTrustedSetup on the dense QAP, device proof == oracle proof, and both verifiers
    accept exactly the right public inputs."""
    n, npub = 12, 3
    A, B, Cc, w = _random_circuit(n, npub, 4242)
    m = n + 1
    csr = [r1csqap.csr_from_rows(x) for x in (A, B, Cc)]
    rng = random.Random(5151)
    toxic = tuple(rng.randrange(1, O.R) for _ in range(5))
    al, be, ga, _ = O.PF.R1CSToQAP(_dense(A, m), _dense(B, m), _dense(Cc, m))
    opk, ovk = O.groth16_GenerateTrustedSetup(m, npub, al, be, ga, toxic)
    dpk, vk = groth16.GenerateTrustedSetupSparse(n, m, npub, *csr, toxic)
    for name, ref in (("G1_At", opk.G1_At), ("G1_BACGamma", opk.G1_BACGamma), ("BACDelta", opk.BACDelta), ("PowersTauDelta", opk.PowersTauDelta)):
        assert groth16.ExportPkArray(dpk, name) == [jac_affine_g1(p) for p in ref], name
    assert vk.IC == [jac_affine_g1(p) for p in ovk.IC] and len(vk.IC) == npub + 1
    _, _, _, px = O.PF.CombinePolynomials(w, al, be, ga)
    r, s = rng.randrange(O.R), rng.randrange(O.R)
    want = O.groth16_GenerateProofs(m, npub, opk, w, px, r, s)
    got = groth16.GenerateProofsWithRS(groth16.Circuit(m, npub), dpk, w, px, r, s)
    assert got.PiA == jac_affine_g1(want[0]) and got.PiB == jac_affine_g2(want[1]) and got.PiC == jac_affine_g1(want[2])
    pub = w[1:1 + npub]
    assert groth16.VerifyProof(vk, got, pub) is True
    assert groth16.VerifyProof(vk, got, [pub[1], pub[0], pub[2]]) is False
    assert groth16.VerifyProof(vk, got, pub[:2]) is False
    # Pinocchio on the same system
    ptox = tuple(rng.randrange(1, O.R) for _ in range(8))
    ppk, pvk = snark.GenerateTrustedSetupSparse(n, m, npub, *csr, ptox)
    pproof = snark.GenerateProofs(snark.Circuit(m, npub), ppk, w, px)
    assert snark.VerifyProof(pvk, pproof, pub) is True
    assert snark.VerifyProof(pvk, pproof, [pub[0], pub[1], (pub[2] + 1) % O.R]) is False


@pytest.mark.parametrize("n,npub", [(200, 2), (1500, 5)])
def test_random_circuits_prove_and_verify_end_to_end(n, npub):
    A, B, Cc, w = _random_circuit(n, npub, 77 + n)
    m = n + 1
    csr = [r1csqap.csr_from_rows(x) for x in (A, B, Cc)]
    rng = random.Random(99 + n)
    dpk, vk = groth16.GenerateTrustedSetupSparse(n, m, npub, *csr, tuple(rng.randrange(1, O.R) for _ in range(5)))
    wa = capi.ints_to_u64(w)
    _, _, _, px = r1csqap.ComputePx(*csr, wa, m)
    proof = groth16.prove_resident(dpk, capi.scalars_upload(wa), capi.scalars_upload(px), rng.randrange(O.R), rng.randrange(O.R))
    pub = w[1:1 + npub]
    assert groth16.VerifyProof(vk, proof, pub) is True
    bad = list(pub)
    bad[-1] = (bad[-1] + 1) % O.R
    assert groth16.VerifyProof(vk, proof, bad) is False
    ppk, pvk = snark.GenerateTrustedSetupSparse(n, m, npub, *csr, tuple(rng.randrange(1, O.R) for _ in range(8)))
    pproof = snark.prove_resident(ppk, capi.scalars_upload(wa), capi.scalars_upload(px))
    assert snark.VerifyProof(pvk, pproof, pub) is True
    assert snark.VerifyProof(pvk, pproof, bad) is False


def test_mixed_tickets_groth16_pinocchio_msm_share_the_three_slots():
    """gs_groth16_prove_begin, gs_pinocchio_prove_begin and gs_msm_g1_begin draw from the same three in-flight slots; results
    collected out of order equal the blocking calls; a fourth begin is refused; ends of the wrong kind are refused."""
    from gosnark_amd import synth
    n = 1 << 12
    g = synth.sqchain_setup_instance(n, 0xA1)
    p = synth.sqchain_pinocchio_instance(n, 0xA2)
    r, s = synth.field_elems(2, 31)
    want_g = groth16.prove_resident(g.device_pk(), g.w, g.px, r, s)
    want_p = snark.prove_resident(p.device_pk(), p.w, p.px)
    bases = capi.g1_fixed_base(U.rand_scalars_u64(3000, 41))
    sc = capi.scalars_upload(U.rand_scalars_u64(3000, 42))
    want_m = capi.msm_resident(bases, sc, 3000)
    for _ in range(3):
        tg = groth16.prove_begin(g.device_pk(), g.w, g.px, r, s)
        tp = snark.prove_begin(p.device_pk(), p.w, p.px)
        tm = capi.msm_begin(bases, sc, 3000)
        with pytest.raises(capi.GosnarkHipError, match="outstanding"):
            snark.prove_begin(p.device_pk(), p.w, p.px)
        with pytest.raises(capi.GosnarkHipError, match="not a Pinocchio proof"):
            snark.prove_end(tg)
        with pytest.raises(capi.GosnarkHipError):
            groth16.prove_end(tp)
        assert capi.msm_end(tm) == want_m
        got_p = snark.prove_end(tp)
        got_g = groth16.prove_end(tg)
        assert (got_g.PiA, got_g.PiB, got_g.PiC) == (want_g.PiA, want_g.PiB, want_g.PiC)
        assert all(getattr(got_p, k) == getattr(want_p, k) for k in snark.Proof.FIELDS)
    # a stream of Pinocchio proofs, three in flight
    tickets, outs = [], []
    for _ in range(7):
        tickets.append(snark.prove_begin(p.device_pk(), p.w, p.px))
        if len(tickets) == 3:
            outs.append(snark.prove_end(tickets.pop(0)))
    while tickets:
        outs.append(snark.prove_end(tickets.pop(0)))
    assert len(outs) == 7 and all(getattr(o, k) == getattr(want_p, k) for o in outs for k in snark.Proof.FIELDS)
    assert snark.VerifyProof(p.vk, outs[-1], p.public) is True


def test_pinocchio_resident_key_round_trips_through_the_binary_container(tmp_path):
    from gosnark_amd import utils
    rec = GU.load("pinocchio_x3_setup")
    toxic = tuple(int.from_bytes(bytes((i * k + 9) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13, 17, 19, 23))
    a, b, c = _x3_csr()
    dpk, vk = snark.GenerateTrustedSetupSparse(7, 8, 1, a, b, c, toxic)
    path = str(tmp_path / "pin.gskey")
    utils.SetupToBinary(path, snark.Circuit(8, 1), dpk, vk)
    circ, dpk2 = utils.UploadPkBinary(path)
    want = snark.GenerateProofs(snark.Circuit(8, 1), dpk, rec["w"], rec["px"])
    got = snark.GenerateProofs(circ, dpk2, rec["w"], rec["px"])
    assert all(getattr(got, k) == getattr(want, k) for k in snark.Proof.FIELDS)
    _, _, vk2 = utils.SetupFromBinary(path)
    assert snark.VerifyProof(vk2, got, [35]) is True and snark.VerifyProof(vk2, got, [34]) is False




def test_2p16_proof_on_random_keys_equals_the_naive_loop_golden():
    """VERDICT r1 next #7a: a committed golden AFFINE proof at n = 2^16 that does not come from this library at all: the key points
    k_i * G, the five MSMs (naive MulScalar / Add loops) and hx (schoolbook Div) were computed offline by oracle/gs_oracle.c on all
    host cores (oracle/gen_golden_large.py -> tests/golden/oracle_groth_2p16.json) for the instance synth.RandomInstance(n, seed)
    defines.  px / Z leaves a remainder here (uniform px), which groth16.go:266 discards: the floor quotient is what is compared."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_groth_2p16.json")) as f:
        rec = json.load(f)
    inst = synth.random_instance(rec["n"], rec["seed"])
    r, s = synth.field_elems(2, rec["seed"] + 10)
    assert (r, s) == (int(rec["r"]), int(rec["s"]))
    want = ((int(rec["PiA"][0]), int(rec["PiA"][1]), 1),
            ((int(rec["PiB"][0][0]), int(rec["PiB"][0][1])), (int(rec["PiB"][1][0]), int(rec["PiB"][1][1])), (1, 0)),
            (int(rec["PiC"][0]), int(rec["PiC"][1]), 1))
    got = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    assert (got.PiA, got.PiB, got.PiC) == want
    for c in (14, 16, 19):
        capi.set_window_bits(c)
        try:
            got = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
        finally:
            capi.set_window_bits(0)
        assert (got.PiA, got.PiB, got.PiC) == want, c


@pytest.mark.parametrize("n,extra", [(2, 0), (3, 0), (7, 0), (8, 0), (64, 0), (100, 1), (1000, 0), (1 << 12, 0), (1 << 12, 1), (5000, 0)])
def test_witness_to_proof_without_px_equals_the_px_route(n, extra):
    """gs_groth16_prove_witness (H(x) straight from the constraint values: node extension, one interpolation, Taylor shift) gives the
    proof of gs_r1cs_px + gs_groth16_prove_resident, for both shapes the reference accepts (m = n + 1: deg Z = n - 1; m = n + 2:
    deg Z = n), powers of two and ragged sizes; the closed form of the setup instance pins the value."""
    from gosnark_amd import synth
    inst = synth.sqchain_setup_instance(n, 0x4D00 + n, extra_vars=extra)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 9000 + n)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    # the key comes from gs_groth16_setup, so it carries the evaluation-basis copy of PowersTauDelta: with the switch on the h-MSM
    # runs over H's values (no interpolation), with it off over H's coefficients (interpolation + Taylor shift) -- same proof
    assert capi.pk_eval_count(inst.device_pk().handle) == n
    for on in (True, False):
        capi.set_eval_basis(on)
        try:
            got = groth16.prove_from_witness(inst.device_pk(), dev, inst.w, r, s)
            piped = groth16.prove_end(groth16.prove_witness_begin(inst.device_pk(), dev, inst.w, r, s))
        finally:
            capi.set_eval_basis(True)
        assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC), on
        assert (piped.PiA, piped.PiB, piped.PiC) == (want.PiA, want.PiB, want.PiC), on
        assert capi.last_timing()["fallbacks"] == 0
    assert groth16.VerifyProof(inst.vk, got, capi.u64_to_ints(inst.w_host[1:2]))
    if n >= 64:
        a, b, c = inst.expected_proof_scalars(r, s)
        assert (got.PiC[0], got.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))


@pytest.mark.parametrize("logn", [12, 18])
def test_realistic_witness_distribution_matches_the_closed_form(logn):
    """a witness of the shape the reference's CalculateWitness produces (circuitcompiler/circuit.go:158-182: about
    half zeros and ones, most of the rest below 2^32, few full-width values) instead of uniform 254-bit scalars -- >80 % of the
    plan's digits are zero (skipped) and the buckets of the digits 1 and of the small values' low windows are cut into thousands of
    chunks (k_heavy_combine).  px route, both witness routes and three pipelined tickets give the closed-form proof of the setup's
    toxic values; the verifier accepts it for the right public input only."""
    from gosnark_amd import synth
    n = 1 << logn
    inst = synth.realistic_setup_instance(n, 0x7EA1 + logn)
    assert inst.counts["zeros"] + inst.counts["ones"] > 0.4 * n and inst.counts["full_width"] < 0.15 * n
    r, s = synth.field_elems(2, 9300 + logn)
    pk = inst.device_pk()
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    a, b, c = inst.expected_proof_scalars(r, s)
    assert (want.PiA[0], want.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    assert (want.PiB[0], want.PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
    assert (want.PiC[0], want.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    for on in (True, False):
        capi.set_eval_basis(on)
        try:
            got = groth16.prove_from_witness(pk, dev, inst.w, r, s)
        finally:
            capi.set_eval_basis(True)
        assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC), on
        assert capi.last_timing()["fallbacks"] == 0
    tickets = [groth16.prove_begin(pk, inst.w, inst.px, r, s) for _ in range(3)]
    for t in tickets:
        p = groth16.prove_end(t)
        assert (p.PiA, p.PiB, p.PiC) == (want.PiA, want.PiB, want.PiC)
    x = capi.u64_to_ints(inst.w_host[1:2])[0]
    assert groth16.VerifyProof(inst.vk, want, [x]) is True and groth16.VerifyProof(inst.vk, want, [(x + 1) % O.R]) is False


@pytest.mark.parametrize("logn", [18, 20])
def test_witness_route_at_config_sizes_equals_px_route_and_closed_form(logn, _init):
    """the witness -> proof routes at 2^18 (BASELINE configs[4]) and 2^20 (configs[2], the headline size), inside
    pytest: evaluation-basis route (blocking and three pipelined tickets), coefficient route, px route and the closed form of
    the setup's toxic values all give the same proof; the verifier accepts it."""
    from gosnark_amd import synth
    n = 1 << logn
    inst = synth.sqchain_setup_instance(n, 0x4D77 + logn)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 9100 + logn)
    pk = inst.device_pk()
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    a, b, c = inst.expected_proof_scalars(r, s)
    assert (want.PiA[0], want.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    assert (want.PiC[0], want.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    got = groth16.prove_from_witness(pk, dev, inst.w, r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    tickets = [groth16.prove_witness_begin(pk, dev, inst.w, r, s) for _ in range(3)]
    for t in tickets:
        p = groth16.prove_end(t)
        assert (p.PiA, p.PiB, p.PiC) == (want.PiA, want.PiB, want.PiC)
    capi.set_eval_basis(False)
    try:
        coef = groth16.prove_from_witness(pk, dev, inst.w, r, s)
    finally:
        capi.set_eval_basis(True)
    assert (coef.PiA, coef.PiB, coef.PiC) == (want.PiA, want.PiB, want.PiC)
    assert groth16.VerifyProof(inst.vk, got, capi.u64_to_ints(inst.w_host[1:2])) is True
    # the table of the evaluation-basis array is visible to the memory accounting (64 B per constraint + its window rows)
    obj_b, tab_b = capi.handle_bytes(pk.handle.h)
    assert obj_b >= 5 * n * 64 + n * 128 and (tab_b >= 6 * 8 * n * 64 if _init == "always" else tab_b == 0)


def test_eval_basis_round_trips_through_export_and_attach():
    """gs_groth16_pk_export which = 7 reads the evaluation-basis array back; a key rebuilt from its exported arrays
    (gs_groth16_pk_create: no evaluation basis, coefficient route) gives the same witness proof, and again after
    gs_groth16_pk_set_eval attached the exported array (evaluation-basis route).  The array is what the header says:
    sum_j H(n+j) E[j-1] == sum_i h_i PowersTauDelta[i] checked through the two MSMs on an independent H."""
    from gosnark_amd import synth
    n = 500
    inst = synth.sqchain_setup_instance(n, 0x4F10)
    pk = inst.device_pk()
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 515)
    want = groth16.prove_from_witness(pk, dev, inst.w, r, s)
    arrays = {k: groth16.ExportPkArray(pk, k) for k in groth16.PK_ARRAYS}
    assert len(arrays["PowersTauDeltaEval"]) == n
    singles = np.zeros(84, dtype=np.uint64)
    capi.check(capi.load_library().gs_groth16_pk_export(capi.Handle(pk.handle.h), 5, capi.ptr64(singles), 5))
    v = capi.u64_to_ints(singles)
    z = np.zeros((inst.m - 1, 4), dtype=np.uint64)
    capi.check(capi.load_library().gs_groth16_pk_export(capi.Handle(pk.handle.h), 6, capi.ptr64(z), inst.m - 1))
    hpk = groth16.Pk(BACDelta=arrays["BACDelta"], Z=capi.u64_to_ints(z), G1_Alpha=(v[0], v[1], v[2]), G1_Beta=(v[3], v[4], v[5]),
                     G1_Delta=(v[6], v[7], v[8]), G1_At=arrays["G1_At"], G1_BACGamma=arrays["G1_BACGamma"],
                     G2_Beta=((v[9], v[10]), (v[11], v[12]), (v[13], v[14])), G2_Delta=((v[15], v[16]), (v[17], v[18]), (v[19], v[20])),
                     G2_BACGamma=arrays["G2_BACGamma"], PowersTauDelta=arrays["PowersTauDelta"])
    foreign = groth16.UploadPk(hpk, groth16.Circuit(inst.m, 1))
    assert capi.pk_eval_count(foreign.handle) == 0
    got = groth16.prove_from_witness(foreign, dev, inst.w, r, s)            # coefficient route
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    groth16.SetEvalBasis(foreign, arrays["PowersTauDeltaEval"])
    assert capi.pk_eval_count(foreign.handle) == n
    got = groth16.prove_from_witness(foreign, dev, inst.w, r, s)            # evaluation-basis route on the attached array
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    # the defining identity on an H that has nothing to do with the instance: random coefficients h (degree n - 1), values by Horner
    rng = random.Random(99)
    h = [rng.randrange(O.R) for _ in range(n)]
    vals = [sum(c * pow(n + j, i, O.R) for i, c in enumerate(h)) % O.R for j in range(1, n + 1)]
    mono = capi.msm(capi.g1_upload(capi.ints_to_u64([c for p in arrays["PowersTauDelta"] for c in p]).reshape(-1, 12)), capi.ints_to_u64(h))
    ev = capi.msm(capi.g1_upload(capi.ints_to_u64([c for p in arrays["PowersTauDeltaEval"] for c in p]).reshape(-1, 12)), capi.ints_to_u64(vals))
    assert mono == ev
    with pytest.raises(capi.GosnarkHipError):
        groth16.SetEvalBasis(foreign, arrays["PowersTauDeltaEval"][:n - 2])     # neither deg Z nor deg Z + 1 points


def test_binary_key_container_carries_the_evaluation_basis(tmp_path):
    """The binary key file written from a device-built key has the optional PowersTauDeltaEval / G1TEval sections; a key uploaded
    from it (file -> memmap -> HBM) takes the evaluation-basis witness route and gives the same proof as the key it came from."""
    from gosnark_amd import synth, utils
    n = 300
    inst = synth.sqchain_setup_instance(n, 0x4F20)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 616)
    want = groth16.prove_from_witness(inst.device_pk(), dev, inst.w, r, s)
    path = str(tmp_path / "groth.key")
    utils.GrothSetupToBinary(path, groth16.Circuit(inst.m, 1), inst.device_pk(), inst.vk)
    assert utils.ReadBinary(path)[3]["PowersTauDeltaEval"].shape == (n, 12)
    _, loaded = utils.UploadGrothPkBinary(path)
    assert capi.pk_eval_count(loaded.handle) == n
    got = groth16.prove_from_witness(loaded, dev, inst.w, r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC) and capi.last_timing()["fallbacks"] == 0
    pin = synth.sqchain_pinocchio_instance(n, 0x4F21)
    pdev = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
    pwant = snark.prove_from_witness(pin.device_pk(), pdev, pin.w)
    ppath = str(tmp_path / "pinocchio.key")
    utils.SetupToBinary(ppath, snark.Circuit(pin.m, 1), pin.device_pk(), pin.vk)
    _, ploaded = utils.UploadPkBinary(ppath)
    assert capi.pk_eval_count(ploaded.handle) == n
    pgot = snark.prove_from_witness(ploaded, pdev, pin.w)
    # (a resident Pinocchio key's A / Ap hold infinity for i <= NPublic, and that is what the file carries: same sums)
    assert all(getattr(pgot, k) == getattr(pwant, k) for k in snark.Proof.FIELDS)


def test_witness_to_proof_falls_back_to_the_exact_quotient_for_a_violated_constraint():
    """A witness that breaks a constraint makes A B - C a non-multiple of Z: the reference still returns floor(px / Z) (remainder
    discarded, groth16.go:266).  gs_groth16_prove_witness detects the violation and takes that route: same (meaningless) proof."""
    from gosnark_amd import synth
    n = 300
    inst = synth.sqchain_setup_instance(n, 0x4E00)
    w_bad = inst.w_host.copy()
    w_bad[17] = (12345, 0, 0, 0)
    _, _, _, px_bad = r1csqap.ComputePx(*inst.r1cs, w_bad, inst.m)
    wh, pxh = capi.scalars_upload(w_bad), capi.scalars_upload(px_bad)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 4242)
    want = groth16.prove_resident(inst.device_pk(), wh, pxh, r, s)
    good = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    for on in (True, False):       # evaluation-basis route: the violation is only seen when the proof is collected, then repeated exactly
        capi.set_eval_basis(on)
        try:
            got = groth16.prove_from_witness(inst.device_pk(), dev, wh, r, s)
            assert capi.last_timing()["fallbacks"] == (1 if on else 0)
            # pipelined: a bad ticket between two good ones falls back inside gs_groth16_prove_end and disturbs neither neighbour
            t = [groth16.prove_witness_begin(inst.device_pk(), dev, x, r, s) for x in (inst.w, wh, inst.w)]
            res = [groth16.prove_end(x) for x in t]
        finally:
            capi.set_eval_basis(True)
        assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC), on
        assert [(p.PiA, p.PiB, p.PiC) for p in res] == [(q.PiA, q.PiB, q.PiC) for q in (good, want, good)], on
    assert not groth16.VerifyProof(inst.vk, got, capi.u64_to_ints(w_bad[1:2]))
    # gs_trim between begin and end drops every cached workspace; the violated-constraint word of the ticket is context-owned
    # and survives: the bad ticket is still detected and repeated exactly, the good one still collected
    t_bad = groth16.prove_witness_begin(inst.device_pk(), dev, wh, r, s)
    t_good = groth16.prove_witness_begin(inst.device_pk(), dev, inst.w, r, s)
    capi.trim()
    p_bad, p_good = groth16.prove_end(t_bad), groth16.prove_end(t_good)
    assert (p_bad.PiA, p_bad.PiB, p_bad.PiC) == (want.PiA, want.PiB, want.PiC)
    assert (p_good.PiA, p_good.PiB, p_good.PiC) == (good.PiA, good.PiB, good.PiC)


def test_memory_accounting_and_table_eviction_leave_results_unchanged(_init):
    """gs_memory_query / gs_handle_bytes see a key's window tables (W - ... rows per base array, many times the key data);
    gs_release_tables frees them (device memory comes back) and the next proof rebuilds them; gs_trim drops every cached
    workspace; the proof is the same before and after both, and an in-flight ticket is waited for, not broken."""
    from gosnark_amd import synth
    if _init != "always":
        pytest.skip("the accounting of window tables needs the policy that builds them (table-free: tests/test_gpu_table_policy.py)")
    n = 1 << 12
    inst = synth.sqchain_setup_instance(n, 0x5A00)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 777)
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    obj_b, tab_b = capi.handle_bytes(pk.handle.h)
    key_points = 4 * (n + 1) * 64 + (n + 1) * 128              # 4 G1 arrays + 1 G2 array of about m entries, packed affine
    assert obj_b >= key_points - 4 * 64 * 8 and tab_b >= 8 * obj_b // 2        # tables: >= 8 rows of every array
    t = groth16.prove_begin(pk, inst.w, inst.px, r, s)          # a ticket in flight reads the tables: release must queue behind it
    before = capi.memory_query()
    assert before["table_bytes"] >= tab_b and before["library_bytes"] >= before["table_bytes"] + before["object_bytes"]
    assert before["device_total_bytes"] > before["device_free_bytes"] > 0 and before["objects"] >= 1
    capi.release_tables(pk.handle.h)
    assert capi.handle_bytes(pk.handle.h) == (obj_b, 0)
    after = capi.memory_query()
    assert after["table_bytes"] == before["table_bytes"] - tab_b
    assert after["library_bytes"] <= before["library_bytes"] - tab_b
    got_t = groth16.prove_end(t)
    assert (got_t.PiA, got_t.PiB, got_t.PiC) == (want.PiA, want.PiB, want.PiC)
    got = groth16.prove_resident(pk, inst.w, inst.px, r, s)     # rebuilds the tables
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    assert capi.handle_bytes(pk.handle.h)[1] == tab_b
    capi.trim()
    trimmed = capi.memory_query()
    assert trimmed["workspace_bytes"] == 0 and trimmed["library_bytes"] < capi.memory_query()["library_bytes"] + 1
    got = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    bases = capi.g1_fixed_base(synth.scalars_u64(1 << 10, 5))
    sc = capi.scalars_upload(synth.scalars_u64(1 << 10, 6))
    p0 = capi.msm_resident(bases, sc, 1 << 10)
    assert capi.handle_bytes(bases)[1] > 0
    capi.release_tables(bases)
    assert capi.handle_bytes(bases)[1] == 0 and capi.msm_resident(bases, sc, 1 << 10) == p0


@pytest.mark.parametrize("n,extra", [(16, 0), (300, 0), (300, 1), (1 << 12, 0), (1 << 12, 1), (1 << 16, 0)])
def test_pinocchio_witness_to_proof_without_px_equals_the_px_route(n, extra):
    """gs_pinocchio_prove_witness (snark.GenerateProofs with H(x) straight from the constraint values) gives the eight elements of
    gs_r1cs_px + gs_pinocchio_prove_resident, for both Z shapes (m = n + 1 / n + 2) and ragged sizes, and the five pairing
    equations accept it; a violated constraint takes the exact quotient route (same meaningless proof as the px route)."""
    from gosnark_amd import synth
    inst = synth.sqchain_pinocchio_instance(n, 0x5B00 + n % 97, extra_vars=extra)
    dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    want = snark.prove_resident(inst.device_pk(), inst.w, inst.px)
    assert capi.pk_eval_count(inst.device_pk().handle) == n       # gs_pinocchio_setup built the evaluation-basis copy of G1T
    for on in (True, False):                                      # H's values against it / H's coefficients against G1T
        capi.set_eval_basis(on)
        try:
            got = snark.prove_from_witness(inst.device_pk(), dev, inst.w)
            piped = snark.prove_end(snark.prove_witness_begin(inst.device_pk(), dev, inst.w))
        finally:
            capi.set_eval_basis(True)
        assert all(getattr(got, k) == getattr(want, k) for k in snark.Proof.FIELDS), on
        assert all(getattr(piped, k) == getattr(want, k) for k in snark.Proof.FIELDS), on
    if n == 300 and extra == 0:                                   # export (which = 9) / attach on a key without one
        ev = snark.ExportPkArray(inst.device_pk(), "G1TEval")
        assert len(ev) == n
    if n <= (1 << 12):
        assert snark.VerifyProof(inst.vk, got, inst.public) is True
    if n == 300:
        w_bad = inst.w_host.copy()
        w_bad[9] = (777, 0, 0, 0)
        _, _, _, px_bad = r1csqap.ComputePx(*inst.r1cs, w_bad, inst.m)
        wh, pxh = capi.scalars_upload(w_bad), capi.scalars_upload(px_bad)
        want_bad = snark.prove_resident(inst.device_pk(), wh, pxh)
        got_bad = snark.prove_from_witness(inst.device_pk(), dev, wh)
        assert all(getattr(got_bad, k) == getattr(want_bad, k) for k in snark.Proof.FIELDS)
        assert capi.last_timing()["fallbacks"] == 1
        piped_bad = snark.prove_end(snark.prove_witness_begin(inst.device_pk(), dev, wh))
        assert all(getattr(piped_bad, k) == getattr(want_bad, k) for k in snark.Proof.FIELDS)


def test_host_buffer_provers_at_2p16_equal_the_resident_ones():
    """gs_groth16_prove / gs_pinocchio_prove with w and px in pageable host memory at a size where both cross the boundary in
    several staged pieces (w 2 MiB, px 4 MiB; csrc/hostcopy.h) and px is copied BEHIND the already enqueued accumulations over w:
    same proof as the resident entry points; twice in a row (the staging buffers are reused)."""
    import ctypes
    from gosnark_amd import synth
    n = 1 << 16
    inst = synth.sqchain_setup_instance(n, 0x6C00)
    lib = capi.load_library()
    r, s = synth.field_elems(2, 31)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    rs = capi.ints_to_u64([r, s])
    for _ in range(2):
        out = np.zeros(32, dtype=np.uint64)
        inf = (ctypes.c_int * 3)()
        capi.check(lib.gs_groth16_prove(capi.Handle(inst.device_pk().handle.h), capi.ptr64(inst.w_host), inst.w_host.shape[0],
                                        capi.ptr64(inst.px_host), inst.px_host.shape[0], capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
        got = groth16._proof_from_words(out, inf)
        assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    pin = synth.sqchain_pinocchio_instance(n, 0x6C01)
    pwant = snark.prove_resident(pin.device_pk(), pin.w, pin.px)
    out = np.zeros(72, dtype=np.uint64)
    inf8 = (ctypes.c_int * 8)()
    capi.check(lib.gs_pinocchio_prove(capi.Handle(pin.device_pk().h), capi.ptr64(pin.w_host), pin.w_host.shape[0],
                                      capi.ptr64(pin.px_host), pin.px_host.shape[0], capi.ptr64(out), inf8))
    pgot = snark._proof_from_words(out, inf8)
    assert all(getattr(pgot, k) == getattr(pwant, k) for k in snark.Proof.FIELDS)


def test_2p16_pinocchio_proof_on_random_keys_equals_the_naive_loop_golden():
    """SURVEY 8 a2 at config size, pinned from outside the library: the eight elements of snark.GenerateProofs (snark.go:254-289) on
    synth.RandomPinocchioInstance(2^16, seed) were computed offline by the C restatement of the reference's naive loops on all host
    cores (oracle/gen_golden_large.py pinocchio -> tests/golden/oracle_pinocchio_2p16.json); resident, pipelined and at another
    window width the device gives the same affine points.  A and Ap sum over i > NPublic only (snark.go:265-268)."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_pinocchio_2p16.json")) as f:
        rec = json.load(f)
    inst = synth.random_pinocchio_instance(rec["n"], rec["seed"])

    def want(k):
        v = rec[k]
        return ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])), (1, 0)) if k == "PiB" else (int(v[0]), int(v[1]), 1)
    got = snark.prove_resident(inst.device_pk(), inst.w, inst.px)
    for k in snark.Proof.FIELDS:
        assert getattr(got, k) == want(k), k
    t = snark.prove_begin(inst.device_pk(), inst.w, inst.px)
    got = snark.prove_end(t)
    assert all(getattr(got, k) == want(k) for k in snark.Proof.FIELDS)
    capi.set_window_bits(13)
    try:
        got = snark.prove_resident(inst.device_pk(), inst.w, inst.px)
    finally:
        capi.set_window_bits(0)
    assert all(getattr(got, k) == want(k) for k in snark.Proof.FIELDS)


def test_2p20_four_of_the_five_sums_of_a_proof_equal_the_naive_loop_golden():
    """The headline size (BASELINE configs[2]) against the literal reference loops: the sums over w of groth16.go:243-250 -- G1.At,
    G1.BACGamma, G2.BACGamma over all variables, BACDelta over i > NPublic -- on synth.RandomInstance(2^20, seed), computed offline by
    oracle/gs_oracle.c on all host cores (oracle/gen_golden_large.py partials20 -> tests/golden/oracle_groth_partials_2p20.json),
    equal what gs_groth16_prove_partials returns in front of the O(1) tail.  (The fifth sum needs px / Z, a day of schoolbook Div
    at this size; it is pinned at 2^16 and, at 2^20, through the closed form and the verifier.)"""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_groth_partials_2p20.json")) as f:
        rec = json.load(f)
    inst = synth.random_instance(rec["n"], rec["seed"])
    pts, _ = groth16.prove_partials(inst.device_pk(), inst.w, inst.px, 0, 1)
    g1 = lambda k: (int(rec[k][0]), int(rec[k][1]))                                                     # noqa: E731
    assert pts[0] == g1("At")
    assert pts[1] == g1("BACGamma1")
    assert pts[2] == ((int(rec["BACGamma2"][0][0]), int(rec["BACGamma2"][0][1])), (int(rec["BACGamma2"][1][0]), int(rec["BACGamma2"][1][1])))
    assert pts[3] == g1("BACDelta")


@pytest.mark.parametrize("logn", [12, 20])
def test_complete_proof_on_a_known_quotient_equals_the_golden_from_outside_the_library(logn):
    """BASELINE configs[2] at its exact size, every element of the proof pinned from outside the library: on
    synth.QuotientInstance(n, seed) px = hx Z + rem, so floor(px / Z) = hx is known and the reference's day-long schoolbook Div is not
    needed.  oracle/gen_golden_large.py prove20 built px by an unrelated exact product (eighteen 31-bit NTT primes + Garner,
    oracle/crt_ntt.py -- checked against the schoolbook Mul in tests/test_oracle_crt_ntt.py), the five MSMs by the naive MulScalar /
    Add loops and the tail by oracle/ref_py.py.  Here: the px the LIBRARY builds (gs_zpoly, gs_poly_mul, gs_poly_add) has the
    recorded SHA-256, and the proof -- blocking, pipelined, and from host buffers -- is the golden one."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_groth_quotient_2p%d.json" % logn)) as f:
        rec = json.load(f)
    assert rec["n"] == 1 << logn and "px_sha256" in rec
    inst = synth.quotient_instance(rec["n"], rec["seed"])
    assert inst.px_sha256 == rec["px_sha256"]
    r, s = synth.field_elems(2, rec["seed"] + 10)
    assert (r, s) == (int(rec["r"]), int(rec["s"]))
    want = ((int(rec["PiA"][0]), int(rec["PiA"][1]), 1),
            ((int(rec["PiB"][0][0]), int(rec["PiB"][0][1])), (int(rec["PiB"][1][0]), int(rec["PiB"][1][1])), (1, 0)),
            (int(rec["PiC"][0]), int(rec["PiC"][1]), 1))
    got = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    assert (got.PiA, got.PiB, got.PiC) == want
    tickets = [groth16.prove_begin(inst.device_pk(), inst.w, inst.px, r, s) for _ in range(3)]
    for t in tickets:
        got = groth16.prove_end(t)
        assert (got.PiA, got.PiB, got.PiC) == want
    # the quotient itself
    hx = np.zeros((rec["n"], 4), dtype=np.uint64)
    capi.check(capi.load_library().gs_poly_div(capi.ptr64(inst.px_host), inst.px_host.shape[0], capi.ptr64(inst.z_host), inst.z_host.shape[0],
                                               capi.ptr64(hx), None))
    assert np.array_equal(hx, inst.hx_host)


@pytest.mark.parametrize("logn", [12, 20])
def test_complete_pinocchio_proof_on_a_known_quotient_equals_the_golden_from_outside_the_library(logn):
    """snark.GenerateProofs at the headline size, all eight elements pinned from outside the library: on
    synth.QuotientPinocchioInstance(n, seed) px = hx Z + rem (built there by the library, by oracle/crt_ntt.py in the generator: same
    SHA-256), the eight sums by the naive loops (oracle/gen_golden_large.py pinocchio20).  Blocking and pipelined."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_pinocchio_quotient_2p%d.json" % logn)) as f:
        rec = json.load(f)
    inst = synth.quotient_pinocchio_instance(rec["n"], rec["seed"])
    assert inst.px_sha256 == rec["px_sha256"]

    def want(k):
        v = rec[k]
        return ((int(v[0][0]), int(v[0][1])), (int(v[1][0]), int(v[1][1])), (1, 0)) if k == "PiB" else (int(v[0]), int(v[1]), 1)
    got = snark.prove_resident(inst.device_pk(), inst.w, inst.px)
    for k in snark.Proof.FIELDS:
        assert getattr(got, k) == want(k), k
    tickets = [snark.prove_begin(inst.device_pk(), inst.w, inst.px) for _ in range(3)]
    for t in tickets:
        got = snark.prove_end(t)
        assert all(getattr(got, k) == want(k) for k in snark.Proof.FIELDS)


@pytest.mark.parametrize("n", [3000, 1 << 14, (1 << 16) + 3])
def test_keys_with_sparse_b_arrays_sum_b1_and_b2_over_a_masked_plan(n):
    """Round 5: the reference's circuit compiler puts a signal into B only as the second operand of a product
    (circuitcompiler/circuit.go:110-128), so two thirds of the G1/G2.BACGamma points of such a key are the point at infinity.  From 4096
    variables on the prover sums B1 and B2 over a second plan of w that leaves those variables out (GrothPkObj::b_index: a second plan over the terms with a finite B point): same proof --
    pinned by the closed form of the setup's toxic values and the verifier -- on every entry route, with fewer G2 additions."""
    from gosnark_amd import synth
    inst = synth.gates_setup_instance(n, 0x9100 + n % 97)
    assert 0.25 < inst.counts["variables_in_B"] / inst.counts["variables"] < 0.45
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 9100 + n % 97)
    got = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    tm = capi.last_timing()
    a, b, c = inst.expected_proof_scalars(r, s)
    assert (got.PiA[0], got.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    assert (got.PiB[0], got.PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
    assert (got.PiC[0], got.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    x = capi.u64_to_ints(inst.w_host[1:2])[0]
    assert groth16.VerifyProof(inst.vk, got, [x]) and not groth16.VerifyProof(inst.vk, got, [(x + 1) % O.R])
    windows = 254 // tm["window_bits"] + 1
    if n >= 4096:      # the G2 sum ran over the masked plan: about a third of the (term, window) pairs
        assert tm["acc_g2_adds"] < 0.5 * windows * n, (tm["acc_g2_adds"], windows * n)
    else:
        assert tm["acc_g2_adds"] > 0.9 * windows * n
    same = lambda p: (p.PiA, p.PiB, p.PiC) == (got.PiA, got.PiB, got.PiC)     # noqa: E731
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    assert same(groth16.prove_from_witness(pk, dr, inst.w, r, s))
    t = [groth16.prove_begin(pk, inst.w, inst.px, r, s), groth16.prove_witness_begin(pk, dr, inst.w, r, s),
         groth16.prove_witness_host_begin(pk, dr, inst.w_host, r, s)]
    assert all(same(groth16.prove_end(k)) for k in t)
    t = [groth16.prove_host_begin(pk, inst.w_host, inst.px_host, r, s) for _ in range(3)]
    assert all(same(groth16.prove_end(k)) for k in t)
    # term ranges: three shards of the full key, and key slices (their masks are scanned per slice)
    from gosnark_amd import parallel
    parts = [groth16.prove_partials(pk, inst.w, inst.px, k, 3)[0] for k in range(3)]
    assert same(groth16.finish(pk, parallel.combine_partials(parts, groth16.SUM_IS_G2), r, s))
    slices = [groth16.ShardPk(pk, k, 2) for k in range(2)]
    parts = [groth16.prove_partials(slices[k], inst.w, inst.px, k, 2)[0] for k in range(2)]
    assert same(groth16.finish(pk, parallel.combine_partials(parts, groth16.SUM_IS_G2), r, s))


@pytest.mark.parametrize("cbits", [15, 17, 18, 19])
def test_sparse_b_keys_under_every_forced_window_width(cbits):
    """the split of B1 / B2 onto their own term-list plan was gated on `c < 19` while build_plan sorts
    partition-first -- a sort that takes no term list -- from c = 18 on (B = 2^17 buckets = 4 ranges of 2^15), so gs_set_window_bits(18)
    on a sparse-B key with its tables resident failed EVERY proof of both protocols with GS_ERR_HIP.  The gate now asks build_plan's own
    condition (msm.h, plan_partitions_first): 15 and 17 split, 18 and 19 keep the single plan, all four give the proof of the automatic
    width; the table-free route clamps a forced width to its 9..16 and always splits."""
    from gosnark_amd import synth
    n = 1 << 13
    inst = synth.gates_setup_instance(n, 0x9400)
    pin = synth.gates_pinocchio_instance(n, 0x9401)
    pk, ppk = inst.device_pk(), pin.device_pk()
    r, s = synth.field_elems(2, 9400)
    want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    pwant = snark.prove_resident(ppk, pin.w, pin.px)
    x = capi.u64_to_ints(inst.w_host[1:2])[0]
    assert groth16.VerifyProof(inst.vk, want, [x]) and snark.VerifyProof(pin.vk, pwant, pin.public)
    capi.set_window_bits(cbits)
    try:
        got = groth16.prove_resident(pk, inst.w, inst.px, r, s)
        tm = capi.last_timing()
        pgot = snark.prove_resident(ppk, pin.w, pin.px)
        t = [groth16.prove_host_begin(pk, inst.w_host, inst.px_host, r, s), groth16.prove_begin(pk, inst.w, inst.px, r, s)]
        tick = [groth16.prove_end(k) for k in t]
    finally:
        capi.set_window_bits(0)
    assert (got.PiA, got.PiB, got.PiC) == (want.PiA, want.PiB, want.PiC)
    assert all((p.PiA, p.PiB, p.PiC) == (want.PiA, want.PiB, want.PiC) for p in tick)
    assert all(getattr(pgot, k) == getattr(pwant, k) for k in snark.Proof.FIELDS)
    windows = 254 // tm["window_bits"] + 1
    split = tm["acc_g2_adds"] < 0.5 * windows * n
    assert split == (tm["window_bits"] < 18), (cbits, tm["window_bits"], tm["acc_g2_adds"])


@pytest.mark.parametrize("finite", ["none", "one", "54%", "56%", "all"])
def test_b_mask_edge_densities(finite):
    """The masked plan at its edges: a key whose B arrays hold NO finite point (the masked plan is empty: every kernel behind it sees zero
    entries), exactly one, a share just below and just above the 55 % threshold, all of them.  Bases are k_i G, so each of the four sums
    over w has the closed form (sum_i w_i k_i) G whatever plan carried it; gs_timing says which plan did."""
    n, npub = 5000, 1
    rng = np.random.Generator(np.random.PCG64(77))
    ks = {name: U.rand_scalars_u64(n, 9300 + j) for j, name in enumerate(("at", "b", "cd", "ptd"))}
    keep = {"none": np.zeros(n, bool), "one": np.arange(n) == 1234, "54%": rng.random(n) < 0.54, "56%": rng.random(n) < 0.56,
            "all": np.ones(n, bool)}[finite]
    ks["b"][~keep] = (0, 0, 0, 0)                                   # k = 0: the point at infinity in G1.BACGamma AND G2.BACGamma
    share = keep.mean()
    at, b1, b2, cd, ptd = (capi.g1_fixed_base(ks["at"]), capi.g1_fixed_base(ks["b"]), capi.g2_fixed_base(ks["b"]), capi.g1_fixed_base(ks["cd"]),
                           capi.g1_fixed_base(ks["ptd"]))
    g = lambda k: C.g1_mul_scalar(O.G1_GEN, k)                       # noqa: E731
    z = U.rand_scalars_u64(n + 1, 9310)
    pk = groth16.device_pk_from_handles(at, b1, b2, cd, ptd, g(3), g(5), g(7), C.g2_mul_scalar(O.G2_GEN, 5), C.g2_mul_scalar(O.G2_GEN, 7), z, n, npub)
    w_host, px_host = U.rand_scalars_u64(n, 9320), U.rand_scalars_u64(2 * n - 1, 9321)
    w, px = capi.scalars_upload(w_host), capi.scalars_upload(px_host)
    sums, _ = groth16.prove_partials(pk, w, px, 0, 1)
    tm = capi.last_timing()
    wi = U.u64_rows_to_ints(w_host)
    dot = lambda name, lo=0: sum(a * k for a, k in zip(wi[lo:], U.u64_rows_to_ints(ks[name])[lo:])) % O.R     # noqa: E731
    aff1 = lambda k: C.g1_affine(C.g1_mul_scalar(O.G1_GEN, k)) if k else None                                      # noqa: E731
    assert sums[0] == aff1(dot("at")) and sums[1] == aff1(dot("b")) and sums[3] == aff1(dot("cd", npub + 1))
    assert sums[2] == (C.g2_affine(C.g2_mul_scalar(O.G2_GEN, dot("b"))) if dot("b") else None)
    assert (sums[1] is None) == (finite == "none")
    windows = 254 // tm["window_bits"] + 1
    if share < 0.55:                  # B1 / B2 ran over the masked plan: as many G2 additions as the share of finite points
        assert tm["acc_g2_adds"] <= (share + 0.01) * windows * n + windows, (finite, tm["acc_g2_adds"])
    else:
        assert tm["acc_g2_adds"] > 0.9 * windows * n
    # term ranges: three shards of the key (the mask is read at the shard's offset)
    parts = [groth16.prove_partials(pk, w, px, k, 3)[0] for k in range(3)]
    from gosnark_amd import parallel
    assert parallel.combine_partials(parts, groth16.SUM_IS_G2) == sums


@pytest.mark.parametrize("n", [3000, (1 << 14) + 1])
def test_pinocchio_keys_with_sparse_b_arrays(n):
    """The same for snark.GenerateProofs: B (G2) and B' over the masked plan, the five other G1 sums over w's.  snark.VerifyProof's five
    equations tie B to B' and A, B, C, H together; every entry route gives the same proof; a proof from the key's two slices equals it."""
    from gosnark_amd import synth
    pin = synth.gates_pinocchio_instance(n, 0x9200 + n % 89)
    pk = pin.device_pk()
    got = snark.prove_resident(pk, pin.w, pin.px)
    tm = capi.last_timing()
    assert snark.VerifyProof(pin.vk, got, pin.public) and not snark.VerifyProof(pin.vk, got, [(pin.public[0] + 1) % O.R])
    windows = 254 // tm["window_bits"] + 1
    assert (tm["acc_g2_adds"] < 0.5 * windows * n) == (n >= 4096)
    same = lambda p: all(getattr(p, k) == getattr(got, k) for k in snark.Proof.FIELDS)     # noqa: E731
    dr = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
    assert same(snark.prove_from_witness(pk, dr, pin.w))
    t = [snark.prove_begin(pk, pin.w, pin.px), snark.prove_witness_host_begin(pk, dr, pin.w_host), snark.prove_host_begin(pk, pin.w_host, pin.px_host)]
    assert all(same(snark.prove_end(k)) for k in t)
    slices = [snark.ShardPk(pk, k, 2) for k in range(2)]
    recs = [snark.prove_partials(slices[k], pin.w, pin.px, k, 2) for k in range(2)]
    assert same(snark.combine(recs))


def test_small_mirrors_of_the_reference_seams():
    """The API names VERDICT r2 found missing, each against the oracle's restatement of the reference: PolynomialField.NewPolZeroAt
    (r1csqap.go:129-147), Transpose (:11-21), the device-side CombinePolynomials (:191-210), and G1 / G2 Double, Neg, Sub, Equal,
    IsZero (bn128/g1.go:28-30, 91-138, 172-193; g2.go likewise) on points of the x^3 + x + 5 key."""
    from gosnark_amd import bn128
    pf = r1csqap.PolynomialField()
    for pos, tot, h in ((1, 4, 1), (3, 5, 7), (6, 6, O.R - 2)):
        assert pf.NewPolZeroAt(pos, tot, h) == O.PF.NewPolZeroAt(pos, tot, h)
    m = [[1, 2, 3], [4, 5, 6]]
    assert r1csqap.Transpose(m) == O.transpose(m) == [[1, 4], [2, 5], [3, 6]]
    al, be, ga, _ = O.PF.R1CSToQAP(O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)
    w = list(O.X3_WITNESS)
    assert pf.CombinePolynomials(w, al, be, ga) == tuple(O.PF.CombinePolynomials(w, al, be, ga))
    # ragged inputs, as the reference's Add / Mul loop treats them: polynomials of different lengths, more polynomials than
    # witness entries (the reference iterates i < len(r)); too few polynomials / an empty witness are errors, not silent truncation
    rag_a = [[3, 1, 4, 1, 5], [9, 2], [6], [5, 3, 5, 8]]
    rag_b = [[2, 7], [1, 8, 2, 8], [1, 8, 2], [8]]
    rag_c = [[1, 4, 1], [4, 2, 1, 3, 5, 6], [2, 3], [7, 3, 0, 9]]
    rr = [O.R - 5, 11, 0]
    assert pf.CombinePolynomials(rr, rag_a, rag_b, rag_c) == tuple(O.PF.CombinePolynomials(rr, rag_a, rag_b, rag_c))
    with pytest.raises(ValueError):
        pf.CombinePolynomials([1, 2, 3, 4, 5], rag_a, rag_b, rag_c)
    with pytest.raises(ValueError):
        pf.CombinePolynomials([], rag_a, rag_b, rag_c)
    opk = GU.groth_pk(GU.load("groth_x3")["setup"])
    P, Qp = opk.G1_At[2], opk.G1_At[3]
    assert bn128.G1.Double(P) == jac_affine_g1(O.G1.Double(P))
    assert bn128.G1.Sub(P, Qp) == jac_affine_g1(O.G1.Sub(P, Qp))
    assert bn128.G1.Neg(P) == tuple(O.G1.Neg(P))
    assert bn128.G1.Equal(P, bn128.G1.MulScalar(P, 1)) and not bn128.G1.Equal(P, Qp)
    assert bn128.G1.IsZero(bn128.G1.Sub(P, P)) and bn128.G1.Equal(bn128.G1.Sub(P, P), bn128.G1_ZERO)
    P2, Q2 = opk.G2_BACGamma[2], opk.G2_BACGamma[3]
    assert bn128.G2.Double(P2) == jac_affine_g2(O.G2.Double(P2))
    assert bn128.G2.Sub(P2, Q2) == jac_affine_g2(O.G2.Sub(P2, Q2))
    assert bn128.G2.Equal(P2, bn128.G2.Add(bn128.G2.Sub(P2, Q2), Q2)) and bn128.G2.IsZero(bn128.G2.Sub(Q2, Q2))


@pytest.mark.gpu
def test_random_interleavings_of_every_pipelined_operation():
    """tools/soak_mixed.py for a few seconds: Groth16 tickets from px and from the witness (2^16 .. 2^20, uniform and realistic), Pinocchio
    tickets, G1 / G2 MSM tickets from 2^12 to 2^22 terms (both sides of the 3 * 2^20 ticket-stream rule) in RANDOM order with one to three
    in flight -- each result must equal the blocking call's.  Round 4 assigns streams by rule per operation; runs of one kind of operation
    cannot see a buffer or stream handed from one kind to another too early.  (400 s of it: profiles/r04_soak_mixed.txt, 18 729 operations.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_mixed.py"), "10", "3"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "every result equal to its blocking twin" in out.stdout, (out.stdout[-800:], out.stderr[-1500:])
