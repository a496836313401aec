"""the verifier (groth16.VerifyProof groth16.go:281-305, snark.VerifyProof snark.go:292-368) and the
pairing under it (bn128.go:179-421).  Host code: these tests run without a GPU.

Anchors: (1) the accept/reject answers the reference's own compiled wasm verifier gave on the recorded instances
(tests/golden/wasm_*.json "verify"); (2) the oracle's line-by-line restatement of bn128.Pairing
(oracle/ref_pairing.py), itself pinned by (1), whose Fq12 VALUE the product must reproduce bit for bit even though it
gets there by a different algorithm; (3) bilinearity and non-degeneracy."""
import json
import os
import random

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import bn128, capi, groth16, snark, utils
from oracle import ref_pairing as RP
from oracle import ref_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rec(name):
    with open(os.path.join(GOLDEN, "wasm_%s.json" % name)) as f:
        r = json.load(f)
    return r, json.loads(r["setup"]), json.loads(r["proof"])


def recorded_answers(r):
    return [(json.loads(v["public"]), v["result"] == "true") for v in r["verify"]]


def test_pairing_value_equals_the_reference_restatement_bit_for_bit():
    rng = random.Random(77)
    cases = [(O.G1_GEN, O.G2_GEN)]
    for _ in range(2):
        a, b = rng.randrange(1, O.R), rng.randrange(1, O.R)
        cases.append((O.G1.MulScalar(O.G1_GEN, a), O.G2.MulScalar(O.G2_GEN, b)))      # raw Jacobian triples, Z != 1
    for p, q in cases:
        assert bn128.Pairing(p, q) == RP.Pairing(p, q)


def test_pairing_is_bilinear_and_non_degenerate():
    rng = random.Random(78)
    a, b = rng.randrange(1, O.R), rng.randrange(1, O.R)
    aP, bQ = O.G1.MulScalar(O.G1_GEN, a), O.G2.MulScalar(O.G2_GEN, b)
    abP = O.G1.MulScalar(O.G1_GEN, a * b % O.R)
    assert bn128.PairingCheck([aP, O.G1.Neg(abP)], [bQ, O.G2_GEN])          # e(aP, bQ) = e(abP, Q)
    assert bn128.PairingCheck([aP, O.G1.Neg(O.G1_GEN)], [O.G2_GEN, O.G2.MulScalar(O.G2_GEN, a)])   # e(aP, Q) = e(P, aQ)
    assert not bn128.PairingCheck([O.G1_GEN], [O.G2_GEN])                   # e(P, Q) != 1
    assert not bn128.PairingCheck([aP, O.G1.Neg(abP)], [bQ, O.G2.MulScalar(O.G2_GEN, 2)])
    assert bn128.PairingCheck([], [])                                       # empty product
    assert bn128.PairingCheck([O.G1_ZERO, aP], [bQ, O.G2_ZERO])             # infinity on either side contributes 1
    one = bn128.Pairing(O.G1_ZERO, O.G2_GEN)
    assert one == (((1, 0), (0, 0), (0, 0)), ((0, 0), (0, 0), (0, 0)))
    # e(P, Q)^r = 1 through the oracle's Fq12 arithmetic on the product's value
    assert RP.FQ12.Equal(RP.FQ12.Exp(bn128.Pairing(O.G1_GEN, O.G2_GEN), O.R), RP.FQ12.One())


def test_off_curve_points_are_rejected():
    bad_g1 = (1, 3, 1)
    assert not bn128.PairingCheck([bad_g1], [O.G2_GEN])
    bad_g2 = ((1, 0), (2, 0), (1, 0))
    assert not bn128.PairingCheck([O.G1_GEN], [bad_g2])
    with pytest.raises(RuntimeError, match="not on the curve"):
        bn128.Pairing(bad_g1, O.G2_GEN)


def test_groth16_verifier_gives_the_reference_verifiers_answers():
    r, setup, proof = rec("groth_x3")
    _, vk = utils.GrothSetupFromString(setup)
    pr = utils.GrothProofFromString(proof)
    answers = recorded_answers(r)
    assert answers == [([35], True), ([34], False)]
    for public, want in answers:
        assert groth16.VerifyProof(vk, pr, public) is want
    # the oracle's restatement of the reference verifier agrees (this is what pins oracle/ref_pairing.py)
    ovk = O.GrothVk()
    ovk.IC, ovk.G1_Alpha, ovk.G2_Beta, ovk.G2_Gamma, ovk.G2_Delta = vk.IC, vk.G1_Alpha, vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta
    assert RP.groth16_VerifyProof(ovk, (pr.PiA, pr.PiB, pr.PiC), [35]) is True
    assert RP.groth16_VerifyProof(ovk, (pr.PiA, pr.PiB, pr.PiC), [34]) is False


def test_groth16_verifier_rejects_tampering_and_accepts_rescaled_jacobians():
    _, setup, proof = rec("groth_x3")
    _, vk = utils.GrothSetupFromString(setup)
    pr = utils.GrothProofFromString(proof)
    assert groth16.VerifyProof(vk, pr, [35])
    two_a = O.G1.Double(pr.PiA)
    assert not groth16.VerifyProof(vk, groth16.Proof(two_a, pr.PiB, pr.PiC), [35])
    assert not groth16.VerifyProof(vk, groth16.Proof(pr.PiA, O.G2.Double(pr.PiB), pr.PiC), [35])
    assert not groth16.VerifyProof(vk, groth16.Proof(pr.PiC, pr.PiB, pr.PiA), [35])
    assert not groth16.VerifyProof(vk, groth16.Proof(O.G1_ZERO, pr.PiB, pr.PiC), [35])
    assert not groth16.VerifyProof(vk, groth16.Proof((pr.PiA[0], pr.PiA[1] ^ 1, pr.PiA[2]), pr.PiB, pr.PiC), [35])   # off the curve
    # the same points as other Jacobian representatives (X l^2, Y l^3, Z l)
    lam = 0x1234567890ABCDEF1234567
    resc = lambda p: (p[0] * lam * lam % O.Q, p[1] * lam ** 3 % O.Q, p[2] * lam % O.Q)   # noqa: E731
    l2 = (5, 7)
    r2 = lambda p: (O.FQ2.Mul(p[0], O.FQ2.Square(l2)), O.FQ2.Mul(p[1], O.FQ2.Mul(l2, O.FQ2.Square(l2))), O.FQ2.Mul(p[2], l2))   # noqa: E731
    assert groth16.VerifyProof(vk, groth16.Proof(resc(pr.PiA), r2(pr.PiB), resc(pr.PiC)), [35])
    # public signal >= r is the same scalar mod r (MulScalar on an order-r point)
    assert groth16.VerifyProof(vk, pr, [35 + O.R])
    # fewer signals than IC points is allowed by the reference's loop; more panics there and raises here
    assert not groth16.VerifyProof(vk, pr, [])
    with pytest.raises(IndexError):
        groth16.VerifyProof(vk, pr, [35, 1])
    ok = capi.ctypes.c_int(0)
    z12, z24, z4 = np.zeros(12, dtype=np.uint64), np.zeros(24, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    st = capi.load_library().gs_groth16_verify(capi.ptr64(z12), capi.ptr64(z24), capi.ptr64(z24), capi.ptr64(z24), capi.ptr64(z12), 1,
                                               capi.ptr64(z4), 2, capi.ptr64(z12), capi.ptr64(z24), capi.ptr64(z12), capi.ctypes.byref(ok))
    assert st == -4 and b"IC points" in capi.load_library().gs_last_error()


@pytest.mark.parametrize("name", ["pinocchio_x3_setup", "pinocchio_x3_fixture"])
def test_pinocchio_verifier_gives_the_reference_verifiers_answers(name):
    r, setup, proof = rec(name)
    _, vk = utils.SetupFromString(setup)
    pr = utils.ProofFromString(proof)
    answers = recorded_answers(r)
    assert answers == [([35], True), ([34], False)]
    for public, want in answers:
        assert snark.VerifyProof(vk, pr, public) is want


def test_pinocchio_oracle_restatement_agrees_and_failed_check_is_reported(capsys):
    _, setup, proof = rec("pinocchio_x3_setup")
    _, vk = utils.SetupFromString(setup)
    pr = utils.ProofFromString(proof)
    ovk = O.PinocchioVk()
    for k in snark.Vk.FIELDS:
        setattr(ovk, k, getattr(vk, k))
    ovk.IC = vk.IC
    pd = {k: getattr(pr, k) for k in snark.Proof.FIELDS}
    assert RP.snark_VerifyProof(ovk, pd, [35]) == (True, 0)
    assert RP.snark_VerifyProof(ovk, pd, [34]) == (False, 4)       # the wrong public input breaks the divisibility check
    assert snark.VerifyProof(vk, pr, [34], debug=True) is False
    out = capsys.readouterr().out
    assert out.count("✓") == 3 and "❌ e(Vkx+piA, piB)" in out

    def first_bad(**change):
        fields = dict(pd)
        fields.update(change)
        words = np.concatenate([capi.g1_points_to_u64([fields["PiA"], fields["PiAp"]]).reshape(-1), capi.g2_points_to_u64([fields["PiB"]]).reshape(-1),
                                capi.g1_points_to_u64([fields[k] for k in ("PiBp", "PiC", "PiCp", "PiH", "PiKp")]).reshape(-1)])
        g1 = capi.g1_points_to_u64([vk.Vkb, vk.G1Kbg])
        g2 = capi.g2_points_to_u64([vk.Vka, vk.Vkc, vk.G2Kbg, vk.G2Kg, vk.Vkz])
        ic, pub = capi.g1_points_to_u64(vk.IC), capi.ints_to_u64([35])
        ok, bad = capi.ctypes.c_int(0), capi.ctypes.c_int(0)
        capi.check(capi.load_library().gs_pinocchio_verify(
            capi.ptr64(g2[0]), capi.ptr64(g1[0]), capi.ptr64(g2[1]), capi.ptr64(g1[1]), capi.ptr64(g2[2]), capi.ptr64(g2[3]), capi.ptr64(g2[4]),
            capi.ptr64(ic), len(vk.IC), capi.ptr64(pub), 1, capi.ptr64(np.ascontiguousarray(words)), capi.ctypes.byref(ok), capi.ctypes.byref(bad)))
        return ok.value, bad.value
    assert first_bad() == (1, 0)
    assert first_bad(PiAp=O.G1.Double(pd["PiAp"])) == (0, 1)
    assert first_bad(PiBp=O.G1.Double(pd["PiBp"])) == (0, 2)
    assert first_bad(PiCp=O.G1.Double(pd["PiCp"])) == (0, 3)
    assert first_bad(PiH=O.G1.Double(pd["PiH"])) == (0, 4)
    assert first_bad(PiKp=O.G1.Double(pd["PiKp"])) == (0, 5)


def test_verifier_needs_no_device_and_no_init():
    """The verifier entry points are host code by design (O(1) pairings): they work before gs_init and on a box
    without a GPU -- unlike every prover entry point, which must fail loudly there."""
    lib = capi.load_library()
    _, setup, proof = rec("groth_x3")
    _, vk = utils.GrothSetupFromString(setup)
    assert groth16.VerifyProof(vk, utils.GrothProofFromString(proof), [35])
    h = capi.Handle(0)
    z = np.zeros((1, 12), dtype=np.uint64)
    import torch
    if not torch.cuda.is_available():
        assert lib.gs_g1_upload(capi.ptr64(z), 1, capi.ctypes.byref(h)) < 0       # prover side still refuses without a device


def test_g2_points_outside_the_order_r_subgroup_are_rejected():
    """E'(Fq2) has cofactor 2q - r: a point on the twist need not be in G2.  Such a PiB must never verify (the reference
    would run its Miller loop on it and compare whatever comes out)."""
    # find a twist point by trying x = (k, 1): y^2 = x^3 + 3/(9+u); a random twist point is outside G2 with overwhelming probability
    b_twist = O.FQ2.Mul(O.FQ2.Inverse((9, 1)), (3, 0))
    q = O.Q
    found = None
    k = 1
    while found is None:
        x = (k, 1)
        rhs = O.FQ2.Add(O.FQ2.Mul(O.FQ2.Square(x), x), b_twist)
        # square root in Fq2 via the norm trick (q = 3 mod 4)
        a0, a1 = rhs
        norm = (a0 * a0 + a1 * a1) % q
        s = pow(norm, (q + 1) // 4, q)
        if s * s % q == norm:
            for sgn in (s, q - s):
                t = (a0 + sgn) * pow(2, q - 2, q) % q
                y0 = pow(t, (q + 1) // 4, q)
                if y0 * y0 % q == t and y0:
                    y1 = a1 * pow(2 * y0, q - 2, q) % q
                    if O.FQ2.Square((y0, y1)) == rhs:
                        found = (x, (y0, y1), (1, 0))
                        break
        k += 1
    on_twist = found
    assert O.G2.Affine(O.G2.MulScalar(on_twist, O.R)) is not None          # [r]Q != infinity: not in G2
    assert not bn128.PairingCheck([O.G1_GEN, O.G1.Neg(O.G1_GEN)], [on_twist, on_twist])    # even e(P,Q) e(-P,Q), trivially 1 in G2
    in_g2 = O.G2.MulScalar(O.G2_GEN, 123456789)
    assert bn128.PairingCheck([O.G1_GEN, O.G1.Neg(O.G1_GEN)], [in_g2, in_g2])
    # cofactor-cleared: [2q - r] Q lands in G2 and is accepted again
    cleared = O.G2.MulScalar(on_twist, 2 * q - O.R)
    assert O.G2.Affine(O.G2.MulScalar(cleared, O.R)) is None
    assert bn128.PairingCheck([O.G1_GEN, O.G1.Neg(O.G1_GEN)], [cleared, cleared])
    _, setup, proof = rec("groth_x3")
    _, vk = utils.GrothSetupFromString(setup)
    pr = utils.GrothProofFromString(proof)
    assert groth16.VerifyProof(vk, pr, [35]) is True
    assert groth16.VerifyProof(vk, groth16.Proof(pr.PiA, on_twist, pr.PiC), [35]) is False
