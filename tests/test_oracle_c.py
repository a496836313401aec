"""Pins the C oracle (oracle/gs_oracle.c) bit-for-bit (raw Jacobian) against the Python oracle,
which is itself pinned against the reference's compiled prover (test_oracle_vs_reference.py),
and directly against the reference wasm goldens."""
import random

import numpy as np
import pytest

import golden_util as GU
from oracle import c_oracle as C
from oracle import ref_py as O


def _g1_arr(points):
    return np.frombuffer(b"".join(int(c).to_bytes(32, "little") for p in points for c in p), dtype="<u8").copy()


def _g2_arr(points):
    return np.frombuffer(b"".join(int(c).to_bytes(32, "little") for p in points for xy in p for c in xy), dtype="<u8").copy()


def _sc_arr(ks):
    return np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype="<u8").copy()


def test_c_g1_loop_equals_python_jacobian():
    rng = random.Random(3)
    pts = [O.G1.MulScalar(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(12)] + [O.G1_ZERO]
    ks = [rng.randrange(O.R) for _ in range(11)] + [0, 5]
    acc = O.G1_ZERO
    for p, k in zip(pts, ks):
        acc = O.G1.Add(acc, O.G1.MulScalar(p, k))
    assert C.g1_msm_naive(_g1_arr(pts), _sc_arr(ks)) == acc
    assert C.g1_affine(acc) == O.G1.Affine(acc)


def test_c_g2_loop_equals_python_jacobian():
    rng = random.Random(4)
    pts = [O.G2.MulScalar(O.G2_GEN, rng.randrange(1, O.R)) for _ in range(6)] + [O.G2_ZERO]
    ks = [rng.randrange(O.R) for _ in range(6)] + [9]
    acc = O.G2_ZERO
    for p, k in zip(pts, ks):
        acc = O.G2.Add(acc, O.G2.MulScalar(p, k))
    assert C.g2_msm_naive(_g2_arr(pts), _sc_arr(ks)) == acc
    assert C.g2_affine(acc) == O.G2.Affine(acc)


@pytest.mark.parametrize("name", ["groth_rand_m17", "groth_x3"])
def test_c_loops_reproduce_reference_wasm_msm_parts(name):
    """PiA before the alpha/delta tail = the first loop of groth16.go:243-247; checked through
    the full proof: ref_py (pinned to the wasm) and the C loops must agree on every MSM."""
    rec = GU.load(name)
    pk = GU.groth_pk(rec["setup"])
    w = rec["w"]
    for pts in (pk.G1_At, pk.G1_BACGamma):
        acc = O.G1_ZERO
        for p, k in zip(pts, w):
            acc = O.G1.Add(acc, O.G1.MulScalar(p, k))
        assert C.g1_msm_naive(_g1_arr(pts), _sc_arr(w)) == acc
    acc = O.G2_ZERO
    for p, k in zip(pk.G2_BACGamma, w):
        acc = O.G2.Add(acc, O.G2.MulScalar(p, k))
    assert C.g2_msm_naive(_g2_arr(pk.G2_BACGamma), _sc_arr(w)) == acc


def test_c_multithreaded_baseline_same_point():
    rng = random.Random(6)
    pts = [O.G1.MulScalar(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(16)]
    ks = [rng.randrange(O.R) for _ in range(16)]
    one = C.g1_msm_naive(_g1_arr(pts), _sc_arr(ks), threads=1)
    four = C.g1_msm_naive(_g1_arr(pts), _sc_arr(ks), threads=4)
    assert O.G1.Equal(one, four)


def test_c_poly_ops_equal_python_and_reference_vectors():
    rng = random.Random(8)
    assert C.poly_mul([1, 0, 5], [3, 0, 1]) == [3, 0, 16, 0, 5]            # r1csqap_test.go:59-62
    q, r = C.poly_div([3, 0, 16, 0, 5], [3, 0, 1])
    assert q == [1, 0, 5] and r == [0, 0]                                  # :64-67
    a = [rng.randrange(O.R) for _ in range(23)]
    b = [rng.randrange(O.R) for _ in range(9)]
    assert C.poly_mul(a, b) == O.PF.Mul(a, b)
    q, r = C.poly_div(a, b)
    pq, pr = O.PF.Div(a, b)
    assert q == pq and r == pr[:len(b) - 1]
    v = [rng.randrange(O.R) for _ in range(7)]
    assert C.lagrange(v) == O.PF.LagrangeInterpolation(v)
    assert C.lagrange([0, 0, 0, 5]) == O.PF.LagrangeInterpolation([0, 0, 0, 5])   # :107
    x = rng.randrange(O.R)
    assert C.poly_eval(a, x) == O.PF.Eval(a, x)
    v30 = [rng.randrange(O.R) for _ in range(30)]                          # beyond the Go-int overflow
    assert C.lagrange(v30) == O.PF.LagrangeInterpolation(v30)


def test_pinocchio_fixture_div_matches():
    rec = GU.load("pinocchio_x3_fixture")
    q, r = C.poly_div(rec["px"], GU.pinocchio_pk(rec["setup"]).Z)
    assert q == O.PF.DivisorPolynomial(rec["px"], GU.pinocchio_pk(rec["setup"]).Z) and all(x == 0 for x in r)
