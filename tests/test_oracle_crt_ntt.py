"""oracle/crt_ntt.py (the exact multi-prime product the large golden generator uses) against the C restatement of the reference's
schoolbook PolynomialField.Mul (r1csqap.go:57-67): test infrastructure checking test infrastructure, no device."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import crt_ntt
from oracle import ref_py as O


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    vals = [int.from_bytes(rng.bytes(32), "little") % O.R for _ in range(n)]
    return C.poly_u64(vals)


@pytest.mark.parametrize("na,nb", [(1, 1), (2, 3), (17, 5), (64, 64), (300, 257), (1000, 1)])
def test_crt_ntt_product_equals_schoolbook(na, nb):
    a, b = _rand(na, 7 * na + nb), _rand(nb, 11 * nb + na)
    assert np.array_equal(crt_ntt.poly_mul_mod_r(a, b), C.poly_mul_u64(a, b))


def test_crt_ntt_extreme_coefficients():
    a = C.poly_u64([O.R - 1] * 40)
    b = C.poly_u64([O.R - 1, 0, O.R - 1, 1, O.R - 2] * 9)
    assert np.array_equal(crt_ntt.poly_mul_mod_r(a, b), C.poly_mul_u64(a, b))
    assert len(crt_ntt.PRIMES) == 18 and all(p < 2**31 and (p - 1) % (1 << 22) == 0 for p in crt_ntt.PRIMES)
    prod = 1
    for p in crt_ntt.PRIMES:
        prod *= p
    assert prod > (1 << 20) * O.R * O.R


def test_golden_generator_draws_the_same_scalars_as_the_product_side_instances():
    """oracle/gen_golden_large.py keeps its own copy of the seeded scalar generator (the oracle must not import the product package);
    the goldens only mean something if it is bit-identical to gosnark_amd.synth.scalars_u64, which the GPU tests use to rebuild the
    instances."""
    import importlib.util
    import os
    import sys
    import gosnark_amd  # noqa: F401
    from gosnark_amd import synth
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "gen_golden_large.py")
    spec = importlib.util.spec_from_file_location("gen_golden_large", path)
    gen = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["gen_golden_large.py", "none"]
    try:
        spec.loader.exec_module(gen)
    finally:
        sys.argv = argv
    for n, seed in ((1, 0), (7, 0x60D0), (1000, 0x60D5 + 11), (4097, 24789)):
        a, b = gen.scalars_u64(n, seed), synth.scalars_u64(n, seed)
        assert a.dtype == b.dtype and np.array_equal(a, b)
        assert all(v < O.R for v in C._ints(a))
