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
