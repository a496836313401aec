"""Mirror of the reference's r1csqap.PolynomialField (r1csqap/r1csqap.go:45-216) on the HIP kernels.
Polynomials are lists of Python ints (coefficients mod r, lowest degree first), like []*big.Int."""
import ctypes

import numpy as np

from . import capi

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _arr(v):
    return capi.ints_to_u64([x % R for x in v]) if len(v) else np.zeros((0, 4), dtype=np.uint64)


class PolynomialField:
    """r1csqap.go:45-54.  `F` is kept for signature parity; the modulus is BN128's r."""

    def __init__(self, f=None):
        self.F = f

    def Mul(self, a, b):                     # r1csqap.go:57-67
        capi.init()
        out = np.zeros((len(a) + len(b) - 1, 4), dtype=np.uint64)
        capi.check(capi.load_library().gs_poly_mul(capi.ptr64(_arr(a)), len(a), capi.ptr64(_arr(b)), len(b), capi.ptr64(out)))
        return capi.u64_to_ints(out)

    def Div(self, a, b):                     # r1csqap.go:70-84 -> (quotient, remainder)
        capi.init()
        nq, nr = len(a) - len(b) + 1, len(b) - 1
        q = np.zeros((nq, 4), dtype=np.uint64)
        r = np.zeros((max(nr, 1), 4), dtype=np.uint64)
        capi.check(capi.load_library().gs_poly_div(capi.ptr64(_arr(a)), len(a), capi.ptr64(_arr(b)), len(b), capi.ptr64(q), capi.ptr64(r)))
        return capi.u64_to_ints(q), capi.u64_to_ints(r)[:nr]

    def _addsub(self, a, b, name):
        capi.init()
        n = max(len(a), len(b))
        out = np.zeros((n, 4), dtype=np.uint64)
        capi.check(getattr(capi.load_library(), name)(capi.ptr64(_arr(a)), len(a), capi.ptr64(_arr(b)), len(b), capi.ptr64(out)))
        return capi.u64_to_ints(out)

    def Add(self, a, b):                     # r1csqap.go:94-103
        return self._addsub(a, b, "gs_poly_add")

    def Sub(self, a, b):                     # r1csqap.go:106-115
        return self._addsub(a, b, "gs_poly_sub")

    def Eval(self, v, x):                    # r1csqap.go:118-126
        capi.init()
        out = np.zeros(4, dtype=np.uint64)
        capi.check(capi.load_library().gs_poly_eval(capi.ptr64(_arr(v)), len(v), capi.ptr64(_arr([x])), capi.ptr64(out)))
        return capi.u64_to_ints(out)[0]

    def LagrangeInterpolation(self, v):      # r1csqap.go:150-158 (nodes 1..len(v))
        capi.init()
        out = np.zeros((len(v), 4), dtype=np.uint64)
        capi.check(capi.load_library().gs_lagrange_interpolation(capi.ptr64(_arr(v)), len(v), capi.ptr64(out)))
        return capi.u64_to_ints(out)

    def DivisorPolynomial(self, px, z):      # r1csqap.go:213-216
        return self.Div(px, z)[0]


def ZPoly(deg):
    """Z(x) = prod_{i=1}^{deg} (x - i)  (r1csqap.go:177-186 / groth16.go:122-131)."""
    return capi.u64_to_ints(capi.zpoly(deg))
