"""-m gpu: the field / curve primitives of csrc/fp29.h, fq2.h and ec.h, one operation at a time ON THE DEVICE, against the
oracle's restatement of fields/fq.go, fields/fq2.go, bn128/g1.go and bn128/g2.go (oracle/ref_py.py).

The kernels live in tests/device/prim_test.hip (test-only library, built by __graft_entry__.build()).  The reference's own
field vectors (fields/fqn_test.go:22-84) are stated over the toy prime 7, which the fixed-modulus device code cannot run; the
same operand pairs (4,4) (3,4) (5,3) (7,2) (5,11) ... are used here over the real moduli q and r, next to edge operands
(0, 1, p-1, p, p+1, 2^256-1: the ABI accepts any 256-bit value) and random ones, and the oracle -- itself pinned to those
p = 7 vectors by tests/test_oracle_vs_reference.py -- supplies the expected values."""
import ctypes
import os
import random

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi
import gpu_util as U
from oracle import ref_py as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "device", "libgs_prim_test.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        capi.load_library()                       # torch's HIP runtime first (see capi.load_library)
        if not os.path.exists(LIB):
            pytest.fail("%s missing: run __graft_entry__.build()" % LIB)
        _lib = ctypes.CDLL(LIB)
        _lib.gs_prim_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
        _lib.gs_prim_run.restype = ctypes.c_int
    return _lib


def run(kind, op, a_ints, b_ints, words):
    """a_ints / b_ints: per element `words` field elements (flat lists of ints) -> flat list of output ints"""
    a = capi.ints_to_u64(a_ints).reshape(-1)
    b = capi.ints_to_u64(b_ints).reshape(-1)
    n = len(a_ints) // words
    out = np.zeros(n * words * 4, dtype=np.uint64)
    rc = lib().gs_prim_run(kind, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
    assert rc == 0, rc
    return capi.u64_to_ints(out)


def operands(p, seed, count=200):
    rng = random.Random(seed)
    small = [4, 3, 5, 7, 2, 11, 1, 0, 6]                                    # fqn_test.go's operands
    edge = [0, 1, 2, p - 1, p - 2, p, p + 1, 2 * p - 1, 2**255, 2**256 - 1, (p + 1) // 2]
    xs = small + edge + [rng.randrange(p) for _ in range(count)] + [rng.randrange(2**256) for _ in range(20)]
    ys = small[1:] + small[:1] + edge[3:] + edge[:3] + [rng.randrange(p) for _ in range(count)] + [rng.randrange(2**256) for _ in range(20)]
    return xs, ys


@pytest.mark.parametrize("kind,F", [(0, O.FQ), (1, O.FR)])
def test_prime_field_operations_on_the_device(kind, F):
    """fields/fq.go:32-98: Add, Sub, Neg, Double, Mul, Square, Inverse for q (kind 0) and r (kind 1)."""
    p = F.Q
    xs, ys = operands(p, 100 + kind)
    ref = {0: lambda x, y: F.Add(x, y), 1: lambda x, y: F.Sub(x, y), 2: lambda x, y: F.Neg(x), 3: lambda x, y: F.Double(x),
           4: lambda x, y: F.Mul(x, y), 5: lambda x, y: F.Square(x), 6: lambda x, y: F.Inverse(x) if x % p else 0,
           7: lambda x, y: (x * y + x * x) % p, 8: lambda x, y: (3 * x + 2 * y) % p, 9: lambda x, y: 1 if (x - y) % p == 0 else 0}
    for op, f in ref.items():
        got = run(kind, op, xs, ys, 1)
        assert got == [f(x % p, y % p) % p for x, y in zip(xs, ys)], "op %d" % op
    assert run(kind, 9, xs, [x + p if x + p < 2**256 else x for x in xs], 1) == [1] * len(xs)      # x == x + p
    assert run(kind, 10, xs, ys, 1) == [1] * len(xs)                    # interleaved chains (dots2 / dots3 / sqr2) == mul / sqr / mul_add


def test_fq2_operations_on_the_device():
    """fields/fq2.go:37-133 (u^2 = -1): Add, Sub, Neg, Double, Mul, Square, Inverse, and the fused forms the curve code uses."""
    q = O.Q
    xs, ys = operands(q, 300, 120)
    a = [(xs[i], xs[-1 - i]) for i in range(len(xs))]
    b = [(ys[i], ys[-1 - i]) for i in range(len(ys))]
    fa, fb = [c for e in a for c in e], [c for e in b for c in e]
    red = lambda e: (e[0] % q, e[1] % q)                                  # noqa: E731
    F2 = O.FQ2
    ref = {0: lambda x, y: F2.Add(x, y), 1: lambda x, y: F2.Sub(x, y), 2: lambda x, y: F2.Neg(x), 3: lambda x, y: F2.Double(x),
           4: lambda x, y: F2.Mul(x, y), 5: lambda x, y: F2.Square(x),
           6: lambda x, y: F2.Inverse(x) if (x[0] or x[1]) else (0, 0), 7: lambda x, y: (0, 0)}
    for op, f in ref.items():
        got = run(2, op, fa, fb, 2)
        want = [c % q for x, y in zip(a, b) for c in F2.Affine(f(red(x), red(y)))]
        assert got == want, "op %d" % op
    flags = run(2, 8, fa, fb, 2)
    assert flags[0::2] == [1] * len(a)                                      # mul2 / sqr2 (four chains) == mul / sqr


def _g1_cases(rng):
    G = O.G1
    pts = [O.G1_GEN, G.Double(O.G1_GEN), O.G1_ZERO] + [U.rand_g1_jac(rng) for _ in range(40)]
    P = pts + pts[:6] + [pts[4], pts[5]]
    Q = pts[1:] + pts[:1] + pts[:6] + [G.Neg(pts[4]), O.G1_ZERO]             # includes P + P, P + (-P), P + 0, 0 + Q
    return P, Q


def _ref_add(G, zero, p, q):
    if G.IsZero(p):
        return q
    if G.IsZero(q):
        return p
    ap, aq = G.Affine(p), G.Affine(q)
    if ap[0] == aq[0]:
        return G.Double(p) if ap[1] == aq[1] else zero                        # the reference's Add has no P == Q branch (g1.go:32-89)
    return G.Add(p, q)


def test_g1_point_operations_on_the_device():
    """bn128/g1.go:32-170 through csrc/ec.h: mixed addition (with and without negation), doubling, complete addition, scalar
    multiplication, the curve-membership test; complete formulas: P + P, P + (-P) and infinity on either side included."""
    rng = random.Random(77)
    G = O.G1
    P, Q = _g1_cases(rng)
    fp, fq = [c for p in P for c in p], [c for p in Q for c in p]
    aff = lambda p: (lambda a: [0, 0, 0] if a is None else [a[0], a[1], 1])(G.Affine(p))       # noqa: E731
    for op, f in ((0, lambda p, q: _ref_add(G, O.G1_ZERO, p, q)), (1, lambda p, q: _ref_add(G, O.G1_ZERO, p, G.Neg(q))),
                  (2, lambda p, q: G.Double(p) if not G.IsZero(p) else p), (3, lambda p, q: _ref_add(G, O.G1_ZERO, p, q)),
                  (6, lambda p, q: _ref_add(G, O.G1_ZERO, p, q))):            # 6: xyzz_add_mem, the tail kernels' memory-operand addition
        got = run(3, op, fp, fq, 3)
        want = [c for p, q in zip(P, Q) for c in aff(f(p, q))]
        assert got == want, "op %d" % op
    ks = [rng.randrange(O.R) for _ in P]
    ks[:4] = [0, 1, 2, O.R - 1]
    kq = [c for k in ks for c in (k, 0, 0)]
    got = run(3, 4, fp, kq, 3)
    assert got == [c for p, k in zip(P, ks) for c in aff(G.MulScalar(p, k))]
    flags = run(3, 5, fp + [1, 3, 1], fq + [0, 0, 0], 3)
    assert flags[0::3] == [1] * len(P) + [0]                                  # (1, 3) is not on y^2 = x^3 + 3


def test_g2_point_operations_on_the_device():
    """bn128/g2.go:32-200 through csrc/ec.h over Fq2 (four interleaved chains per pair of products)."""
    rng = random.Random(78)
    G = O.G2
    pts = [O.G2_GEN, G.Double(O.G2_GEN), O.G2_ZERO] + [U.rand_g2_jac(rng) for _ in range(16)]
    P = pts + pts[:4] + [pts[4], pts[5]]
    Q = pts[1:] + pts[:1] + pts[:4] + [G.Neg(pts[4]), O.G2_ZERO]
    flat = lambda L: [c for p in L for xy in p for c in xy]                 # noqa: E731
    aff = lambda p: (lambda a: [0] * 6 if a is None else [a[0][0], a[0][1], a[1][0], a[1][1], 1, 0])(G.Affine(p))   # noqa: E731
    for op, f in ((0, lambda p, q: _ref_add(G, O.G2_ZERO, p, q)), (1, lambda p, q: _ref_add(G, O.G2_ZERO, p, G.Neg(q))),
                  (2, lambda p, q: G.Double(p) if not G.IsZero(p) else p), (3, lambda p, q: _ref_add(G, O.G2_ZERO, p, q)),
                  (6, lambda p, q: _ref_add(G, O.G2_ZERO, p, q))):
        got = run(4, op, flat(P), flat(Q), 6)
        want = [c for p, q in zip(P, Q) for c in aff(f(p, q))]
        assert got == want, "op %d" % op
    ks = [rng.randrange(O.R) for _ in P]
    ks[:3] = [0, 1, O.R - 1]
    kq = [c for k in ks for c in (k, 0, 0, 0, 0, 0)]
    got = run(4, 4, flat(P), kq, 6)
    assert got == [c for p, k in zip(P, ks) for c in aff(G.MulScalar(p, k))]
