"""Mirror of the reference's bn128.G1 / bn128.G2 seams used by the prover loops
(bn128/g1.go:140 MulScalar + :32 Add; bn128/g2.go:142 / :32), backed by the HIP MSM kernels.
Points are Jacobian tuples of Python ints exactly like the reference's [3]*big.Int /
[3][2]*big.Int; results come back in the affine normal form [x, y, 1]."""
from . import capi

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583      # bn128.go:40-45
G1_ZERO = (0, 0, 0)
G2_ZERO = ((0, 0), (0, 0), (0, 0))


class _Group:
    def __init__(self, g2):
        self.g2 = g2

    def _upload(self, pts):
        return capi.g2_upload(capi.g2_points_to_u64(pts)) if self.g2 else capi.g1_upload(capi.g1_points_to_u64(pts))

    def _jac(self, aff):
        if aff is None:
            return G2_ZERO if self.g2 else G1_ZERO
        return (aff[0], aff[1], (1, 0)) if self.g2 else (aff[0], aff[1], 1)

    def MSM(self, points, scalars):
        """sum_i MulScalar(points[i], scalars[i]) -- the loop of groth16.go:243-250."""
        if not points:
            return self._jac(None)
        h = self._upload(points)
        return self._jac(capi.msm(h, capi.ints_to_u64(scalars), g2=self.g2))

    def MulScalar(self, p, e):               # g1.go:140-155 / g2.go:142-181
        return self.MSM([p], [abs(e)])

    def Add(self, p1, p2):                   # g1.go:32-89 / g2.go:32-89 (complete here)
        return self.MSM([p1, p2], [1, 1])

    def Affine(self, p):                     # g1.go:157-170 / g2.go:183-200
        q = self.MSM([p], [1])
        return None if q == self._jac(None) else (q[0], q[1])

    def IsZero(self, p):                     # g1.go:28-30 / g2.go:28-30: only Z is tested
        return p[2] == (0, 0) if self.g2 else p[2] == 0

    def Neg(self, p):                        # g1.go:91-96 / g2.go:91-97: (X, -Y, Z) -- a sign flip, no field product: host side
        if self.g2:
            return (p[0], ((-p[1][0]) % Q, (-p[1][1]) % Q), p[2])
        return (p[0], (-p[1]) % Q, p[2])

    def Sub(self, a, b):                     # g1.go:98-100 / g2.go:99-101
        return self.Add(a, self.Neg(b))

    def Double(self, p):                     # g1.go:101-138 / g2.go:103-140: the complete addition doubles (device: xyzz_dbl_affine)
        return self.Add(p, p)

    def Equal(self, p1, p2):                 # g1.go:172-193 / g2.go:202-223, on the affine normal form
        if self.IsZero(p1) or self.IsZero(p2):
            return self.IsZero(p1) and self.IsZero(p2)
        return self.Affine(p1) == self.Affine(p2)


G1 = _Group(False)
G2 = _Group(True)


def Pairing(p1, p2):
    """bn128.Pairing(p1, p2) (bn128.go:179-186): the reduced optimal ate pairing as the reference's nested
    [2][3][2] integers (gs_pairing, host side)."""
    import numpy as np
    out = np.zeros(48, dtype=np.uint64)
    a, b = capi.g1_points_to_u64([p1]), capi.g2_points_to_u64([p2])
    capi.check(capi.load_library().gs_pairing(capi.ptr64(a), capi.ptr64(b), capi.ptr64(out)))
    v = capi.u64_to_ints(out)
    return tuple(tuple((v[6 * i + 2 * j], v[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2))


def PairingCheck(g1_points, g2_points):
    """prod_i e(g1_i, g2_i) == 1 with one shared final exponentiation (gs_pairing_check)."""
    import ctypes
    import numpy as np
    if len(g1_points) != len(g2_points):
        raise ValueError("PairingCheck: %d G1 points, %d G2 points" % (len(g1_points), len(g2_points)))
    a = capi.g1_points_to_u64(g1_points) if g1_points else np.zeros((1, 12), dtype=np.uint64)
    b = capi.g2_points_to_u64(g2_points) if g2_points else np.zeros((1, 24), dtype=np.uint64)
    ok = ctypes.c_int(0)
    capi.check(capi.load_library().gs_pairing_check(capi.ptr64(a), capi.ptr64(b), len(g1_points), ctypes.byref(ok)))
    return bool(ok.value)
