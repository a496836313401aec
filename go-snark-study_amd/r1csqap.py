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

    def R1CSToQAP(self, a, b, c):            # r1csqap.go:161-188 (dense n x m matrices, small instances)
        """-> (alphas, betas, gammas, z): per variable the interpolant of its column over the nodes 1..n."""
        def cols(mat):
            n, m = len(mat), len(mat[0])
            return [self.LagrangeInterpolation([mat[j][i] for j in range(n)]) for i in range(m)]
        alphas, betas, gammas = cols(a), cols(b), cols(c)
        return alphas, betas, gammas, ZPoly(len(alphas) - 2)      # r1csqap.go:177-186: degree len(alphas) - 2

    def NewPolZeroAt(self, pointPos, totalPoints, height):   # r1csqap.go:129-147
        """The polynomial of degree totalPoints - 1 that is `height` at node pointPos and 0 at the other nodes of 1..totalPoints:
        the interpolant of height * e_pointPos.  Exact for every n (the reference's int64 factorials overflow from n = 22 on,
        r1csqap.go:130-136; equal results for n <= 21)."""
        v = [0] * totalPoints
        v[pointPos - 1] = height % R
        return self.LagrangeInterpolation(v)

    def CombinePolynomials(self, r, ap, bp, cp):   # r1csqap.go:191-210
        """ax = sum_i r_i ap_i, bx, cx likewise, px = ax * bx - cx -- on the device: ax is the interpolant of the values (A r)_j at the
        nodes 1..n, so the dense polynomials are turned back into their column values (gs_poly_eval at the nodes: host loop over
        m n evaluations, device arithmetic) and gs_r1cs_to_px does the linear combinations, the interpolations and the product.
        No field arithmetic on the host."""
        n = len(ap[0])

        def rows(polys):
            vals = [[self.Eval(p, j) for p in polys] for j in range(1, n + 1)]        # vals[j-1][i] = polys[i](j)
            return csr_from_rows([{i: v for i, v in enumerate(row) if v} for row in vals])
        ax, bx, cx, px = ComputePx(rows(ap), rows(bp), rows(cp), capi.ints_to_u64([x % R for x in r]), len(ap))
        return capi.u64_to_ints(ax), capi.u64_to_ints(bx), capi.u64_to_ints(cx), capi.u64_to_ints(px)


def Transpose(matrix):                       # r1csqap.go:11-21
    return [list(col) for col in zip(*matrix)]


def csr_from_rows(rows):
    """rows: list (one per constraint) of {variable index: coefficient} -> (row_ptr uint32, col uint32, val [nnz,4] uint64)."""
    rowptr = np.zeros(len(rows) + 1, dtype=np.uint32)
    cols, vals = [], []
    for j, row in enumerate(rows):
        for k in sorted(row):
            cols.append(k)
            vals.append(row[k] % R)
        rowptr[j + 1] = len(cols)
    col = np.asarray(cols, dtype=np.uint32) if cols else np.zeros(0, dtype=np.uint32)
    val = capi.ints_to_u64(vals) if vals else np.zeros((0, 4), dtype=np.uint64)
    return rowptr, col, val


def ComputePx(a_csr, b_csr, c_csr, w_u64, nvars):
    """Sparse R1CS (three CSR triples over n constraints x nvars variables) + witness ([nvars,4] uint64) ->
    (ax, bx, cx, px) as uint64 limb arrays: the scalable form of R1CSToQAP + CombinePolynomials (gs_r1cs_to_px)."""
    capi.init()
    n = a_csr[0].shape[0] - 1
    w = np.ascontiguousarray(w_u64, dtype=np.uint64).reshape(-1, 4)
    assert w.shape[0] == nvars
    out = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)] + [np.zeros((2 * n - 1, 4), dtype=np.uint64)]
    args = []
    for rp, cl, vl in (a_csr, b_csr, c_csr):
        rp = np.ascontiguousarray(rp, dtype=np.uint32)
        cl = np.ascontiguousarray(cl, dtype=np.uint32)
        vl = np.ascontiguousarray(vl, dtype=np.uint64).reshape(-1, 4)
        if cl.size == 0:
            cl, vl = np.zeros(1, dtype=np.uint32), np.zeros((1, 4), dtype=np.uint64)
        args += [rp, cl, vl]
    capi.check(capi.load_library().gs_r1cs_to_px(
        n, nvars, capi.ptr32(args[0]), capi.ptr32(args[1]), capi.ptr64(args[2]), capi.ptr32(args[3]), capi.ptr32(args[4]), capi.ptr64(args[5]),
        capi.ptr32(args[6]), capi.ptr32(args[7]), capi.ptr64(args[8]), capi.ptr64(w),
        capi.ptr64(out[0]), capi.ptr64(out[1]), capi.ptr64(out[2]), capi.ptr64(out[3])))
    return tuple(out)


def ZPoly(deg):
    """Z(x) = prod_{i=1}^{deg} (x - i)  (r1csqap.go:177-186 / groth16.go:122-131)."""
    return capi.u64_to_ints(capi.zpoly(deg))


def _csr_args(csrs):
    args = []
    for rp, cl, vl in csrs:
        rp = np.ascontiguousarray(rp, dtype=np.uint32)
        cl = np.ascontiguousarray(cl, dtype=np.uint32)
        vl = np.ascontiguousarray(vl, dtype=np.uint64).reshape(-1, 4)
        if cl.size == 0:
            cl, vl = np.zeros(1, dtype=np.uint32), np.zeros((1, 4), dtype=np.uint64)
        args += [rp, cl, vl]
    return args


class DeviceR1CS:
    """A sparse R1CS resident on the device (gs_r1cs_upload): upload once per circuit, then ComputePxResident per proof."""

    def __init__(self, a_csr, b_csr, c_csr, nvars):
        capi.init()
        self.n, self.nvars = a_csr[0].shape[0] - 1, nvars
        a = _csr_args((a_csr, b_csr, c_csr))
        h = capi.Handle(0)
        capi.check(capi.load_library().gs_r1cs_upload(self.n, nvars, capi.ptr32(a[0]), capi.ptr32(a[1]), capi.ptr64(a[2]), capi.ptr32(a[3]),
                                                      capi.ptr32(a[4]), capi.ptr64(a[5]), capi.ptr32(a[6]), capi.ptr32(a[7]), capi.ptr64(a[8]),
                                                      ctypes.byref(h)))
        self.handle = capi.DeviceHandle(h.value)

    def ComputePxResident(self, w_handle, px_handle=None):
        """resident w (capi.scalars_upload) -> resident px (a capi.DeviceHandle; pass the previous one to overwrite it)."""
        h = capi.Handle(px_handle.h if px_handle is not None else 0)
        capi.check(capi.load_library().gs_r1cs_px(capi.Handle(self.handle.h), capi.Handle(w_handle.h), ctypes.byref(h)))
        return px_handle if px_handle is not None else capi.DeviceHandle(h.value)
