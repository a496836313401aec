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

    def _lincomb(self, r, polys):
        """sum_i r_i * polys[i] (coefficient-wise, shorter polynomials padded like the reference's Add, r1csqap.go:94-103) with ONE
        device product: the coefficients are packed as F(x) = sum_k sum_i polys[i][k] x^(2 m k + i) and multiplied by
        rev(r)(x) = sum_i r_i x^(m - 1 - i); the blocks of 2 m exponents do not overlap, so the coefficient of x^(2 m k + m - 1) is
        sum_i r_i polys[i][k].  The host only places integers; every field operation runs in gs_poly_mul."""
        m = len(r)
        n = max(len(p) for p in polys[:m])
        stride = 2 * m
        packed = [0] * (stride * n)
        for i in range(m):
            for k, v in enumerate(polys[i]):
                packed[k * stride + i] = v % R
        prod = self.Mul(packed, [r[m - 1 - i] % R for i in range(m)])
        return [prod[k * stride + m - 1] for k in range(n)]

    def CombinePolynomials(self, r, ap, bp, cp):   # r1csqap.go:191-210
        """ax = sum_i r_i ap_i, bx, cx likewise, px = ax * bx - cx, as the reference forms them (it iterates i < len(r) and its Add
        pads the shorter operand) -- five device calls whatever m and n are: one packed product per linear combination (_lincomb),
        one product and one subtraction for px.  (Round 3 recovered the column values with 3 m n blocking gs_poly_eval round trips and
        assumed equal lengths: ADVICE r3.)  No field arithmetic on the host."""
        m = len(r)
        if m == 0:
            raise ValueError("CombinePolynomials: empty witness (the reference's Mul panics on empty operands)")
        for name, polys in (("ap", ap), ("bp", bp), ("cp", cp)):
            if len(polys) < m:
                raise ValueError("CombinePolynomials: len(r) = %d but len(%s) = %d (the reference indexes %s[i] for every i < len(r))"
                                 % (m, name, len(polys), name))
            if any(len(p) == 0 for p in polys[:m]):
                raise ValueError("CombinePolynomials: %s holds an empty polynomial" % name)
        ax, bx, cx = self._lincomb(r, ap), self._lincomb(r, bp), self._lincomb(r, cp)
        px = self.Sub(self.Mul(ax, bx), cx)
        return ax, bx, cx, px


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
