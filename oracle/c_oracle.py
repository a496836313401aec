"""ctypes loader for oracle/libgs_oracle.so (C restatement of the reference's naive prover
arithmetic, see gs_oracle.c).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
u64p = ctypes.POINTER(ctypes.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgs_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "gs_oracle.c")):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_g1_affine.restype = ctypes.c_int
        _LIB.oracle_g2_affine.restype = ctypes.c_int
    return _LIB


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _ints(arr):
    raw = np.ascontiguousarray(arr, dtype="<u8").tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def _u64(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype="<u8").copy()


def g1_msm_naive(pts_u64, scalars_u64, threads=1):
    """literal reference loop on [n,12] Jacobian points, [n,4] scalars -> Jacobian (X,Y,Z) ints"""
    pts = np.ascontiguousarray(pts_u64, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars_u64, dtype=np.uint64)
    n = sc.size // 4
    out = np.zeros(12, dtype=np.uint64)
    if threads <= 1:
        lib().oracle_g1_msm_naive(_p(pts), _p(sc), ctypes.c_size_t(n), _p(out))
    else:
        lib().oracle_msm_naive_mt(_p(pts), _p(sc), ctypes.c_size_t(n), 0, int(threads), _p(out))
    return tuple(_ints(out))


def g2_msm_naive(pts_u64, scalars_u64, threads=1):
    pts = np.ascontiguousarray(pts_u64, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars_u64, dtype=np.uint64)
    n = sc.size // 4
    out = np.zeros(24, dtype=np.uint64)
    if threads <= 1:
        lib().oracle_g2_msm_naive(_p(pts), _p(sc), ctypes.c_size_t(n), _p(out))
    else:
        lib().oracle_msm_naive_mt(_p(pts), _p(sc), ctypes.c_size_t(n), 1, int(threads), _p(out))
    v = _ints(out)
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def g1_affine(jac):
    a = _u64(jac)
    out = np.zeros(8, dtype=np.uint64)
    inf = lib().oracle_g1_affine(_p(a), _p(out))
    return None if inf else tuple(_ints(out))


def g2_affine(jac):
    a = _u64([c for xy in jac for c in xy])
    out = np.zeros(16, dtype=np.uint64)
    inf = lib().oracle_g2_affine(_p(a), _p(out))
    if inf:
        return None
    v = _ints(out)
    return ((v[0], v[1]), (v[2], v[3]))


def g1_add(a, b):
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_g1_add(_p(_u64(a)), _p(_u64(b)), _p(out))
    return tuple(_ints(out))


def g1_mul_scalar(p, k):
    out = np.zeros(12, dtype=np.uint64)
    lib().oracle_g1_mul_scalar(_p(_u64(p)), _p(_u64([k])), _p(out))
    return tuple(_ints(out))


def g2_add(a, b):
    out = np.zeros(24, dtype=np.uint64)
    fa = [c for xy in a for c in xy]
    fb = [c for xy in b for c in xy]
    lib().oracle_g2_add(_p(_u64(fa)), _p(_u64(fb)), _p(out))
    v = _ints(out)
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def g2_mul_scalar(p, k):
    out = np.zeros(24, dtype=np.uint64)
    fp = [c for xy in p for c in xy]
    lib().oracle_g2_mul_scalar(_p(_u64(fp)), _p(_u64([k])), _p(out))
    v = _ints(out)
    return ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))


def poly_mul(a, b):
    out = np.zeros(4 * (len(a) + len(b) - 1), dtype=np.uint64)
    lib().oracle_poly_mul(_p(_u64(a)), ctypes.c_size_t(len(a)), _p(_u64(b)), ctypes.c_size_t(len(b)), _p(out))
    return _ints(out)


def poly_div(a, b):
    nq, nr = len(a) - len(b) + 1, len(b) - 1
    q = np.zeros(4 * nq, dtype=np.uint64)
    r = np.zeros(4 * max(nr, 1), dtype=np.uint64)
    lib().oracle_poly_div(_p(_u64(a)), ctypes.c_size_t(len(a)), _p(_u64(b)), ctypes.c_size_t(len(b)), _p(q), _p(r))
    return _ints(q), _ints(r)[:nr]


def poly_div_u64(a_u64, b_u64):
    a = np.ascontiguousarray(a_u64, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b_u64, dtype=np.uint64).reshape(-1, 4)
    nq, nr = a.shape[0] - b.shape[0] + 1, b.shape[0] - 1
    q = np.zeros((nq, 4), dtype=np.uint64)
    r = np.zeros((max(nr, 1), 4), dtype=np.uint64)
    lib().oracle_poly_div(_p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(q), _p(r))
    return q, r[:nr]


def lagrange(values):
    out = np.zeros(4 * len(values), dtype=np.uint64)
    lib().oracle_lagrange(_p(_u64(values)), ctypes.c_size_t(len(values)), _p(out))
    return _ints(out)


def poly_eval(v, x):
    out = np.zeros(4, dtype=np.uint64)
    lib().oracle_poly_eval(_p(_u64(v)), ctypes.c_size_t(len(v)), _p(_u64([x])), _p(out))
    return _ints(out)[0]


def mul_scalar_batch(base_jac, scalars_u64, g2=False, threads=1):
    """[MulScalar(base, k_i)] as an [n, 12] / [n, 24] Jacobian limb array (the setup's encryption loops, multi-threaded)."""
    sc = np.ascontiguousarray(scalars_u64, dtype=np.uint64)
    n = sc.size // 4
    b = _u64([c for xy in base_jac for c in xy]) if g2 else _u64(base_jac)
    out = np.zeros((n, 24 if g2 else 12), dtype=np.uint64)
    lib().oracle_mul_scalar_batch_mt(_p(b), _p(sc), ctypes.c_size_t(n), 1 if g2 else 0, int(threads), _p(out))
    return out


def poly_u64(coeffs):
    """list of ints -> [n, 4] limb array"""
    return _u64(coeffs).reshape(-1, 4)


def poly_mul_u64(a_u64, b_u64):
    a = np.ascontiguousarray(a_u64, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b_u64, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((a.shape[0] + b.shape[0] - 1, 4), dtype=np.uint64)
    lib().oracle_poly_mul(_p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(out))
    return out
