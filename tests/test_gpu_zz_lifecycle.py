"""-m gpu: library lifetime (runs after the other GPU modules: it shuts the context down and brings it back)."""
import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi
import gpu_util as U
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


def test_shutdown_invalidates_handles_and_init_brings_the_library_back():
    capi.init()
    n = 3000
    ks, sc = U.rand_scalars_u64(n, 31), U.rand_scalars_u64(n, 32)
    bases = capi.g1_fixed_base(ks)
    before = capi.msm(bases, sc)
    old = bases.h
    capi.shutdown()
    lib = capi.load_library()
    h = capi.Handle(0)
    z = np.zeros((1, 12), dtype=np.uint64)
    assert lib.gs_g1_upload(capi.ptr64(z), 1, capi.ctypes.byref(h)) == -5          # GS_ERR_NOT_INIT, loudly
    assert b"gs_init" in lib.gs_last_error()
    capi.shutdown()                                                                # idempotent
    capi.init()
    out, inf = np.zeros(8, dtype=np.uint64), capi.ctypes.c_int(0)
    assert lib.gs_msm_g1(capi.Handle(old), capi.ptr64(sc), 0, n, capi.ptr64(out), capi.ctypes.byref(inf)) < 0   # the old handle is gone
    bases.h = 0                                                                    # nothing left to free
    again = capi.g1_fixed_base(ks)
    assert capi.msm(again, sc) == before
    pts = capi.g1_download(again)
    assert before == C.g1_affine(C.g1_msm_naive(pts, sc, threads=8))
