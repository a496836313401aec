"""CPU-side checks of the drop-in boundary: libgosnark_hip.so loads without a GPU, exports exactly the entry points
include/gosnark_hip.h declares, and every compute call fails loudly (no CPU fallback) when there is no device."""
import ctypes
import os
import re

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gosnark_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes binding covers the whole header
    assert sorted(capi.EXPORTS) == names


def test_version_and_error_strings():
    lib = capi.load_library()
    assert b"gfx950" in lib.gs_version()
    assert isinstance(lib.gs_last_error(), bytes)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_no_device_means_loud_failure_not_fallback():
    lib = capi.load_library()
    dev = (ctypes.c_int * 1)(0)
    assert lib.gs_init(dev, 1) == -1                        # GS_ERR_NO_DEVICE
    assert b"no CPU path" in lib.gs_last_error() or b"no HIP device" in lib.gs_last_error()
    s = np.zeros((4, 4), dtype=np.uint64)
    h = capi.Handle(0)
    assert lib.gs_scalars_upload(capi.ptr64(s), 4, ctypes.byref(h)) == -5     # GS_ERR_NOT_INIT
    out = np.zeros((3, 4), dtype=np.uint64)
    assert lib.gs_poly_mul(capi.ptr64(s), 2, capi.ptr64(s), 2, capi.ptr64(out)) == -5
    with pytest.raises(capi.GosnarkHipError):
        capi.init(0)


def test_sum_affine_is_host_side_complete_addition():
    """gs_g1_sum_affine (multi-GPU combine) runs on the host core: P + P, P + (-P), infinities."""
    from oracle import ref_py as O
    p = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, 12345))
    q = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, 54321))
    want = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, 12345 + 54321))
    assert capi.sum_affine([p, q]) == want
    assert capi.sum_affine([p, None, q, None]) == want
    assert capi.sum_affine([p, p]) == O.G1.Affine(O.G1.MulScalar(O.G1_GEN, 2 * 12345))
    assert capi.sum_affine([p, (p[0], O.Q - p[1])]) is None
    assert capi.sum_affine([]) is None
    g2 = O.G2.Affine(O.G2.MulScalar(O.G2_GEN, 777))
    assert capi.sum_affine([g2, g2], g2=True) == O.G2.Affine(O.G2.MulScalar(O.G2_GEN, 1554))


def test_header_is_plain_c_as_cgo_needs_it(tmp_path):
    """cgo compiles include/gosnark_hip.h as C: it must be valid C99 on its own, and a C translation unit that calls the
    entry points the Go binding uses must compile and link against the library."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    hdr = os.path.join(root, "include", "gosnark_hip.h")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "use.c"
    src.write_text('''#include "gosnark_hip.h"
#include <stdio.h>
int main(void) {
  int dev = 0, ok = -1;
  uint64_t one[12] = {1,0,0,0, 2,0,0,0, 1,0,0,0};
  uint64_t q[24] = {0};
  printf("%s\\n", gs_version());
  if (gs_pairing_check(one, q, 1, &ok) != 0 || ok != 1) return 2;      /* e(G, infinity) = 1: host-side entry point */
  if (gs_init(&dev, 1) == 0) gs_shutdown();                              /* no device here: must fail, loudly, not crash */
  else printf("%s\\n", gs_last_error());
  return 0;
}
''')
    exe = tmp_path / "use"
    libdir = os.path.join(root, "go-snark-study_amd")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lgosnark_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gosnark-hip" in out.stdout


def test_multi_device_entry_points_without_a_device():
    """No GPU here: the per-device contexts, the communicator and the multi-device entry points must say so, not crash."""
    import ctypes
    lib = capi.load_library()
    assert lib.gs_device_count() == 0
    assert lib.gs_set_device(0) == -3 and b"no logical device" in lib.gs_last_error()
    assert lib.gs_comm_init_local() == -5                                   # GS_ERR_NOT_INIT
    nr, rk, loc, cnt = ctypes.c_int(9), ctypes.c_int(9), ctypes.c_int(9), ctypes.c_uint64(9)
    assert lib.gs_comm_info(ctypes.byref(nr), ctypes.byref(rk), ctypes.byref(loc), ctypes.cast(ctypes.byref(cnt), capi.u64p)) == 0
    assert (nr.value, rk.value, loc.value) == (0, -1, 0)
    h = (capi.Handle * 2)(1, (1 << 56) | 1)
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    used = ctypes.c_int(0)
    rs = capi.ints_to_u64([1, 2])
    st = lib.gs_groth16_prove_multi(h, h, h, 2, capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf, ctypes.byref(used))
    assert st < 0 and b"gs_init did not create" in lib.gs_last_error()
    assert lib.gs_handle_device(capi.Handle((5 << 56) | 77)) == 5
    assert lib.gs_verify_set_strict(1) == 0 and lib.gs_verify_set_strict(0) == 0


def test_c_drivers_of_the_go_wrappers_compile_and_link(tmp_path):
    """tests/c/*.c (the executable mirror of go/: every wrapper's exact call sequence) must at least build everywhere; they RUN in
    the -m gpu suite (tests/test_gpu_c_drivers.py, tests/test_gpu_zy_multi.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import c_util
    for name in sorted(os.listdir(c_util.CDIR)):
        if name.endswith(".c"):
            exe = c_util.compile_c(name, tmp_path)
            assert os.path.exists(exe)


def test_verifier_strict_mode_rejects_aliased_inputs():
    """with gs_verify_set_strict(1) a public signal x + r, a coordinate X + q or a short signal list no longer pass;
    the default stays the reference's big.Int behaviour."""
    import golden_util as GU
    from gosnark_amd import groth16, utils
    from oracle import ref_py as O
    rec = GU.load("groth_x3")
    _, vk = utils.GrothSetupFromString(rec["setup"])
    pr = utils.GrothProofFromString(rec["proof"]) if hasattr(utils, "GrothProofFromString") else None
    if pr is None:
        pr = groth16.Proof(GU.g1(rec["proof"]["PiA"]), GU.g2(rec["proof"]["PiB"]), GU.g1(rec["proof"]["PiC"]))
    lib = capi.load_library()

    def raw_verify(pub_ints, proof):
        import ctypes
        ic = capi.g1_points_to_u64(vk.IC)
        pub = capi.ints_to_u64(pub_ints) if pub_ints else np.zeros((1, 4), dtype=np.uint64)
        g1 = capi.g1_points_to_u64([vk.G1_Alpha, proof.PiA, proof.PiC])
        g2 = capi.g2_points_to_u64([vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta, proof.PiB])
        ok = ctypes.c_int(7)
        st = lib.gs_groth16_verify(capi.ptr64(g1[0]), capi.ptr64(g2[0]), capi.ptr64(g2[1]), capi.ptr64(g2[2]), capi.ptr64(ic), len(vk.IC),
                                   capi.ptr64(pub), len(pub_ints), capi.ptr64(g1[1]), capi.ptr64(g2[3]), capi.ptr64(g1[2]), ctypes.byref(ok))
        return st, ok.value
    a = O.G1.Affine(pr.PiA)
    aliased = groth16.Proof((a[0] + O.Q, a[1], 1), pr.PiB, pr.PiC)                 # X + q < 2^256 names the same point
    try:
        assert raw_verify([35], pr) == (0, 1) and raw_verify([35 + O.R], pr) == (0, 1) and raw_verify([35], aliased) == (0, 1)
        assert raw_verify([], pr) == (0, 0)                                        # lenient: missing inputs are zeros -> reject, no error
        assert lib.gs_verify_set_strict(1) == 0
        assert raw_verify([35], pr) == (0, 1)
        assert raw_verify([35 + O.R], pr) == (0, 0) and raw_verify([35], aliased) == (0, 0)
        st, _ = raw_verify([], pr)
        assert st == -4 and b"strict" in lib.gs_last_error()
    finally:
        lib.gs_verify_set_strict(0)


def test_memory_and_witness_entry_points_without_a_device():
    """The memory accounting / eviction calls and the witness -> proof provers exist and say GS_ERR_NOT_INIT without a GPU (nothing
    falls back to host code)."""
    import ctypes
    lib = capi.load_library()
    m = capi.Memory()
    assert lib.gs_memory_query(ctypes.byref(m)) == -5 and b"gs_init" in lib.gs_last_error()
    a = ctypes.c_uint64(7)
    assert lib.gs_handle_bytes(capi.Handle(1), ctypes.cast(ctypes.byref(a), capi.u64p), None) == -5 and a.value == 7
    assert lib.gs_release_tables(capi.Handle(1)) == -5
    assert lib.gs_trim() == -5
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    rs = capi.ints_to_u64([1, 2])
    assert lib.gs_pinocchio_prove_witness(capi.Handle(1), capi.Handle(2), capi.Handle(3), capi.ptr64(out), inf) == -5
    assert lib.gs_groth16_prove_witness(capi.Handle(1), capi.Handle(2), capi.Handle(3), capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf) == -5
    assert not out.any()
