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
