"""go-snark-study_amd: MI355X (gfx950) prover hot path of arnaucube/go-snark-study.

Host-side mirror (Python, because no Go toolchain exists in this image -- see INTEGRATION.md
for the cgo binding) of the reference's prover interface, on top of the C ABI of
libgosnark_hip.so (include/gosnark_hip.h):

    gosnark_amd.groth16.GenerateProofs(circuit, pk, w, px)     <- groth16/groth16.go:225
    gosnark_amd.snark.GenerateProofs(circuit, pk, w, px)       <- snark.go:254
    gosnark_amd.bn128.G1 / G2 (MulScalar/Add loops -> MSM)     <- bn128/g1.go, g2.go
    gosnark_amd.r1csqap.PolynomialField                        <- r1csqap/r1csqap.go

There is NO CPU fallback: importing works anywhere (so CPU-only test collection succeeds), but
every compute call needs the HIP library and a gfx950 device and raises GosnarkHipError otherwise.
"""
from . import capi                      # noqa: F401
from .capi import GosnarkHipError, lib_path, load_library, init   # noqa: F401

__all__ = ["capi", "GosnarkHipError", "lib_path", "load_library", "init"]
