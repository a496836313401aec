"""Mirror of the reference's Pinocchio prover interface (snark.go:16-26, 59-69, 254-289)."""
import ctypes

import numpy as np

from . import capi
from .groth16 import Circuit, R   # noqa: F401


class Pk:
    """snark.Pk (snark.go:16-26)."""

    def __init__(self, G1T, A, B, C, Kp, Ap, Bp, Cp, Z):
        self.G1T, self.A, self.B, self.C = G1T, A, B, C
        self.Kp, self.Ap, self.Bp, self.Cp, self.Z = Kp, Ap, Bp, Cp, Z
        self._dev = None


class Proof:
    """snark.Proof (snark.go:59-69)."""
    FIELDS = ("PiA", "PiAp", "PiB", "PiBp", "PiC", "PiCp", "PiH", "PiKp")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])


def UploadPk(pk, circuit):
    if pk._dev is not None:
        return pk._dev
    capi.init()
    g1 = {k: capi.g1_upload(capi.g1_points_to_u64(getattr(pk, k))) for k in ("A", "Ap", "Bp", "C", "Cp", "Kp", "G1T")}
    b2 = capi.g2_upload(capi.g2_points_to_u64(pk.B))
    z = capi.ints_to_u64([x % R for x in pk.Z])
    h = capi.Handle(0)
    H = lambda x: capi.Handle(x.h)   # noqa: E731
    capi.check(capi.load_library().gs_pinocchio_pk_create(
        H(g1["A"]), H(g1["Ap"]), H(b2), H(g1["Bp"]), H(g1["C"]), H(g1["Cp"]), H(g1["Kp"]), H(g1["G1T"]),
        capi.ptr64(z), z.shape[0], circuit.NVars, circuit.NPublic, ctypes.byref(h)))
    pk._dev = capi.DeviceHandle(h.value)
    return pk._dev


def GenerateProofs(circuit, pk, w, px):
    """snark.GenerateProofs(circuit, pk, w, px) (snark.go:254-289).  Deterministic."""
    dev = UploadPk(pk, circuit)
    if any(x < 0 for x in w):
        raise ValueError("negative witness values are not supported")
    wa = capi.ints_to_u64([x % R for x in w])
    pa = capi.ints_to_u64([x % R for x in px])
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove(capi.Handle(dev.h), capi.ptr64(wa), len(w), capi.ptr64(pa), len(px),
                                                      capi.ptr64(out), inf))
    v = capi.u64_to_ints(out)
    res, pos = {}, 0
    for i, k in enumerate(Proof.FIELDS):
        if k == "PiB":
            res[k] = ((0, 0), (0, 0), (0, 0)) if inf[i] else ((v[pos], v[pos + 1]), (v[pos + 2], v[pos + 3]), (1, 0))
            pos += 4
        else:
            res[k] = (0, 0, 0) if inf[i] else (v[pos], v[pos + 1], 1)
            pos += 2
    return Proof(**res)
