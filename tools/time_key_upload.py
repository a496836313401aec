"""Where the time of a key upload goes (dev tool, round 6): the five base arrays of a 2^log2n Groth16 key, from a memory-mapped key file
(what a CLI does: utils.ReadBinary) and from anonymous memory, with GS_HOST_TRACE=1 printing hostcopy.h's breakdown per array.

    GS_HOST_TRACE=1 python tools/time_key_upload.py [log2n = 20]
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import gosnark_amd  # noqa: F401
    from gosnark_amd import capi, synth, groth16, utils
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    capi.init()
    n = 1 << logn
    inst = synth.sqchain_setup_instance(n, 3)
    pk = inst.device_pk()
    path = os.path.join(tempfile.gettempdir(), "gs_upload_key_%d.bin" % os.getpid())
    utils.GrothSetupToBinary(path, groth16.Circuit(pk.nvars, pk.npublic), pk, None)
    names = ["G1.At", "G1.BACGamma", "BACDelta", "PowersTauDelta", "G2.BACGamma"]
    try:
        for source in ("memory-mapped key file", "anonymous memory (np.array copies of the same)", "memory-mapped key file again"):
            protocol, nvars, npublic, sec = utils.ReadBinary(path)
            arrays = {k: np.ascontiguousarray(sec[k], dtype=np.uint64) for k in names}
            if source.startswith("anonymous"):
                arrays = {k: np.array(v, copy=True) for k, v in arrays.items()}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hs, per = [], []
            for k in names:
                ta = time.perf_counter()
                hs.append(capi.g2_upload(arrays[k]) if k.startswith("G2") else capi.g1_upload(arrays[k]))
                per.append((time.perf_counter() - ta) * 1e3)
            torch.cuda.synchronize()
            total = (time.perf_counter() - t0) * 1e3
            mb = sum(a.nbytes for a in arrays.values()) / 1e6
            print("%-48s %.0f MB in %.1f ms = %.1f GB/s | per array ms: %s" % (source, mb, total, mb / total, " ".join("%.1f" % p for p in per)))
            for h in hs:
                h.free()
            del sec, arrays
    finally:
        os.remove(path)


if __name__ == "__main__":
    main()
