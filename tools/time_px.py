"""Dev probe: time the sparse R1CS -> px stage (host-buffer and resident) and the device trusted setup at 2^log2n."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gosnark_amd
from gosnark_amd import capi, synth, r1csqap, groth16
capi.init()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
a, b, c, w = synth.sqchain_r1cs(n, 12345)
for rep in range(2):
    t0 = time.time(); out = r1csqap.ComputePx(a, b, c, w, n + 1); dt = time.time() - t0
    print("ComputePx (host buffers) 2^%d: %.1f ms" % (logn, dt * 1e3))
dev = r1csqap.DeviceR1CS(a, b, c, n + 1)
wh = capi.scalars_upload(w)
px = None
for rep in range(4):
    t0 = time.time(); px = dev.ComputePxResident(wh, px); dt = time.time() - t0
    print("ComputePxResident 2^%d: wall %.2f ms, device %.2f ms" % (logn, dt * 1e3, capi.last_timing()["poly_ms"]))
