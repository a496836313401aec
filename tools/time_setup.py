import time, sys
sys.path.insert(0, "/root/repo")
import torch
import gosnark_amd
from gosnark_amd import capi, synth, groth16, r1csqap
capi.init()
synth.sqchain_setup_instance(1 << 10, 1)
for logn in (16, 20):
    n = 1 << logn
    x = synth.field_elems(1, 5)[0]
    a, b, c, w = synth.sqchain_r1cs(n, x, 0)
    tox = synth.field_elems(5, 9)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        pk, vk = groth16.GenerateTrustedSetupSparse(n, n + 1, 1, a, b, c, tox)
        dt = time.perf_counter() - t
        print("groth16 device setup n=2^%d: %.3f s" % (logn, dt), flush=True)
        pk.handle.free()
