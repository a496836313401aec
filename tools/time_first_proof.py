"""Time from a fresh resident key to its first proof (window tables are built on the first prove) -- dev tool."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gosnark_amd
from gosnark_amd import capi, synth, groth16
capi.init()
synth.sqchain_setup_instance(1 << 10, 1)
for logn in (16, 20):
    inst = synth.sqchain_setup_instance(1 << logn, 3)
    r, s = synth.field_elems(2, 5)
    torch.cuda.synchronize(); t = time.perf_counter()
    groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    t1 = time.perf_counter() - t
    t = time.perf_counter()
    groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    t2 = time.perf_counter() - t
    print("2^%d: first proof (tables + proof) %.1f ms, second %.1f ms" % (logn, t1 * 1e3, t2 * 1e3), flush=True)
