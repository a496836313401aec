"""Where does a small proof's wall time go?  Host time inside prove_begin / prove_end vs device time (dev tool)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gosnark_amd
from gosnark_amd import capi, synth, groth16
capi.init()
for logn in (16, 18, 20):
    inst = synth.sqchain_setup_instance(1 << logn, 3)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 5)
    for _ in range(4):
        groth16.prove_resident(pk, inst.w, inst.px, r, s)
    K = 24
    tb = te = 0.0
    tickets = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        a = time.perf_counter()
        tickets.append(groth16.prove_begin(pk, inst.w, inst.px, r, s))
        b = time.perf_counter()
        tb += b - a
        if len(tickets) == 3:
            groth16.prove_end(tickets.pop(0))
            te += time.perf_counter() - b
    while tickets:
        a = time.perf_counter()
        groth16.prove_end(tickets.pop(0))
        te += time.perf_counter() - a
    wall = time.perf_counter() - t0
    tm = capi.last_timing()
    print("2^%d: wall %.3f ms/proof; host in begin %.3f, in end (wait + tail) %.3f; last proof device total %.3f ms acc %.3f" % (
        logn, wall / K * 1e3, tb / K * 1e3, te / K * 1e3, tm["total_ms"], tm["accumulate_ms"]), flush=True)
