"""Soak: many proofs through every prove entry point; device memory and results must stay put (dev tool)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gosnark_amd
from gosnark_amd import capi, synth, groth16, snark
capi.init()
inst = synth.sqchain_setup_instance(1 << 18, 11)
pin = synth.sqchain_pinocchio_instance(1 << 16, 12)
pk = inst.device_pk()
r, s = synth.field_elems(2, 5)
want = groth16.prove_resident(pk, inst.w, inst.px, r, s)
wantp = snark.prove_resident(pin.device_pk(), pin.w, pin.px)
def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
# first use of every path allocates its grow-only workspaces (three tickets' worth) and tables: warm them before the baseline
ts = [groth16.prove_begin(pk, inst.w, inst.px, r, s) for _ in range(3)]
for t in ts:
    groth16.prove_end(t)
_h = capi.g1_fixed_base(synth.scalars_u64(1 << 16, 3)); _sc = capi.scalars_upload(synth.scalars_u64(1 << 16, 4))
capi.msm_resident(_h, _sc, 1 << 16)
f0 = free_mb()
t0 = time.perf_counter()
tickets, done = [], 0
for i in range(1500):
    tickets.append(groth16.prove_begin(pk, inst.w, inst.px, r, s))
    if len(tickets) == 3:
        p = groth16.prove_end(tickets.pop(0)); done += 1
        if done % 250 == 0:
            assert (p.PiA, p.PiB, p.PiC) == (want.PiA, want.PiB, want.PiC)
            print("pipelined", done, "free MiB %.0f" % free_mb(), flush=True)
while tickets:
    groth16.prove_end(tickets.pop(0))
for i in range(300):
    p = groth16.prove_resident(pk, inst.w, inst.px, r, s)
assert (p.PiA, p.PiB, p.PiC) == (want.PiA, want.PiB, want.PiC)
for i in range(200):
    q = snark.prove_resident(pin.device_pk(), pin.w, pin.px)
assert all(getattr(q, k) == getattr(wantp, k) for k in snark.Proof.FIELDS)
h, sc = _h, _sc
m0 = capi.msm_resident(h, sc, 1 << 16)
for i in range(500):
    m = capi.msm_resident(h, sc, 1 << 16)
assert m == m0
f1 = free_mb()
print("done in %.1f s; free device memory before %.0f MiB, after %.0f MiB (delta %.1f)" % (time.perf_counter() - t0, f0, f1, f1 - f0))
assert abs(f1 - f0) < 64, "device memory drifted"
