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
# ---- round 3: witness -> proof tickets (evaluation-basis key) with violated witnesses in between (late detection, exact repeat inside
# prove_end), abandoned tickets (gs_ticket_cancel), gs_trim under outstanding tickets, and values-route partials
import random
from gosnark_amd import r1csqap
rng = random.Random(5)
dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
w_bad = inst.w_host.copy(); w_bad[17] = (12345, 0, 0, 0)
wh = capi.scalars_upload(w_bad)
good = groth16.prove_from_witness(pk, dev, inst.w, r, s)
bad = groth16.prove_from_witness(pk, dev, wh, r, s)
assert (good.PiA, good.PiB, good.PiC) == (want.PiA, want.PiB, want.PiC) and (bad.PiA, bad.PiB, bad.PiC) != (want.PiA, want.PiB, want.PiC)
f_w0 = free_mb()
tickets, done, fallbacks, cancelled = [], 0, 0, 0
for i in range(600):
    is_bad = rng.random() < 0.1
    tickets.append((groth16.prove_witness_begin(pk, dev, wh if is_bad else inst.w, r, s), is_bad))
    if rng.random() < 0.02:
        t, _ = tickets.pop(rng.randrange(len(tickets))); capi.ticket_cancel(t); cancelled += 1
    if rng.random() < 0.01:
        capi.trim()                                # outstanding tickets are waited for, their violated-constraint words survive
    if len(tickets) >= 3:
        t, is_bad = tickets.pop(0)
        p = groth16.prove_end(t); done += 1; fallbacks += int(is_bad)
        ref = bad if is_bad else good
        assert (p.PiA, p.PiB, p.PiC) == (ref.PiA, ref.PiB, ref.PiC), (i, is_bad)
        if done % 200 == 0:
            print("witness tickets", done, "fallbacks", fallbacks, "cancelled", cancelled, "free MiB %.0f" % free_mb(), flush=True)
for t, is_bad in tickets:
    p = groth16.prove_end(t); ref = bad if is_bad else good
    assert (p.PiA, p.PiB, p.PiC) == (ref.PiA, ref.PiB, ref.PiC)
hv, violated = groth16.witness_values(pk, dev, inst.w)
assert violated == 0
slices = groth16.scatter_values(hv, 1)
sums0, _flags = groth16.prove_partials_values(pk, inst.w, slices[0], 0, 1)
for i in range(150):
    ts = [groth16.partials_values_begin(pk, inst.w, slices[0], 0, 1) for _ in range(3)]
    for t in ts:
        assert groth16.partials_end(t) == sums0
pv = groth16.finish(pk, sums0, r, s)
assert (pv.PiA, pv.PiB, pv.PiC) == (want.PiA, want.PiB, want.PiC)
capi.trim()
for _ in range(3):                                 # re-warm what gs_trim dropped before the memory comparison
    groth16.prove_end(groth16.prove_begin(pk, inst.w, inst.px, r, s))
capi.msm_resident(h, sc, 1 << 16)
f1 = free_mb()
print("done in %.1f s; free device memory before %.0f MiB, after %.0f MiB (delta %.1f)" % (time.perf_counter() - t0, f0, f1, f1 - f0))
assert f0 - f1 < 64, "device memory leaked"      # gs_trim may have returned MORE than the baseline held (tables of the other keys)
