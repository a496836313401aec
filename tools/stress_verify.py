"""Stress: pipelined proofs with fresh (r, s) each, every single proof verified by the pairing check (dev tool)."""
import sys, time, random
sys.path.insert(0, "/root/repo")
import gosnark_amd
from gosnark_amd import capi, synth, groth16, snark
capi.init()
rng = random.Random(2026)
total = 0
for logn in (12, 16, 18):
    inst = synth.sqchain_setup_instance(1 << logn, 100 + logn)
    pub = capi.u64_to_ints(inst.w_host[1:2])
    pk = inst.device_pk()
    tickets = []
    count = 300 if logn <= 16 else 100
    t0 = time.perf_counter()
    for i in range(count):
        r, s = rng.randrange(groth16.R), rng.randrange(groth16.R)
        tickets.append(groth16.prove_begin(pk, inst.w, inst.px, r, s))
        if len(tickets) == 3:
            p = groth16.prove_end(tickets.pop(0))
            assert groth16.VerifyProof(inst.vk, p, pub), "a pipelined proof failed the pairing check"
            total += 1
    while tickets:
        p = groth16.prove_end(tickets.pop(0))
        assert groth16.VerifyProof(inst.vk, p, pub)
        total += 1
    print("2^%d: %d proofs with fresh randomness, all verified, %.1f s" % (logn, count, time.perf_counter() - t0), flush=True)
pin = synth.sqchain_pinocchio_instance(1 << 14, 77)
for i in range(60):
    ts = [snark.prove_begin(pin.device_pk(), pin.w, pin.px) for _ in range(3)]
    for t in ts:
        assert snark.VerifyProof(pin.vk, snark.prove_end(t), pin.public)
        total += 1
print("all %d proofs verified" % total)
