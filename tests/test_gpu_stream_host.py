"""-m gpu: the reference's call shape at the pipelined rate -- a NEW witness (and px) in host memory with every call
(groth16/groth16.go:225, snark.go:254; cli/main.go:480-501 computes them per proof).

 * host-buffer tickets (gs_*_host_begin) stage the caller's arrays into buffers their slot owns: 50 distinct witnesses streamed three in
   flight equal their blocking proofs, each pinned by its own closed form / the verifier; no hipMalloc / hipFree in steady state;
 * gs_scalars_update overwrites a resident vector in place, ordered behind the reads of the tickets that still use it;
 * violated constraints, shape errors and a full pipeline behave as with resident inputs."""
import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi, groth16, snark, r1csqap, synth
from oracle import c_oracle as C
from oracle import ref_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _init():
    capi.init()
    capi.set_table_policy("always")
    yield
    capi.set_table_policy("auto")


def same(p, q):
    return (p.PiA, p.PiB, p.PiC) == (q.PiA, q.PiB, q.PiC)


def stream(begin, items, depth=3):
    """begin(item) -> ticket, three in flight, proofs in order"""
    out, tickets = [], []
    for it in items:
        tickets.append(begin(it))
        if len(tickets) == depth:
            out.append(groth16.prove_end(tickets.pop(0)))
    while tickets:
        out.append(groth16.prove_end(tickets.pop(0)))
    return out


@pytest.mark.parametrize("logn", [10, 14])
def test_fifty_distinct_host_witnesses_streamed_equal_their_blocking_proofs(logn):
    n = 1 << logn
    inst = synth.sqchain_setup_instance(n, 0x7100 + logn)
    pk = inst.device_pk()
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 61 + logn)
    count = 50 if logn == 10 else 12
    xs = synth.field_elems(count, 0x7200 + logn)
    ws = [synth.sqchain_witness(n, x) for x in xs]
    want, pxs = [], []
    for w in ws:                                                    # blocking proofs of resident copies: the reference result
        h = capi.scalars_upload(w)
        ph = dr.ComputePxResident(h)
        want.append(groth16.prove_resident(pk, h, ph, r, s))
        pxs.append(capi.scalars_download(ph))
        h.free()
        ph.free()
    assert not same(want[0], want[1])
    # witness route, host-buffer tickets; the first lap over the three slots may allocate, after that nothing moves
    got = stream(lambda k: groth16.prove_witness_host_begin(pk, dr, ws[k], r, s), range(6))
    a0 = capi.alloc_counters()
    got = stream(lambda k: groth16.prove_witness_host_begin(pk, dr, ws[k], r, s), range(count))
    assert capi.alloc_counters() == a0, "hipMalloc / hipFree in the steady state of a host-witness stream"
    assert all(same(g, w) for g, w in zip(got, want))
    # px route from the host (w and px per call)
    stream(lambda k: groth16.prove_host_begin(pk, ws[k], pxs[k], r, s), range(6))
    a0 = capi.alloc_counters()
    got = stream(lambda k: groth16.prove_host_begin(pk, ws[k], pxs[k], r, s), range(count))
    assert capi.alloc_counters() == a0, "hipMalloc / hipFree in the steady state of a host-px stream"
    assert all(same(g, w) for g, w in zip(got, want))
    # integers instead of arrays; the blocking form
    ints = capi.u64_to_ints(ws[3])
    assert same(groth16.prove_end(groth16.prove_witness_host_begin(pk, dr, ints, r, s)), want[3])
    assert same(groth16.prove_from_witness_host(pk, dr, ws[4], r, s), want[4])
    # in-place updates of four rotating resident vectors
    rot = [capi.scalars_upload(ws[k]) for k in range(4)]
    stream(lambda k: groth16.prove_witness_begin(pk, dr, rot[k % 4], r, s), range(4))

    def upd(k):
        capi.scalars_update(rot[k % 4], ws[k])
        return groth16.prove_witness_begin(pk, dr, rot[k % 4], r, s)
    a0 = capi.alloc_counters()
    got = stream(upd, range(count))
    assert capi.alloc_counters() == a0
    assert all(same(g, w) for g, w in zip(got, want))
    # every streamed proof is the proof of ITS witness: closed form from the toxic values (three of them) and the verifier (all)
    for k in (0, count // 2, count - 1):
        a, b, c = inst.expected_proof_scalars(r, s, ws[k])
        assert (got[k].PiA[0], got[k].PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
        assert (got[k].PiB[0], got[k].PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
        assert (got[k].PiC[0], got[k].PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    for k in range(count):
        assert groth16.VerifyProof(inst.vk, got[k], [xs[k]]) and not groth16.VerifyProof(inst.vk, got[k], [xs[(k + 1) % count]])


def test_update_is_ordered_behind_the_tickets_that_read_the_vector():
    """gs_scalars_update right after a _begin that reads the vector: the ticket still proves the OLD values (the copy waits for the
    plan's digit pass and the witness copy), the next one the new."""
    n = 1 << 16
    inst = synth.sqchain_setup_instance(n, 0x7300)
    pk = inst.device_pk()
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 62)
    wa, wb = inst.w_host, synth.sqchain_witness(n, 424242)
    ha, hb = capi.scalars_upload(wa), capi.scalars_upload(wb)
    pa, pb = groth16.prove_from_witness(pk, dr, ha, r, s), groth16.prove_from_witness(pk, dr, hb, r, s)
    pxa, pxb = dr.ComputePxResident(ha), dr.ComputePxResident(hb)
    assert not same(pa, pb)
    h = capi.scalars_upload(wa)
    for _ in range(3):
        # two tickets keep the device busy, the third reads h -- and h is overwritten while all three are queued
        t = [groth16.prove_witness_begin(pk, dr, ha, r, s), groth16.prove_witness_begin(pk, dr, hb, r, s), groth16.prove_witness_begin(pk, dr, h, r, s)]
        capi.scalars_update(h, wb)
        res = [groth16.prove_end(x) for x in t]
        assert same(res[0], pa) and same(res[1], pb) and same(res[2], pa)
        assert same(groth16.prove_end(groth16.prove_witness_begin(pk, dr, h, r, s)), pb)
        # px route: the ticket reads both vectors; px is overwritten as well
        hp = capi.scalars_clone(pxb, capi.get_device())
        t = [groth16.prove_begin(pk, ha, pxa, r, s), groth16.prove_begin(pk, h, hp, r, s)]
        capi.scalars_update(h, wa)
        capi.scalars_update(hp, capi.scalars_download(pxa))
        res = [groth16.prove_end(x) for x in t]
        assert same(res[0], pa) and same(res[1], pb)
        assert same(groth16.prove_resident(pk, h, hp, r, s), pa)
        hp.free()
    # MSM tickets mark their scalar vector too
    bases = capi.g1_fixed_base(synth.scalars_u64(1 << 14, 9))
    s1, s2 = synth.scalars_u64(1 << 14, 10), synth.scalars_u64(1 << 14, 11)
    hs = capi.scalars_upload(s1)
    m1 = capi.msm_resident(bases, hs, 1 << 14)
    t = capi.msm_begin(bases, hs, 1 << 14)
    capi.scalars_update(hs, s2)
    assert capi.msm_end(t) == m1
    assert capi.msm_resident(bases, hs, 1 << 14) == capi.msm(bases, s2)
    with pytest.raises(capi.GosnarkHipError) as e:
        capi.scalars_update(hs, s2[:100])
    assert e.value.code == -3


def test_host_tickets_errors_busy_and_violated_constraints():
    n = 300
    inst = synth.sqchain_setup_instance(n, 0x7400)
    pk = inst.device_pk()
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    r, s = synth.field_elems(2, 63)
    good = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.prove_witness_host_begin(pk, dr, inst.w_host[:-1], r, s)
    assert e.value.code == -4                                         # GS_ERR_SHAPE, nothing staged, no slot taken
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.prove_host_begin(pk, inst.w_host[:-1], inst.px_host, r, s)
    assert e.value.code == -4
    t = [groth16.prove_witness_host_begin(pk, dr, inst.w_host, r, s) for _ in range(3)]
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.prove_witness_host_begin(pk, dr, inst.w_host, r, s)
    assert e.value.code == -6                                         # GS_ERR_BUSY
    assert all(same(groth16.prove_end(x), good) for x in t)
    # a witness that violates a constraint: detected when the ticket is collected, repeated on the exact route FROM THE SLOT'S COPY
    w_bad = inst.w_host.copy()
    w_bad[17] = (12345, 0, 0, 0)
    _, _, _, px_bad = r1csqap.ComputePx(*inst.r1cs, w_bad, inst.m)
    wh, pxh = capi.scalars_upload(w_bad), capi.scalars_upload(px_bad)
    want_bad = groth16.prove_resident(pk, wh, pxh, r, s)
    t = [groth16.prove_witness_host_begin(pk, dr, x, r, s) for x in (inst.w_host, w_bad, inst.w_host)]
    w_bad_copy = w_bad.copy()
    w_bad[:] = 0                                                      # the caller's array was consumed by _begin
    res = [groth16.prove_end(x) for x in t]
    assert same(res[0], good) and same(res[1], want_bad) and same(res[2], good)
    assert capi.last_timing()["fallbacks"] == 0                        # (the last collected ticket was a good one)
    assert same(groth16.prove_from_witness_host(pk, dr, w_bad_copy, r, s), want_bad) and capi.last_timing()["fallbacks"] == 1
    assert same(groth16.prove_end(groth16.prove_host_begin(pk, w_bad_copy, px_bad, r, s)), want_bad)
    # gs_free of the key and the R1CS while a host ticket is in flight: deferred
    t1 = groth16.prove_witness_host_begin(pk, dr, inst.w_host, r, s)
    dr.handle.free()
    assert same(groth16.prove_end(t1), good)


@pytest.mark.parametrize("n", [200, 1 << 12])
def test_pinocchio_host_tickets_equal_the_resident_proofs(n):
    pin = synth.sqchain_pinocchio_instance(n, 0x7500 + n % 97)
    pk = pin.device_pk()
    dr = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
    want = snark.prove_resident(pk, pin.w, pin.px)
    eq = lambda p: all(getattr(p, k) == getattr(want, k) for k in snark.Proof.FIELDS)     # noqa: E731
    t = [snark.prove_host_begin(pk, pin.w_host, pin.px_host), snark.prove_witness_host_begin(pk, dr, pin.w_host),
         snark.prove_host_begin(pk, pin.w_host, pin.px_host)]
    assert all(eq(snark.prove_end(x)) for x in t)
    a0 = capi.alloc_counters()
    t = [snark.prove_witness_host_begin(pk, dr, pin.w_host), snark.prove_host_begin(pk, pin.w_host, pin.px_host),
         snark.prove_witness_host_begin(pk, dr, pin.w_host)]
    assert all(eq(snark.prove_end(x)) for x in t)
    assert capi.alloc_counters() == a0
    assert eq(snark.prove_from_witness_host(pk, dr, pin.w_host))
    assert snark.VerifyProof(pin.vk, snark.prove_end(snark.prove_witness_host_begin(pk, dr, pin.w_host)), pin.public)
    with pytest.raises(capi.GosnarkHipError) as e:
        snark.prove_host_begin(pk, pin.w_host[:-1], pin.px_host)
    assert e.value.code == -4


def test_streaming_prover_mirrors_submit_collect_in_order():
    """Round 6: the streaming drop-in (go/groth16hip.Prover / snarkhip.Prover; here their Python mirrors groth16.Prover / snark.Prover and
    GenerateProofsFromWitnessWithRS).  20 distinct witnesses through Submit / Collect with three in flight come back in submission order,
    each equal to the blocking proof of ITS witness (same r, s) and accepted by the verifier for its own public input only; a Submit on a
    full pipeline collects the oldest ticket into the done-queue instead of failing; px route and witness route give the same proofs; the
    Pinocchio twin does the same; Close cancels what is left and frees the slots."""
    n = 1 << 11
    inst = synth.sqchain_setup_instance(n, 0x7600)
    pk = inst.device_pk()
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    circ = groth16.Circuit(pk.nvars, pk.npublic)
    xs = synth.field_elems(20, 0x7601)
    ws = [synth.sqchain_witness(n, x) for x in xs]
    rs = [synth.field_elems(2, 0x7700 + k) for k in range(20)]
    want = [groth16.prove_from_witness_host(pk, dr, w, *rs[k]) for k, w in enumerate(ws)]
    p = groth16.NewProver(circ, pk, dr)
    got = []
    for k, w in enumerate(ws):
        p.SubmitWithRS(w, None, *rs[k])                      # never GS_ERR_BUSY: the fourth Submit parks the first proof in the done-queue
        assert p.InFlight() == k + 1 - len(got)
        if k % 5 == 4:
            while p.InFlight():
                got.append(p.Collect())
    assert len(got) == 20 and all(same(a, b) for a, b in zip(got, want))
    for k in (0, 7, 19):
        assert groth16.VerifyProof(inst.vk, got[k], [xs[k]]) and not groth16.VerifyProof(inst.vk, got[k], [xs[(k + 1) % 20]])
    with pytest.raises(ValueError):
        p.Collect()
    # the px route through the same prover object, ints instead of limb arrays for one of them, and the one-call forms
    h = capi.scalars_upload(ws[3])
    ph = dr.ComputePxResident(h)
    px3 = capi.scalars_download(ph)
    p.SubmitWithRS(ws[3], px3, *rs[3])
    p.SubmitWithRS(capi.u64_to_ints(ws[4]), None, *rs[4])
    assert same(p.Collect(), want[3]) and same(p.Collect(), want[4])
    assert same(groth16.GenerateProofsWithRS(circ, pk, ws[3], px3, *rs[3]), want[3])
    assert same(groth16.GenerateProofsFromWitnessWithRS(circ, pk, dr, ws[5], *rs[5]), want[5])
    # three tickets of ANOTHER owner occupy the device: the one-call forms fall back to the blocking entry points, the prover makes room by itself
    held = [groth16.prove_witness_host_begin(pk, dr, ws[k], *rs[k]) for k in range(3)]
    assert same(groth16.GenerateProofsFromWitnessWithRS(circ, pk, dr, ws[6], *rs[6]), want[6])
    assert same(groth16.GenerateProofsWithRS(circ, pk, ws[3], px3, *rs[3]), want[3])
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.NewProver(circ, pk, dr).SubmitWithRS(ws[7], None, *rs[7])     # nothing of its own to collect: the busy device is reported
    assert e.value.code == -6
    assert all(same(groth16.prove_end(t), want[k]) for k, t in enumerate(held))
    # Close abandons what is in flight; the slots are free again
    p.SubmitWithRS(ws[8], None, *rs[8]); p.SubmitWithRS(ws[9], None, *rs[9])
    p.Close()
    assert p.InFlight() == 0
    t = [groth16.prove_witness_host_begin(pk, dr, ws[k], *rs[k]) for k in range(3)]
    assert all(same(groth16.prove_end(x), want[k]) for k, x in enumerate(t))
    h.free(); ph.free()
    # snark.Prover
    pin = synth.sqchain_pinocchio_instance(n, 0x7602)
    ppk = pin.device_pk()
    pdr = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
    pwant = snark.prove_resident(ppk, pin.w, pin.px)
    sp = snark.NewProver(snark.Circuit(ppk.nvars, ppk.npublic) if hasattr(snark, "Circuit") else None, ppk, pdr)
    for _ in range(5):
        sp.Submit(pin.w_host)
    sp.Submit(pin.w_host, pin.px_host)
    outs = []
    while sp.InFlight():
        outs.append(sp.Collect())
    assert len(outs) == 6 and all(all(getattr(o, f) == getattr(pwant, f) for f in snark.Proof.FIELDS) for o in outs)
    assert all(getattr(snark.GenerateProofsFromWitness(None, ppk, pdr, pin.w_host), f) == getattr(pwant, f) for f in snark.Proof.FIELDS)
