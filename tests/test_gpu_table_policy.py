"""-m gpu: WHEN a base array gets its window table (gs_set_table_policy, round 5) -- and that it never matters for the result.

The reference proves once per key load (cli/main.go:330-349); rounds 1-4 spent ~140 ms and 15x the key's memory on window tables before
a 2^20 key's first proof.  Under the default policy `auto` a base array is summed TABLE-FREE (a bucket set per window, every window
adds the base point itself, the window sums recombined by Horner on the host) until its second use, then its table is built in the
background and the first call that finds it complete switches over.  tests/test_gpu_prove.py and tests/test_gpu_msm.py run every
parity test on both routes; here: the schedule itself, gs_build_tables, every table-free window width, eviction under a memory cap."""
import time

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi, groth16, snark, r1csqap, synth
import gpu_util as U
from oracle import c_oracle as C
from oracle import ref_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _init():
    capi.init()
    capi.set_table_policy("auto")
    yield
    capi.set_window_bits(0)
    capi.set_memory_limit(0)
    capi.set_table_policy("auto")


def same(p, q):
    return (p.PiA, p.PiB, p.PiC) == (q.PiA, q.PiB, q.PiC)


def wait_for_table_route(prove, free_width, timeout=30.0):
    """keep using the key until a call finds the background builds complete and switches to the window tables (gs_timing.window_bits
    is the width of the last plan: the table-free route's differs from the table route's at this size)"""
    t0 = time.time()
    while time.time() - t0 < timeout:
        prove()
        if capi.last_timing()["window_bits"] != free_width:
            return True
        time.sleep(0.02)
    return False


def test_auto_first_proof_builds_nothing_then_tables_arrive_in_the_background():
    n = 1 << 14
    inst = synth.sqchain_setup_instance(n, 0x8100)
    pk = inst.device_pk()
    r, s = synth.field_elems(2, 81)
    obj_b, tab_b = capi.handle_bytes(pk.handle)
    assert tab_b == 0
    first = groth16.prove_resident(pk, inst.w, inst.px, r, s)                 # table-free
    assert capi.handle_bytes(pk.handle)[1] == 0
    free_width = capi.last_timing()["window_bits"]
    assert capi.TABLE_POLICY["auto"] == 0 and 9 <= free_width <= 16
    a, b, c = inst.expected_proof_scalars(r, s)
    assert (first.PiA[0], first.PiA[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, a))
    assert (first.PiB[0], first.PiB[1]) == C.g2_affine(C.g2_mul_scalar(O.G2_GEN, b))
    assert (first.PiC[0], first.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, c))
    # second use: still table-free, but the builds start; soon a call finds them and the key holds >= 8 rows of every array it used
    second = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    assert same(second, first)
    assert wait_for_table_route(lambda: groth16.prove_resident(pk, inst.w, inst.px, r, s), free_width)
    tabled = groth16.prove_resident(pk, inst.w, inst.px, r, s)
    table_width = capi.last_timing()["window_bits"]
    assert same(tabled, first) and table_width != free_width and capi.handle_bytes(pk.handle)[1] >= 8 * 5 * n * 64
    # pipelined tickets across the switch-over, the witness route (its evaluation-basis array gets its table the same way)
    dr = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    for _ in range(6):
        t = [groth16.prove_witness_begin(pk, dr, inst.w, r, s) for _ in range(3)]
        assert all(same(groth16.prove_end(x), first) for x in t)
    # release: back to table-free at once, same proof; gs_build_tables: blocking, whatever the policy
    capi.release_tables(pk.handle)
    assert capi.handle_bytes(pk.handle)[1] == 0 and same(groth16.prove_resident(pk, inst.w, inst.px, r, s), first)
    capi.set_table_policy("never")
    capi.build_tables(pk.handle, 1)
    tab_px = capi.handle_bytes(pk.handle)[1]
    capi.build_tables(pk.handle, 0)
    assert capi.handle_bytes(pk.handle)[1] > tab_px >= 8 * 5 * n * 64
    assert same(groth16.prove_resident(pk, inst.w, inst.px, r, s), first) and capi.last_timing()["window_bits"] == table_width
    assert same(groth16.prove_from_witness(pk, dr, inst.w, r, s), first)


@pytest.mark.parametrize("c", [9, 10, 11, 12, 13, 14, 15, 16, 8, 20])
def test_every_table_free_window_width_gives_the_same_msm_and_proof(c):
    """gs_set_window_bits on the table-free route (9..16; 8 and 20 are clamped): G1 and G2 MSMs against the closed form of random
    scalars on bases k_i G (sum = (sum k_i s_i) G), ragged term counts incl. 1 and 2, and a proof."""
    capi.set_table_policy("never")
    capi.set_window_bits(c)
    for n, g2 in ((1, False), (2, True), (300, False), (5000, True), (70001, False)):
        ks, sc = U.rand_scalars_u64(n, 900 + n), U.rand_scalars_u64(n, 901 + n)
        bases = capi.g2_fixed_base(ks) if g2 else capi.g1_fixed_base(ks)
        tot = sum(k * s for k, s in zip(U.u64_rows_to_ints(ks), U.u64_rows_to_ints(sc))) % O.R
        want = C.g2_affine(C.g2_mul_scalar(O.G2_GEN, tot)) if g2 else C.g1_affine(C.g1_mul_scalar(O.G1_GEN, tot))
        assert capi.msm(bases, sc, g2=g2) == want, (n, g2)
        assert capi.handle_bytes(bases)[1] == 0
        h = capi.scalars_upload(sc)
        assert capi.msm_end(capi.msm_begin(bases, h, n, g2=g2)) == want
        assert min(max(c, 9), 16) == capi.last_timing()["window_bits"]
    inst = synth.sqchain_setup_instance(3001, 0x8200 + c)
    r, s = synth.field_elems(2, 82)
    got = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    a, b, cc = inst.expected_proof_scalars(r, s)
    assert (got.PiC[0], got.PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, cc))
    assert groth16.VerifyProof(inst.vk, got, capi.u64_to_ints(inst.w_host[1:2]))


def test_table_free_handles_heavy_buckets_zeros_and_infinities():
    """0/1-heavy scalars (one bucket of window 0 holds almost every term: the heavy-bucket tree), zero scalars, duplicate and negated
    bases, infinities in the base array -- table-free, against the table route."""
    n = 1 << 15
    rng = np.random.Generator(np.random.PCG64(5))
    sc = U.rand_scalars_u64(n, 77)
    kind = rng.integers(0, 10, size=n)
    sc[kind < 4] = (0, 0, 0, 0)
    sc[(kind >= 4) & (kind < 8)] = (1, 0, 0, 0)
    sc[kind == 8, 1:] = 0                                  # below 2^64
    ks = U.rand_scalars_u64(n, 78)
    ks[::7] = ks[3]                                        # duplicate bases
    ks[5::11] = (0, 0, 0, 0)                               # k = 0: the point at infinity in the array
    res = {}
    for policy in ("always", "never"):
        capi.set_table_policy(policy)
        b1, b2 = capi.g1_fixed_base(ks), capi.g2_fixed_base(ks[: n // 4])
        res[policy] = (capi.msm(b1, sc), capi.msm(b2, sc[: n // 4], g2=True), capi.last_timing()["heavy_buckets"])
    assert res["always"][:2] == res["never"][:2] and res["never"][2] >= 1
    tot = sum(k * s for k, s in zip(U.u64_rows_to_ints(ks), U.u64_rows_to_ints(sc))) % O.R
    assert res["never"][0] == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, tot))


def test_pinocchio_and_sharded_partials_table_free_equal_the_table_route():
    n = 2000
    pin = synth.sqchain_pinocchio_instance(n, 0x8300)
    inst = synth.sqchain_setup_instance(n, 0x8301)
    r, s = synth.field_elems(2, 83)
    res = {}
    for policy in ("always", "never"):
        capi.set_table_policy(policy)
        capi.release_tables(pin.device_pk().handle)
        capi.release_tables(inst.device_pk().handle)
        p = snark.prove_resident(pin.device_pk(), pin.w, pin.px)
        parts = [groth16.prove_partials(inst.device_pk(), inst.w, inst.px, k, 3)[0] for k in range(3)]
        res[policy] = ([getattr(p, k) for k in snark.Proof.FIELDS], parts)
        assert snark.VerifyProof(pin.vk, p, pin.public)
        assert (capi.handle_bytes(pin.device_pk().handle)[1] > 0) == (policy == "always")
    assert res["always"] == res["never"]


def test_out_of_memory_evicts_idle_tables_instead_of_failing():
    """keys whose tables do not fit together prove round-robin under a cap (gs_set_memory_limit); every proof
    equals its closed form, gs_memory_query shows the evictions, a ticket's key is never the victim."""
    capi.set_table_policy("always")
    n = 1 << 12
    insts = [synth.sqchain_setup_instance(n, 0x8400 + i) for i in range(3)]
    rs = [synth.field_elems(2, 840 + i) for i in range(3)]
    want = [groth16.prove_resident(k.device_pk(), k.w, k.px, *rs[i]) for i, k in enumerate(insts)]
    tab = capi.handle_bytes(insts[0].device_pk().handle)[1]
    assert tab > 0 and all(capi.handle_bytes(k.device_pk().handle)[1] == tab for k in insts)
    for k in insts[1:]:
        capi.release_tables(k.device_pk().handle)
    base = capi.memory_query()
    capi.set_memory_limit(base["library_bytes"] + tab // 3)             # room for ONE key's tables (and a third of another)
    for lap in range(3):
        for i, k in enumerate(insts):
            got = groth16.prove_resident(k.device_pk(), k.w, k.px, *rs[i])
            assert same(got, want[i]), (lap, i)
            held = [capi.handle_bytes(x.device_pk().handle)[1] for x in insts]
            now = capi.memory_query()
            # the prover's own tables are complete, the cap holds, and the three keys together never hold two full sets
            assert held[i] == tab and now["library_bytes"] <= base["library_bytes"] + tab // 3 and sum(held) < 2 * tab, (lap, i, held, tab)
    m = capi.memory_query()
    assert m["evictions"] >= base["evictions"] + 8 and m["library_bytes"] <= base["library_bytes"] + tab // 3
    # a ticket holds key 2 (the one with tables now): key 0 cannot take them -> a clean GS_ERR_HIP, and the ticket is unharmed
    t = groth16.prove_begin(insts[2].device_pk(), insts[2].w, insts[2].px, *rs[2])
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.prove_resident(insts[0].device_pk(), insts[0].w, insts[0].px, *rs[0])
    assert e.value.code == -2
    assert same(groth16.prove_end(t), want[2])
    # ... while under `auto` the same call simply goes table-free (no tables to build) -- given room for its bucket sets
    capi.set_table_policy("auto")
    capi.set_memory_limit(capi.memory_query()["library_bytes"] + (256 << 20))
    t = groth16.prove_begin(insts[2].device_pk(), insts[2].w, insts[2].px, *rs[2])
    assert same(groth16.prove_resident(insts[0].device_pk(), insts[0].w, insts[0].px, *rs[0]), want[0])
    assert same(groth16.prove_end(t), want[2])
    closed = insts[0].expected_proof_scalars(*rs[0])
    assert (want[0].PiC[0], want[0].PiC[1]) == C.g1_affine(C.g1_mul_scalar(O.G1_GEN, closed[2]))
