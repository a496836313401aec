"""-m gpu: several logical devices behind the C ABI (multi.hip), the RCCL gather, deferred frees and concurrent callers.

Runs after the single-device modules (it re-initialises the library with EIGHT logical devices that all name GPU 0 -- how
BASELINE configs[3] / [4] are exercised on a 1-GPU box) and hands a single-device library back to test_gpu_zz_lifecycle."""
import ctypes
import os
import threading

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi, groth16, parallel, synth
import gpu_util as U

pytestmark = pytest.mark.gpu

NLOG = 8


@pytest.fixture(scope="module", autouse=True)
def eight_logical_devices():
    capi.shutdown()
    capi.init([0] * NLOG)
    capi.set_table_policy("always")        # (the routes themselves: tests/test_gpu_prove.py runs every test on both)
    assert capi.device_count() == NLOG
    yield
    capi.comm_destroy()
    capi.shutdown()
    capi.init()


@pytest.fixture(scope="module")
def inst():
    capi.set_device(0)
    return synth.sqchain_setup_instance(1 << 12, 0xD1CE)


def _spread(inst, n, full_keys=False):
    """key slices (or full replicas) + witness / px replicas on logical devices 0..n-1"""
    full = inst.device_pk()
    pks = [groth16.ShardPkTo(full, 0, 1, d) if full_keys else groth16.ShardPkTo(full, d, n, d) for d in range(n)]
    ws = [capi.scalars_clone(inst.w, d) for d in range(n)]
    pxs = [capi.scalars_clone(inst.px, d) for d in range(n)]
    for d in range(n):
        assert capi.handle_device(pks[d].handle) == d and capi.handle_device(ws[d]) == d
    return pks, ws, pxs


def _same(p, q):
    return (p.PiA, p.PiB, p.PiC) == (q.PiA, q.PiB, q.PiC)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_one_proof_over_n_logical_devices_is_the_single_device_proof(inst, n):
    """the same physical device listed N = 2, 3, 8 times.  Key slices (each logical device holds 1/N of
    every key array) and full replicas; host-memory exchange (no communicator yet)."""
    r, s = synth.field_elems(2, 77 + n)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    assert inst.vk is not None and groth16.VerifyProof(inst.vk, want, capi.u64_to_ints(inst.w_host[1:2]))
    capi.comm_destroy()
    for full_keys in (False, True):
        pks, ws, pxs = _spread(inst, n, full_keys)
        got, used = groth16.prove_multi(pks, ws, pxs, r, s)
        assert _same(got, want) and used is False


def test_records_travel_through_rccl_allgather_with_a_local_communicator(inst):
    """(b): gs_comm_init_local = ncclCommInitAll over the distinct physical devices (one here): the eight 416-byte records of a
    proof and the partial points of an MSM pass through ncclAllGather before they are added."""
    capi.comm_destroy()
    capi.comm_init_local()
    info = capi.comm_info()
    assert info["nranks"] == 1 and info["local"] is True
    before = info["collectives"]
    r, s = synth.field_elems(2, 99)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    pks, ws, pxs = _spread(inst, NLOG)
    got, used = groth16.prove_multi(pks, ws, pxs, r, s)
    assert _same(got, want) and used is True
    assert capi.comm_info()["collectives"] == before + 1
    # raw byte gather: 8 blocks in, the same 8 blocks out
    blocks = bytes(range(256)) * 13
    assert capi.comm_allgather(blocks, 1) == blocks
    capi.comm_destroy()
    assert capi.comm_info()["nranks"] == 0


@pytest.mark.parametrize("g2", [False, True])
def test_msm_sharded_over_logical_devices(g2):
    """configs[3] in miniature: a ragged 3-way split of the term range, each shard resident on its own logical device."""
    capi.set_device(0)
    n = 5001 if not g2 else 1203
    ks, sc = U.rand_scalars_u64(n, 5), U.rand_scalars_u64(n, 6)
    bases = capi.g2_fixed_base(ks) if g2 else capi.g1_fixed_base(ks)
    scal = capi.scalars_upload(sc)
    want = capi.msm_resident(bases, scal, n, g2=g2)
    clone = capi.g2_clone if g2 else capi.g1_clone
    for ndev, with_comm in ((3, False), (8, True)):
        capi.comm_destroy()
        if with_comm:
            capi.comm_init_local()
        bs, ss = [], []
        for d in range(ndev):
            lo, hi = parallel.shard_range(n, ndev, d)
            bs.append(clone(bases, d, lo, hi - lo))
            ss.append(capi.scalars_clone(scal, d, lo, hi - lo))
        got, used = capi.msm_multi(bs, ss, g2=g2)
        assert got == want and used is with_comm
    capi.comm_destroy()


def test_one_process_per_gpu_gather_inside_the_library_world_1(inst):
    """The deployment bench.py uses (one process per GPU): gs_comm_unique_id -> gs_comm_init_rank -> gs_groth16_prove_sharded /
    gs_msm_g1_sharded.  World size 1 here: the 1-rank communicator is created and ncclAllGather runs."""
    capi.comm_destroy()
    capi.set_device(0)
    capi.comm_init_rank(capi.comm_unique_id(), 1, 0)
    info = capi.comm_info()
    assert (info["nranks"], info["rank"], info["local"]) == (1, 0, False)
    r, s = synth.field_elems(2, 123)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    got = groth16.prove_sharded_rccl(inst.device_pk(), inst.w, inst.px, r, s)
    assert _same(got, want)
    n = 3000
    bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 7))
    scal = capi.scalars_upload(U.rand_scalars_u64(n, 8))
    assert capi.msm_sharded(bases, scal) == capi.msm_resident(bases, scal, n)
    assert capi.comm_info()["collectives"] == info["collectives"] + 2
    with pytest.raises(capi.GosnarkHipError, match="already exists"):
        capi.comm_init_local()
    capi.comm_destroy()


@pytest.mark.parametrize("n", [2, 3, 8])
def test_values_route_polynomial_stage_on_one_owner_then_scattered(inst, n):
    """strong scaling without the replicated H(x).  The proof's owner (they take turns: owner = proof index mod n)
    turns the witness into H's n values ONCE (gs_groth16_witness_values), scatters slice d to logical device d, and every device sums
    only its term ranges (gs_groth16_prove_multi_values): the single-device proof, with key slices that carry their share of the
    evaluation-basis array.  A violated constraint is reported by the owner instead of producing values."""
    from gosnark_amd import r1csqap
    capi.comm_destroy()
    full = inst.device_pk()
    assert capi.pk_eval_count(full.handle) == inst.n
    pks = [groth16.ShardPkTo(full, d, n, d) for d in range(n)]
    assert sum(capi.pk_eval_count(k.handle) for k in pks) == inst.n
    ws = [capi.scalars_clone(inst.w, d) for d in range(n)]
    for proof in range(3):
        owner = proof % n
        r, s = synth.field_elems(2, 500 + 10 * n + proof)
        want = groth16.prove_resident(full, inst.w, inst.px, r, s)
        capi.set_device(owner)
        try:
            dev = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)          # the owner needs the sparse system (every rank keeps it: they rotate)
        finally:
            capi.set_device(0)
        hv, bad = groth16.witness_values(pks[owner], dev, ws[owner])
        assert bad == 0 and len(hv) == inst.n and capi.handle_device(hv) == owner
        slices = groth16.scatter_values(hv, n)
        got, used = groth16.prove_multi_values(pks, ws, slices, r, s)
        assert _same(got, want) and used is False
        if proof == 0:                                            # ... and piecewise, as one process per GPU would
            parts = [groth16.prove_partials_values(pks[d], ws[d], slices[d], d, n)[0] for d in range(n)]
            combined = [capi.sum_affine([parts[k][i] for k in range(n)], g2=g2) for i, g2 in enumerate(groth16.SUM_IS_G2)]
            assert _same(groth16.finish(pks[0], combined, r, s), want)
            # ... and pipelined: three shard tickets of device 0 in flight; a proof ticket's collector refuses them
            t = [groth16.partials_values_begin(pks[0], ws[0], slices[0], 0, n) for _ in range(3)]
            with pytest.raises(capi.GosnarkHipError, match="partial-sums"):
                groth16.prove_end(t[0])
            assert all(groth16.partials_end(x) == parts[0] for x in t)
            with pytest.raises(capi.GosnarkHipError, match="covers"):
                groth16.prove_partials_values(pks[0], ws[0], capi.scalars_clone(hv, 0), 0, n)       # all n values where a slice belongs
    w_bad = inst.w_host.copy()
    w_bad[5] = (4242, 0, 0, 0)
    _, bad = groth16.witness_values(pks[0], r1csqap.DeviceR1CS(*inst.r1cs, inst.m), capi.scalars_upload(w_bad))
    assert bad > 0


def test_values_route_one_process_per_gpu_world_1(inst):
    """The rank-mode entry points of the values route at world size 1: gs_scalars_scatter (ncclSend / ncclRecv group, here only the
    root's own copy) and gs_groth16_prove_sharded_values (record gather through the 1-rank communicator)."""
    from gosnark_amd import r1csqap
    capi.comm_destroy()
    capi.set_device(0)
    capi.comm_init_rank(capi.comm_unique_id(), 1, 0)
    before = capi.comm_info()["collectives"]
    r, s = synth.field_elems(2, 321)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    hv, bad = groth16.witness_values(inst.device_pk(), r1csqap.DeviceR1CS(*inst.r1cs, inst.m), inst.w)
    assert bad == 0
    mine = capi.scalars_scatter(hv, inst.n, 0)
    assert len(mine) == inst.n and np.array_equal(capi.scalars_download(mine), capi.scalars_download(hv))
    got = groth16.prove_sharded_values_rccl(inst.device_pk(), inst.w, mine, r, s)
    assert _same(got, want)
    assert capi.comm_info()["collectives"] == before + 2
    capi.comm_destroy()


@pytest.fixture(scope="module")
def pin():
    capi.set_device(0)
    return synth.sqchain_pinocchio_instance(3000, 0xD1CF)        # ragged: 3001 variables, 3000 constraints over 2, 3, 8 shards


def _same_pin(p, q):
    from gosnark_amd import snark
    return all(getattr(p, k) == getattr(q, k) for k in snark.Proof.FIELDS)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_pinocchio_proof_over_n_logical_devices_is_the_single_device_proof(pin, n):
    """SURVEY 8e applied to snark.GenerateProofs (snark.go:254-289): a Pinocchio proof is eight plain sums, so the ranks' eight
    partial points add up to it.  Key slices (1/N of the seven per-variable arrays, of G1T and of the evaluation-basis array) and full
    replicas; px route and values route (the polynomial stage on ONE owner, its slices scattered); piecewise through
    gs_pinocchio_prove_partials + gs_pinocchio_combine as one process per GPU would; slices refuse the wrong shard."""
    from gosnark_amd import r1csqap, snark
    capi.comm_destroy()
    full = pin.device_pk()
    want = snark.prove_resident(full, pin.w, pin.px)
    assert snark.VerifyProof(pin.vk, want, capi.u64_to_ints(pin.w_host[1:2]))
    ws = [capi.scalars_clone(pin.w, d) for d in range(n)]
    pxs = [capi.scalars_clone(pin.px, d) for d in range(n)]
    slices_pk = [snark.ShardPk(full, d, n, d) for d in range(n)]
    replicas = [snark.ShardPk(full, 0, 1, d) for d in range(n)]
    assert sum(capi.pk_eval_count(k.handle) for k in slices_pk) == pin.n == capi.pk_eval_count(full.handle)
    assert all(capi.handle_device(slices_pk[d].handle) == d for d in range(n))
    for pks in (slices_pk, replicas):
        got, used = snark.prove_multi(pks, ws, pxs)
        assert _same_pin(got, want) and used is False
    recs = [snark.prove_partials(slices_pk[d], ws[d], pxs[d], d, n) for d in range(n)]
    assert _same_pin(snark.combine(recs), want)
    with pytest.raises(capi.GosnarkHipError, match="holds shard"):
        snark.prove_partials(slices_pk[1], ws[1], pxs[1], 0, n)
    with pytest.raises(capi.GosnarkHipError, match="slice"):
        snark.ShardPk(slices_pk[0], 0, 2)
    # values route: owner = proof index mod n
    for proof in range(2):
        owner = (proof + 1) % n
        capi.set_device(owner)
        try:
            dev = r1csqap.DeviceR1CS(*pin.r1cs, pin.m)
        finally:
            capi.set_device(0)
        hv, bad = snark.witness_values(slices_pk[owner], dev, ws[owner])
        assert bad == 0 and len(hv) == pin.n and capi.handle_device(hv) == owner
        hs = groth16.scatter_values(hv, n)
        got, used = snark.prove_multi(slices_pk, ws, hs, values=True)
        assert _same_pin(got, want) and used is False
    recs = [snark.prove_partials_values(replicas[d], ws[d], hs[d], d, n) for d in range(n)]
    assert _same_pin(snark.combine(recs), want)
    with pytest.raises(capi.GosnarkHipError, match="covers"):
        snark.prove_partials_values(slices_pk[0], ws[0], capi.scalars_clone(hv, 0), 0, n)
    w_bad = pin.w_host.copy()
    w_bad[7] = (99, 0, 0, 0)
    _, bad = snark.witness_values(full, r1csqap.DeviceR1CS(*pin.r1cs, pin.m), capi.scalars_upload(w_bad))
    assert bad > 0


def test_pinocchio_records_through_rccl_and_rank_mode_world_1(pin):
    """The 616-byte Pinocchio records through ncclAllGather: a local communicator (one rank per distinct physical device; here one)
    with the logical devices spread evenly over it, then the one-process-per-GPU entry points at world size 1, px and values route."""
    from gosnark_amd import r1csqap, snark
    full = pin.device_pk()
    want = snark.prove_resident(full, pin.w, pin.px)
    capi.comm_destroy()
    capi.comm_init_local()
    n = 4
    pks = [snark.ShardPk(full, d, n, d) for d in range(n)]
    ws = [capi.scalars_clone(pin.w, d) for d in range(n)]
    pxs = [capi.scalars_clone(pin.px, d) for d in range(n)]
    before = capi.comm_info()["collectives"]
    got, used = snark.prove_multi(pks, ws, pxs)
    assert _same_pin(got, want) and used is True and capi.comm_info()["collectives"] == before + 1
    capi.comm_destroy()
    capi.set_device(0)
    capi.comm_init_rank(capi.comm_unique_id(), 1, 0)
    before = capi.comm_info()["collectives"]
    assert _same_pin(snark.prove_sharded_rccl(full, pin.w, pin.px), want)
    hv, bad = snark.witness_values(full, r1csqap.DeviceR1CS(*pin.r1cs, pin.m), pin.w)
    mine = capi.scalars_scatter(hv, pin.n, 0)
    assert bad == 0 and _same_pin(snark.prove_sharded_rccl(full, pin.w, mine, values=True), want)
    assert capi.comm_info()["collectives"] == before + 3
    capi.comm_destroy()
    with pytest.raises(capi.GosnarkHipError, match="communicator"):
        snark.prove_sharded_rccl(full, pin.w, pin.px)


def test_pinocchio_batch_of_independent_proofs_round_robin_over_devices(pin):
    """gs_pinocchio_prove_batch: 12 proofs with distinct witnesses over 4 logical devices, three in flight per device, equal to
    the single-device proofs one by one; a bad entry leaves no ticket behind."""
    from gosnark_amd import r1csqap, snark
    ndev, nproofs = 4, 12
    full = pin.device_pk()
    keys = [snark.ShardPk(full, 0, 1, d) for d in range(ndev)]
    hosts = [synth.sqchain_witness(pin.n, 1000 + i) for i in range(nproofs)]
    ws, pxs, want = [], [], []
    for i, wh in enumerate(hosts):
        _, _, _, px = r1csqap.ComputePx(*pin.r1cs, wh, pin.m)
        w0, p0 = capi.scalars_upload(wh), capi.scalars_upload(px)
        want.append(snark.prove_resident(full, w0, p0))
        ws.append(capi.scalars_clone(w0, i % ndev)); pxs.append(capi.scalars_clone(p0, i % ndev))
    assert len({w.PiA for w in want}) == nproofs
    got = snark.prove_batch(keys, ws, pxs)
    assert all(_same_pin(g, w) for g, w in zip(got, want))
    bad_px = list(pxs)
    bad_px[5] = capi.scalars_clone(capi.scalars_upload(np.zeros((3 * pin.n, 4), dtype=np.uint64)), 5 % ndev)    # len(hx) > len(G1T)
    with pytest.raises(capi.GosnarkHipError, match="G1T"):
        snark.prove_batch(keys, ws, bad_px)
    assert all(_same_pin(g, w) for g, w in zip(snark.prove_batch(keys, ws, pxs), want))     # every slot is free again


def test_torch_distributed_nccl_branch_runs_at_world_1(inst, monkeypatch):
    """parallel.allgather_points over torch.distributed's nccl (= RCCL) backend, forced at world size 1."""
    import torch
    import torch.distributed as dist
    monkeypatch.setenv("GS_FORCE_COLLECTIVE", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29641")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        r, s = synth.field_elems(2, 321)
        want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
        got = groth16.prove_sharded(inst.device_pk(), inst.w, inst.px, r, s)
        assert _same(got, want)
    finally:
        dist.destroy_process_group()


def test_batch_of_independent_proofs_round_robin_over_devices(inst):
    """configs[4] in miniature: 11 proofs (two different witnesses, fresh randomness each) over 4 logical devices, one full key
    per device, three in flight per device, no collective."""
    from gosnark_amd import r1csqap
    ndev, nproofs = 4, 11
    capi.set_device(0)
    _, _, _, w2 = synth.sqchain_r1cs(inst.n, 31337)
    _, _, _, px2 = r1csqap.ComputePx(*inst.r1cs, w2, inst.m)
    w2h, px2h = capi.scalars_upload(w2), capi.scalars_upload(px2)
    rs = [tuple(synth.field_elems(2, 900 + i)) for i in range(nproofs)]
    src = [(inst.w, inst.px) if i % 3 else (w2h, px2h) for i in range(nproofs)]
    want = [groth16.prove_resident(inst.device_pk(), w, px, r, s) for (w, px), (r, s) in zip(src, rs)]
    pks = [groth16.ShardPkTo(inst.device_pk(), 0, 1, d) for d in range(ndev)]
    ws = [capi.scalars_clone(w, i % ndev) for i, (w, _) in enumerate(src)]
    pxs = [capi.scalars_clone(px, i % ndev) for i, (_, px) in enumerate(src)]
    got = groth16.prove_batch(pks, ws, pxs, rs)
    assert len(got) == nproofs and all(_same(g, w) for g, w in zip(got, want))
    assert groth16.VerifyProof(inst.vk, got[0], capi.u64_to_ints(w2[1:2]))
    with pytest.raises(capi.GosnarkHipError, match="do not share one logical device"):
        groth16.prove_batch(pks, [ws[1]], [pxs[0]], rs[:1])


def test_handles_route_to_their_device_and_foreign_handles_are_refused(inst):
    capi.set_device(3)
    assert capi.get_device() == 3
    sc = capi.scalars_upload(U.rand_scalars_u64(100, 1))
    bases = capi.g1_fixed_base(U.rand_scalars_u64(100, 2))
    assert capi.handle_device(sc) == 3 and capi.handle_device(bases) == 3
    capi.set_device(0)                                               # the handle, not the thread's device, decides
    got = capi.msm_resident(bases, sc, 100)
    on0 = capi.scalars_clone(sc, 0)
    b0 = capi.g1_clone(bases, 0)
    assert capi.msm_resident(b0, on0, 100) == got
    assert (capi.scalars_download(on0) == capi.scalars_download(sc)).all()
    with pytest.raises(capi.GosnarkHipError):                        # scalars on device 0, bases on device 3
        capi.msm_resident(bases, on0, 100)
    with pytest.raises(capi.GosnarkHipError, match="no logical device"):
        capi.set_device(NLOG)
    with pytest.raises(capi.GosnarkHipError):
        capi.scalars_clone(sc, NLOG)


def test_free_while_tickets_are_outstanding_is_deferred(inst):
    """gs_free of the key / w / px a ticket reads must not fail and must not pull the objects from under the
    proof; other entry points keep working meanwhile (they queue behind the outstanding device work)."""
    capi.set_device(0)
    r, s = synth.field_elems(2, 55)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    pk = groth16.ShardPkTo(inst.device_pk(), 0, 1, 0)
    w, px = capi.scalars_clone(inst.w, 0), capi.scalars_clone(inst.px, 0)
    tickets = [groth16.prove_begin(pk, w, px, r, s) for _ in range(3)]
    with pytest.raises(capi.GosnarkHipError) as e:
        groth16.prove_begin(pk, w, px, r, s)
    assert e.value.code == -6                                         # GS_ERR_BUSY: a bounded queue, not a broken library
    pk.handle.free(); w.free(); px.free()
    assert pk.handle.h == 0 and w.h == 0 and px.h == 0
    # a blocking MSM and an upload while three proofs are in flight
    bases = capi.g1_fixed_base(U.rand_scalars_u64(500, 3))
    sc = U.rand_scalars_u64(500, 4)
    m1 = capi.msm(bases, sc)
    fresh = capi.scalars_upload(sc)
    assert capi.msm_resident(bases, fresh, 500) == m1
    for t in tickets:
        assert _same(groth16.prove_end(t), want)


def test_concurrent_callers_pipelined_proofs_uploads_and_msms(inst):
    """one thread streams pipelined proofs while two others upload witnesses, compute px from the resident
    R1CS and run MSMs on the same logical device; every result equals the serial one."""
    from gosnark_amd import r1csqap
    capi.set_device(0)
    r, s = synth.field_elems(2, 66)
    want = groth16.prove_resident(inst.device_pk(), inst.w, inst.px, r, s)
    n = 2000
    bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 11))
    scs = [U.rand_scalars_u64(n, 20 + i) for i in range(6)]
    want_msm = [capi.msm(bases, sc) for sc in scs]
    dr1cs = r1csqap.DeviceR1CS(*inst.r1cs, inst.m)
    want_px = capi.scalars_download(dr1cs.ComputePxResident(inst.w))
    errors, proofs, msms, pxs = [], [], [None] * len(scs), []

    def prover():
        try:
            capi.set_device(0)
            tickets = []
            for _ in range(12):
                tickets.append(groth16.prove_begin(inst.device_pk(), inst.w, inst.px, r, s))
                if len(tickets) == 3:
                    proofs.append(groth16.prove_end(tickets.pop(0)))
            while tickets:
                proofs.append(groth16.prove_end(tickets.pop(0)))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    def msm_worker(lo, hi):
        try:
            capi.set_device(0)
            for i in range(lo, hi):
                h = capi.scalars_upload(scs[i])
                msms[i] = capi.msm_resident(bases, h, n)
                h.free()
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    def px_worker():
        try:
            capi.set_device(0)
            for _ in range(3):
                w = capi.scalars_upload(inst.w_host)
                pxs.append(capi.scalars_download(dr1cs.ComputePxResident(w)))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=prover), threading.Thread(target=msm_worker, args=(0, 3)), threading.Thread(target=msm_worker, args=(3, 6)),
          threading.Thread(target=px_worker)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert len(proofs) == 12 and all(_same(p, want) for p in proofs)
    assert msms == want_msm
    assert len(pxs) == 3 and all((p == want_px).all() for p in pxs)


def test_plain_c_process_proves_over_three_logical_devices(tmp_path):
    """(c): the plain-C driver (no Python, no torch in the process) lists GPU 0 three times, cuts the reference's x^3 + x + 5
    key into slices, creates the local RCCL communicator and reproduces the single-device proof with gs_groth16_prove_multi."""
    import c_util
    import golden_util as GU
    blob = c_util.write_groth_instance(tmp_path, GU.load("groth_x3"))
    out = c_util.build_and_run("multi_device.c", [str(blob)], tmp_path)
    assert out.strip().endswith("OK"), out
    assert "used_rccl=1" in out and "collectives=1" in out


def test_plain_c_process_proves_pinocchio_over_three_logical_devices(tmp_path):
    """go/gosnarkhip/pinocchio_multi.go's call sequences as tests/c/snark_multi_device.c, on the reference's own wasm/index.js
    Pinocchio fixture (m = 8 variables over 3 slices: ragged): gs_pinocchio_prove_multi through the local RCCL communicator,
    partial records + gs_pinocchio_combine, gs_pinocchio_prove_batch -- each byte-identical to gs_pinocchio_prove, verifier accepts."""
    import c_util
    import golden_util as GU
    rec = GU.load("pinocchio_x3_fixture")
    public = [x for x in rec["w"][1:1 + rec["circuit"]["NPublic"]]]
    blob = c_util.write_pinocchio_instance(tmp_path, rec, public)
    out = c_util.build_and_run("snark_multi_device.c", [str(blob)], tmp_path)
    assert out.strip().endswith("OK"), out
    assert "used_rccl=1" in out and "collectives=1" in out


def test_config_4_2p22_msm_over_8_logical_devices_equals_the_naive_loop_golden():
    """BASELINE configs[3]: 'Synthetic 2^22-term MSM sharded over 8 GPUs' -- here over 8 logical devices of the one GPU, with the
    record exchange through the in-library ncclAllGather: shard d holds terms [d 2^19, (d + 1) 2^19).  The expected point was
    computed offline by the C restatement of the reference's MulScalar / Add loop over all 2^22 terms on all host cores
    (oracle/gen_golden_large.py msm22 -> tests/golden/oracle_msm_g1_2p22.json); the single-device MSM gives it too."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_msm_g1_2p22.json")) as f:
        rec = json.load(f)
    n, S = rec["n"], 8
    want = (int(rec["x"]), int(rec["y"]))
    capi.set_device(0)
    bases = capi.g1_fixed_base(synth.scalars_u64(n, rec["seed_bases"]))
    sc = capi.scalars_upload(synth.scalars_u64(n, rec["seed_scalars"]))
    assert capi.msm_resident(bases, sc, n) == want
    capi.comm_init_local()
    per = n // S
    bs = [capi.g1_clone(bases, d, d * per, per) for d in range(S)]
    ss = [capi.scalars_clone(sc, d, d * per, per) for d in range(S)]
    got, used_rccl = capi.msm_multi(bs, ss)
    assert got == want and used_rccl
    for h in bs + ss + [bases, sc]:
        h.free()
    capi.comm_destroy()


def test_config_5_2p18_proofs_round_robin_over_8_logical_devices_against_the_golden():
    """BASELINE configs[4] in its own size class: independent 2^18-constraint proofs round-robin over 8 devices (here 8 logical devices of
    the one GPU), one full key per device, three in flight per device, no collective.  The instance is synth.QuotientInstance(2^18,
    seed), whose proof for the golden's (r, s) was computed outside the library (oracle/gen_golden_large.py prove20 18 ->
    tests/golden/oracle_groth_quotient_2p18.json: naive-loop MSMs, px by oracle/crt_ntt.py): every device produces exactly it; the
    proofs with other randomness equal the single-device blocking ones."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_groth_quotient_2p18.json")) as f:
        rec = json.load(f)
    capi.set_device(0)
    q = synth.quotient_instance(rec["n"], rec["seed"])
    assert q.px_sha256 == rec["px_sha256"]
    want = ((int(rec["PiA"][0]), int(rec["PiA"][1]), 1),
            ((int(rec["PiB"][0][0]), int(rec["PiB"][0][1])), (int(rec["PiB"][1][0]), int(rec["PiB"][1][1])), (1, 0)),
            (int(rec["PiC"][0]), int(rec["PiC"][1]), 1))
    ndev, nproofs = 8, 16
    golden_rs = (int(rec["r"]), int(rec["s"]))
    rs = [golden_rs if i < ndev else tuple(synth.field_elems(2, 700 + i)) for i in range(nproofs)]       # the first round: one golden proof per device
    pks = [groth16.ShardPkTo(q.device_pk(), 0, 1, d) for d in range(ndev)]
    ws = [capi.scalars_clone(q.w, i % ndev) for i in range(nproofs)]
    pxs = [capi.scalars_clone(q.px, i % ndev) for i in range(nproofs)]
    got = groth16.prove_batch(pks, ws, pxs, rs)
    for i in range(ndev):
        assert (got[i].PiA, got[i].PiB, got[i].PiC) == want, i
    for i in range(ndev, nproofs):
        assert _same(got[i], groth16.prove_resident(q.device_pk(), q.w, q.px, *rs[i]))
    for h in ws + pxs:
        h.free()
    for k in pks:
        k.handle.free()


def test_config_5_batch_of_64_proofs_with_distinct_witnesses():
    """BASELINE configs[4] as worded -- "batch of 64 independent 2^18-constraint Groth16 proofs, one proof per GPU" -- with 64 DIFFERENT
    witnesses (round 2 shared one witness): the sqchain circuit is the same, so one key serves all (one replica per
    logical device), every proof has its own public input x_i, witness, px (from the sparse system on the device) and randomness.
    Each of the 64 proofs must verify against the device-built vk for ITS x_i and for no other; four of them are compared with the
    single-device blocking prover as well."""
    from gosnark_amd import r1csqap
    n, ndev, nproofs = 1 << 18, 8, 64
    capi.set_device(0)
    inst = synth.sqchain_setup_instance(n, 0xC0F4)
    pks = [groth16.ShardPkTo(inst.device_pk(), 0, 1, d) for d in range(ndev)]
    devs = []
    for d in range(ndev):
        capi.set_device(d)
        devs.append(r1csqap.DeviceR1CS(*inst.r1cs, inst.m))
    capi.set_device(0)
    xs = synth.field_elems(nproofs, 0xC0F5)
    ws, pxs, rs = [], [], []
    for i, x in enumerate(xs):
        w = synth.sqchain_witness(n, x)
        d = i % ndev
        capi.set_device(d)
        ws.append(capi.scalars_upload(w))
        pxs.append(devs[d].ComputePxResident(ws[-1]))
        rs.append(tuple(synth.field_elems(2, 0xC100 + i)))
    capi.set_device(0)
    assert len({capi.scalars_download(w)[2].tobytes() for w in ws[:8]}) == 8          # the witnesses really differ
    got = groth16.prove_batch(pks, ws, pxs, rs)
    assert len(got) == nproofs
    for i, p in enumerate(got):
        assert groth16.VerifyProof(inst.vk, p, [xs[i]]) is True, i
        assert groth16.VerifyProof(inst.vk, p, [xs[(i + 1) % nproofs]]) is False, i
    for i in (0, 9, 31, 63):
        w0, px0 = capi.scalars_clone(ws[i], 0), capi.scalars_clone(pxs[i], 0)
        assert _same(got[i], groth16.prove_resident(inst.device_pk(), w0, px0, *rs[i])), i
    for h in ws + pxs:
        h.free()
    for k in pks:
        k.handle.free()
