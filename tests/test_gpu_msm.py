"""-m gpu parity: HIP Pippenger MSM (through the C ABI) vs the oracles, affine normal form.

Oracle = literal restatement of `acc = Add(acc, MulScalar(base_i, k_i))`
(groth16.go:243-250 / g1.go:140-155 / g2.go:142-181): oracle/ref_py.py for tiny n,
oracle/gs_oracle.c beyond.  Bit-exact on the affine coordinates (integers mod q)."""
import os
import random

import numpy as np
import pytest

import gosnark_amd
from gosnark_amd import capi
import gpu_util as U
from oracle import c_oracle as C
from oracle import ref_py as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True, params=["always", "never"])
def _init(request):
    """every test twice: on window tables and table-free (see tests/test_gpu_prove.py)"""
    capi.init()
    capi.set_table_policy(request.param)
    yield request.param
    capi.set_window_bits(0)
    capi.set_table_policy("auto")


def test_upload_download_affine_normal_form():
    rng = random.Random(21)
    pts = [U.rand_g1_jac(rng, 0.2) for _ in range(37)]
    h = capi.g1_upload(capi.g1_points_to_u64(pts))
    back = capi.u64_to_ints(capi.g1_download(h))
    for i, p in enumerate(pts):
        got = tuple(back[3 * i:3 * i + 3])
        aff = O.G1.Affine(p)
        assert got == ((0, 0, 0) if aff is None else (aff[0], aff[1], 1))
    pts2 = [U.rand_g2_jac(rng, 0.2) for _ in range(9)]
    h2 = capi.g2_upload(capi.g2_points_to_u64(pts2))
    back = capi.u64_to_ints(capi.g2_download(h2))
    for i, p in enumerate(pts2):
        got = back[6 * i:6 * i + 6]
        aff = O.G2.Affine(p)
        want = [0] * 6 if aff is None else [aff[0][0], aff[0][1], aff[1][0], aff[1][1], 1, 0]
        assert got == want


@pytest.mark.parametrize("n", [0, 1, 2, 7, 64])
def test_g1_msm_small_vs_python_oracle(n):
    rng = random.Random(100 + n)
    pts = [U.rand_g1_jac(rng, 0.15) for _ in range(n)]
    ks = [rng.choice([0, 1, 2, O.R - 1]) if rng.random() < 0.3 else rng.randrange(O.R) for _ in range(n)]
    h = capi.g1_upload(capi.g1_points_to_u64(pts))
    got = capi.msm(h, capi.ints_to_u64(ks) if n else np.zeros((0, 4), dtype=np.uint64))
    assert got == U.ref_msm_affine(O.G1, pts, ks)


@pytest.mark.parametrize("n", [1, 5, 33])
def test_g2_msm_small_vs_python_oracle(n):
    rng = random.Random(200 + n)
    pts = [U.rand_g2_jac(rng, 0.15) for _ in range(n)]
    ks = [rng.choice([0, 1, O.R - 1]) if rng.random() < 0.3 else rng.randrange(O.R) for _ in range(n)]
    h = capi.g2_upload(capi.g2_points_to_u64(pts))
    assert capi.msm(h, capi.ints_to_u64(ks), g2=True) == U.ref_msm_affine(O.G2, pts, ks)


def test_g1_msm_degenerate_inputs_complete_addition():
    """Same base several times, P and -P, all-equal scalars: buckets see P+P and P+(-P), which the
    reference's Add cannot handle (g1.go:32-89); the kernels must."""
    p = O.G1.MulScalar(O.G1_GEN, 987654321)
    q = O.G1.Neg(p)
    pts = [p, p, p, q, p, q, q, p]
    ks = [5, 5, 5, 5, 7, 7, 3, O.R - 2]
    h = capi.g1_upload(capi.g1_points_to_u64(pts))
    assert capi.msm(h, capi.ints_to_u64(ks)) == U.ref_msm_affine(O.G1, pts, ks)
    # everything cancels -> infinity
    assert capi.msm(h, capi.ints_to_u64([9, 0, 0, 9, 0, 0, 0, 0])) is None
    # non-canonical scalars (>= r) are reduced mod r
    big = [O.R + 5, 2 * O.R + 1, (1 << 256) - 1, 0, 0, 0, 0, 0]
    assert capi.msm(h, capi.ints_to_u64(big)) == U.ref_msm_affine(O.G1, pts, big)


def test_g2_msm_degenerate_inputs_complete_addition():
    """The G2 accumulation kernel keeps its own accumulator type (ec.h XyzzAcc, y below 2p): the same degenerate chains as the G1
    test -- P + P inside a bucket (the doubling whose y has to be reduced back), P + (-P), everything cancelling, further points on
    top of a doubled one -- plus many copies, so that chunks start, continue and end inside such runs."""
    p = O.G2.MulScalar(O.G2_GEN, 987654321)
    q = O.G2.Neg(p)
    t = O.G2.MulScalar(O.G2_GEN, 4242)
    pts = [p, p, p, q, p, q, q, p, t, t, p, t]
    ks = [5, 5, 5, 5, 7, 7, 3, O.R - 2, 5, 5, 5, 7]
    h = capi.g2_upload(capi.g2_points_to_u64(pts))
    assert capi.msm(h, capi.ints_to_u64(ks), g2=True) == U.ref_msm_affine(O.G2, pts, ks)
    assert capi.msm(h, capi.ints_to_u64([9, 0, 0, 9] + [0] * 8), g2=True) is None
    many = [p, q, t] * 50 + [p] * 40                       # 190 points, a handful of distinct scalars: long runs of equal points per bucket
    mk = [3, 3, 3] * 50 + [3] * 20 + [O.R - 3] * 20
    hm = capi.g2_upload(capi.g2_points_to_u64(many))
    assert capi.msm(hm, capi.ints_to_u64(mk), g2=True) == U.ref_msm_affine(O.G2, many, mk)


@pytest.mark.parametrize("c", [8, 9, 11, 16, 17, 18, 20])
def test_g1_msm_every_window_width(c):
    rng = random.Random(300 + c)
    n = 200
    pts = [U.rand_g1_jac(rng, 0.05) for _ in range(n)]
    ks = U.u64_rows_to_ints(U.rand_scalars_u64(n, 300 + c))
    h = capi.g1_upload(capi.g1_points_to_u64(pts))
    capi.set_window_bits(c)
    try:
        got = capi.msm(h, capi.ints_to_u64(ks))
    finally:
        capi.set_window_bits(0)
    want = C.g1_affine(C.g1_msm_naive(capi.g1_points_to_u64(pts), capi.ints_to_u64(ks)))
    assert got == want


def test_g1_msm_offset_and_ragged_ranges():
    rng = random.Random(41)
    n = 300
    pts = [U.rand_g1_jac(rng) for _ in range(n)]
    arr = capi.g1_points_to_u64(pts)
    h = capi.g1_upload(arr)
    ks = U.rand_scalars_u64(n, 41)
    for off, cnt in ((0, 300), (1, 299), (123, 100), (299, 1), (300, 0)):
        got = capi.msm(h, ks[:cnt], off=off)
        want = C.g1_affine(C.g1_msm_naive(arr[off:off + cnt], ks[:cnt])) if cnt else None
        assert got == want, (off, cnt)
    with pytest.raises(capi.GosnarkHipError):
        capi.msm(h, ks[:10], off=295)


@pytest.mark.parametrize("logn", [10, 12])
def test_g1_msm_vs_c_oracle(logn):
    n = 1 << logn
    ks = U.rand_scalars_u64(n, 500 + logn)
    bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 600 + logn))      # uniform random group elements
    arr = capi.g1_download(bases)
    got = capi.msm(bases, ks)
    assert got == C.g1_affine(C.g1_msm_naive(arr, ks, threads=8))


def test_g2_msm_vs_c_oracle():
    n = 1 << 10
    ks = U.rand_scalars_u64(n, 700)
    bases = capi.g2_fixed_base(U.rand_scalars_u64(n, 701))
    arr = capi.g2_download(bases)
    got = capi.msm(bases, ks, g2=True)
    assert got == C.g2_affine(C.g2_msm_naive(arr, ks, threads=8))


def test_fixed_base_equals_reference_mulscalar():
    """gs_g1_fixed_base / gs_g2_fixed_base vs MulScalar(G, k) (groth16.go:139-175 hot loop)."""
    ks = [0, 1, 2, 77, O.R - 1, 0x1234567890abcdef1234567890abcdef]
    h = capi.g1_fixed_base(capi.ints_to_u64(ks))
    back = capi.u64_to_ints(capi.g1_download(h))
    for i, k in enumerate(ks):
        aff = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, k))
        assert tuple(back[3 * i:3 * i + 3]) == ((0, 0, 0) if aff is None else (aff[0], aff[1], 1))
    # bn128/g1_test.go:29-30
    assert back[9:11] == [0x2f978c0ab89ebaa576866706b14787f360c4d6c3869efe5a72f7c3651a72ff00,
                          0x12e4ba7f0edca8b4fa668fe153aebd908d322dc26ad964d4cd314795844b62b2]
    h2 = capi.g2_fixed_base(capi.ints_to_u64(ks[:4]))
    back = capi.u64_to_ints(capi.g2_download(h2))
    for i, k in enumerate(ks[:4]):
        aff = O.G2.Affine(O.G2.MulScalar(O.G2_GEN, k))
        want = [0] * 6 if aff is None else [aff[0][0], aff[0][1], aff[1][0], aff[1][1], 1, 0]
        assert back[6 * i:6 * i + 6] == want


@pytest.mark.parametrize("logn", [16, 20])
def test_g1_msm_full_size_linearity(logn):
    """Size-independent property at BASELINE sizes: with bases P_i = k_i G,
    sum_i s_i P_i == (sum_i s_i k_i mod r) G  (checked through the fixed-base path)."""
    n = 1 << logn
    k = U.rand_scalars_u64(n, 800 + logn)
    s = U.rand_scalars_u64(n, 900 + logn)
    bases = capi.g1_fixed_base(k)
    got = capi.msm(bases, s)
    ki, si = U.u64_rows_to_ints(k), U.u64_rows_to_ints(s)
    dot = sum(a * b for a, b in zip(ki, si)) % O.R
    want = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, dot))
    assert got == want


def test_sum_affine_combines_partial_sums():
    rng = random.Random(77)
    parts = [O.G1.Affine(U.rand_g1_jac(rng)) for _ in range(7)] + [None]
    acc = O.G1_ZERO
    for p in parts:
        if p is not None:
            acc = O.G1.Add(acc, (p[0], p[1], 1))
    assert capi.sum_affine(parts) == O.G1.Affine(acc)
    parts2 = [O.G2.Affine(U.rand_g2_jac(rng)) for _ in range(3)]
    acc = O.G2_ZERO
    for p in parts2:
        acc = O.G2.Add(acc, (p[0], p[1], (1, 0)))
    assert capi.sum_affine(parts2, g2=True) == O.G2.Affine(acc)


@pytest.mark.parametrize("g2", [False, True])
def test_msm_skewed_witness_heavy_buckets(g2):
    """Real witnesses are dominated by 0/1 and small values (boolean constraints): a few buckets then hold
    thousands of entries and are cut across many 32-entry chunks (partials + block-wide tree combine),
    while most buckets are empty.  8192 terms, 60 % ones, 20 % zeros, 10 % r-1, rest uniform -- vs the C
    oracle's naive loop over the very same points."""
    n = 1 << 13
    rng = random.Random(9090 + g2)
    uni = U.u64_rows_to_ints(U.rand_scalars_u64(n, 77))
    ks = []
    for i in range(n):
        x = rng.random()
        ks.append(1 if x < 0.6 else 0 if x < 0.8 else O.R - 1 if x < 0.9 else uni[i])
    ks = capi.ints_to_u64(ks)
    if g2:
        bases = capi.g2_fixed_base(U.rand_scalars_u64(n, 78))
        want = C.g2_affine(C.g2_msm_naive(capi.g2_download(bases), ks, threads=8))
    else:
        bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 78))
        want = C.g1_affine(C.g1_msm_naive(capi.g1_download(bases), ks, threads=8))
    for c in (0, 8, 13, 17, 19, 20):
        capi.set_window_bits(c)
        try:
            assert capi.msm(bases, ks, g2=g2) == want, c
        finally:
            capi.set_window_bits(0)


@pytest.mark.parametrize("g2", [False, True])
def test_several_heavy_buckets_of_every_size_through_the_sliced_tree(g2):
    """Round 4's heavy path (k_heavy_combine over (bucket, slice) items + k_heavy_finish): heavy buckets of very different sizes in one
    MSM -- one that holds 40 % of all terms, some of a few hundred chunks, and some just over the 64-chunk threshold, where most of a
    bucket's 16 slices hold 4-5 chunks and the last ones are partly or wholly empty -- beside ordinary buckets, against the C oracle's
    naive loop over the same points; in both forms of the tail kernels (the MSM is small: one wave per SIMD) and pipelined."""
    n = 1 << (14 if g2 else 15)
    rng = random.Random(4242 + g2)
    uni = U.u64_rows_to_ints(U.rand_scalars_u64(n, 79))
    small = [(1, 0.40), (2, 0.15), (3, 0.10), (5, 0.05), (7, 0.0335), (O.R - 2, 0.034)]      # value, share of the terms
    ks = []
    for i in range(n):
        x, acc = rng.random(), 0.0
        v = uni[i]
        for val, share in small:
            acc += share
            if x < acc:
                v = val
                break
        ks.append(v)
    ks = capi.ints_to_u64(ks)
    if g2:
        bases = capi.g2_fixed_base(U.rand_scalars_u64(n, 80))
        want = C.g2_affine(C.g2_msm_naive(capi.g2_download(bases), ks, threads=8))
    else:
        bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 80))
        want = C.g1_affine(C.g1_msm_naive(capi.g1_download(bases), ks, threads=8))
    sc = capi.scalars_upload(ks)
    for c in (0, 10, 16):
        capi.set_window_bits(c)
        try:
            assert capi.msm(bases, ks, g2=g2) == want, c
            assert capi.last_timing()["heavy_buckets"] >= 2, c       # (the chunk size follows the window width: at least the two largest stay heavy)
            tickets = [capi.msm_begin(bases, sc, n, g2=g2) for _ in range(3)]
            assert all(capi.msm_end(t) == want for t in tickets), c
        finally:
            capi.set_window_bits(0)


def test_reference_g1_g2_tests_through_the_c_abi():
    """bn128/g1_test.go:14-31 and g2_test.go:12-25 through the mirror gosnark_amd.bn128 (MulScalar / Add as MSM calls):
    g*33 + g*44 == g*77, and the affine KAT of 77*G1 the reference pins."""
    from gosnark_amd import bn128
    gr1 = bn128.G1.MulScalar(O.G1_GEN, 33)
    gr2 = bn128.G1.MulScalar(O.G1_GEN, 44)
    grsum1 = bn128.G1.Add(gr1, gr2)
    grsum2 = bn128.G1.MulScalar(O.G1_GEN, 33 + 44)
    assert grsum1 == grsum2
    a = bn128.G1.Affine(grsum1)
    assert "%064x" % a[0] == "2f978c0ab89ebaa576866706b14787f360c4d6c3869efe5a72f7c3651a72ff00"      # g1_test.go:29
    assert "%064x" % a[1] == "12e4ba7f0edca8b4fa668fe153aebd908d322dc26ad964d4cd314795844b62b2"      # g1_test.go:30
    h1 = bn128.G2.MulScalar(O.G2_GEN, 33)
    h2 = bn128.G2.MulScalar(O.G2_GEN, 44)
    assert bn128.G2.Add(h1, h2) == bn128.G2.MulScalar(O.G2_GEN, 77)
    assert bn128.G2.Affine(bn128.G2.Add(h1, h2)) == O.G2.Affine(O.G2.MulScalar(O.G2_GEN, 77))
    # identities the reference's Add special-cases (g1.go:33-38): P + 0 = P, 0 + 0 = 0
    assert bn128.G1.Add(gr1, bn128.G1_ZERO) == gr1 and bn128.G1.Add(bn128.G1_ZERO, bn128.G1_ZERO) == bn128.G1_ZERO
    assert bn128.G1.MulScalar(gr1, 0) == bn128.G1_ZERO


def test_pipelined_msm_tickets_equal_blocking_calls():
    """gs_msm_g1_begin / gs_msm_g2_begin / gs_msm_end with two operations outstanding (mixed G1 / G2, different ranges)."""
    n = 1 << 12
    b1 = capi.g1_fixed_base(U.rand_scalars_u64(n, 901))
    b2 = capi.g2_fixed_base(U.rand_scalars_u64(n, 902))
    s = capi.scalars_upload(U.rand_scalars_u64(n, 903))
    jobs = [(b1, False, 0, 0, n), (b2, True, 0, 0, n), (b1, False, 100, 7, 1000), (b2, True, 5, 5, 333), (b1, False, 0, 0, 1)]
    want = [capi.msm_resident(b, s, cnt, off=off, soff=soff, g2=g2) for b, g2, off, soff, cnt in jobs]
    got, tickets = [], []
    for b, g2, off, soff, cnt in jobs:
        tickets.append(capi.msm_begin(b, s, cnt, off=off, soff=soff, g2=g2))
        if len(tickets) == 3:
            got.append(capi.msm_end(tickets.pop(0)))
    while tickets:
        got.append(capi.msm_end(tickets.pop(0)))
    assert got == want
    with pytest.raises(capi.GosnarkHipError):
        capi.msm_end((987654, False))


def test_concurrent_callers_from_several_threads_get_their_own_results():
    """SURVEY 8b threading: the shim must be callable from several goroutines.  ctypes drops the GIL during a call, so
    these threads really overlap on the library (its mutex serialises device work; gs_last_error is thread-local; the
    verifier entry points take no lock at all).  Every thread must get exactly what a lone caller gets."""
    import json
    import os
    import threading
    from gosnark_amd import groth16, utils
    n = 5000
    bases = capi.g1_fixed_base(U.rand_scalars_u64(n, 900))
    bases2 = capi.g2_fixed_base(U.rand_scalars_u64(257, 901))
    scal = [U.rand_scalars_u64(n, 910 + t) for t in range(6)]
    want1 = [capi.msm(bases, s) for s in scal]
    want2 = [capi.msm(bases2, s[:257], g2=True) for s in scal]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wasm_groth_x3.json")) as f:
        rec = json.load(f)
    _, vk = utils.GrothSetupFromString(json.loads(rec["setup"]))
    proof = utils.GrothProofFromString(json.loads(rec["proof"]))
    errors, lib = [], capi.load_library()

    def worker(t):
        try:
            for it in range(4):
                k = (t + it) % 6
                assert capi.msm(bases, scal[k]) == want1[k]
                assert capi.msm(bases2, scal[k][:257], g2=True) == want2[k]
                assert groth16.VerifyProof(vk, proof, [35]) and not groth16.VerifyProof(vk, proof, [34])
                # a failing call on this thread leaves ITS message, not another thread's
                h = capi.Handle(0)
                assert lib.gs_msm_g1(capi.Handle(987654 + t), capi.ptr64(scal[k]), 0, 1, capi.ptr64(np.zeros(8, dtype=np.uint64)),
                                     capi.ctypes.byref(capi.ctypes.c_int(0))) < 0
                assert b"handle" in lib.gs_last_error()
                del h
        except Exception as e:   # noqa: BLE001
            errors.append((t, repr(e)))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_upload_rejects_points_that_are_not_on_their_curve():
    rng = random.Random(61)
    pts = [U.rand_g1_jac(rng, 0.1) for _ in range(50)]
    capi.g1_upload(capi.g1_points_to_u64(pts))                             # fine
    bad = list(pts)
    bad[17] = (5, 7, 1)
    bad[33] = (bad[33][0], (bad[33][1] + 1) % O.Q, bad[33][2])
    with pytest.raises(capi.GosnarkHipError, match=r"2 of the 50 points are not on the curve \(first at index 17\)"):
        capi.g1_upload(capi.g1_points_to_u64(bad))
    pts2 = [U.rand_g2_jac(rng, 0.1) for _ in range(9)]
    capi.g2_upload(capi.g2_points_to_u64(pts2))
    bad2 = list(pts2)
    k = next(i for i, p in enumerate(bad2) if p[2] != (0, 0))
    bad2[k] = (bad2[k][0], ((bad2[k][1][0] + 1) % O.Q, bad2[k][1][1]), bad2[k][2])
    with pytest.raises(capi.GosnarkHipError, match="not on the curve"):
        capi.g2_upload(capi.g2_points_to_u64(bad2))
    # the Jacobian representative does not matter, infinity passes
    capi.g1_upload(capi.g1_points_to_u64([(0, 0, 0), O.G1.MulScalar(O.G1_GEN, 12345), (7, 11, 0)]))


def test_upload_of_normalised_points_takes_the_same_values_as_any_other_representative():
    """Round 6: points that arrive with Z = 1 skip the inversion of k_jacobian_to_affine (a key that was normalised when it was written
    uploads in 16 ms instead of 39 at 2^20).  The same 300 G1 / 70 G2 points as [x, y, 1], as [x l^2, y l^3, l] with random l, and mixed
    lane by lane (a wave then takes both branches), infinities among them: the resident arrays are equal word for word."""
    rng = random.Random(77)
    for g2, count in ((False, 300), (True, 70)):
        mul, gen = (O.G2.MulScalar, O.G2_GEN) if g2 else (O.G1.MulScalar, O.G1_GEN)
        aff = (C.g2_affine if g2 else C.g1_affine)
        to_u64 = capi.g2_points_to_u64 if g2 else capi.g1_points_to_u64
        up, down = (capi.g2_upload, capi.g2_download) if g2 else (capi.g1_upload, capi.g1_download)
        base = mul(gen, rng.randrange(1, O.R))
        norm, other = [], []
        for i in range(count):
            if i % 41 == 7:
                norm.append(((0, 0), (0, 0), (0, 0)) if g2 else (0, 0, 0))
                other.append(((3, 4), (5, 6), (0, 0)) if g2 else (3, 4, 0))          # infinity under another name
                continue
            x, y = aff(mul(base, rng.randrange(1, 1 << 40)))
            lam = rng.randrange(2, O.Q)
            if g2:
                f2 = O.FQ2
                l = (lam, rng.randrange(O.Q))
                l2 = f2.Square(l)
                norm.append((x, y, (1, 0)))
                other.append((f2.Mul(x, l2), f2.Mul(y, f2.Mul(l2, l)), l))
            else:
                norm.append((x, y, 1))
                other.append((x * lam * lam % O.Q, y * pow(lam, 3, O.Q) % O.Q, lam))
        mixed = [norm[i] if (i * 7 + i // 3) % 2 else other[i] for i in range(count)]
        got = [np.asarray(down(up(to_u64(v)))) for v in (norm, other, mixed)]
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])


def test_plain_c_process_drives_the_library_without_python_or_torch(tmp_path):
    """What a cgo caller sees: a C program (no Python, no torch in the process) initialises the device, builds bases with
    gs_g1_fixed_base, runs gs_msm_g1 and checks sum_i s_i * (k_i G) == (sum_i s_i k_i) G through a second fixed-base call."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "drive.c"
    src.write_text(r'''#include "gosnark_hip.h"
#include <stdio.h>
#include <string.h>
#define N 1000
#define CHECK(x) do { int _s = (x); if (_s != 0) { printf("FAIL %s: %d %s\n", #x, _s, gs_last_error()); return 1; } } while (0)
int main(void) {
  int dev = 0, inf = 0;
  static uint64_t k[N * 4], s[N * 4], pts[N * 12], one[12], sum[8];
  uint64_t dot[4] = {0, 0, 0, 0};
  gs_handle bases = 0, single = 0;
  CHECK(gs_init(&dev, 1));
  memset(k, 0, sizeof k); memset(s, 0, sizeof s);
  for (int i = 0; i < N; ++i) { k[4 * i] = 3 + 7 * (uint64_t)i; s[4 * i] = 1000003 + 13 * (uint64_t)i; dot[0] += k[4 * i] * s[4 * i]; }   /* < 2^64 */
  CHECK(gs_g1_fixed_base(k, N, &bases));
  CHECK(gs_msm_g1(bases, s, 0, N, sum, &inf));
  CHECK(gs_g1_fixed_base(dot, 1, &single));
  CHECK(gs_g1_download(single, one, 1));
  if (inf || memcmp(sum, one, 64) != 0) { printf("FAIL: MSM result differs from the fixed-base multiple\n"); return 2; }
  CHECK(gs_g1_download(bases, pts, N));
  CHECK(gs_free(bases)); CHECK(gs_free(single));
  gs_shutdown();
  printf("OK %s\n", gs_version());
  return 0;
}
''')
    exe = tmp_path / "drive"
    libdir = os.path.join(root, "go-snark-study_amd")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lgosnark_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr


def test_2p20_g1_msm_equals_the_naive_loop_golden():
    """The headline size against the literal reference loop: sum_i MulScalar(MulScalar(G, k_i), s_i) over 2^20 seeded terms, computed
    offline by the C restatement of bn128/g1.go on all host cores (oracle/gen_golden_large.py msm20 ->
    tests/golden/oracle_msm_g1_2p20.json).  Host scalars, resident scalars, three MSMs in flight, and the widest window."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_msm_g1_2p20.json")) as f:
        rec = json.load(f)
    n = rec["n"]
    bases = capi.g1_fixed_base(synth.scalars_u64(n, rec["seed_bases"]))
    sc = synth.scalars_u64(n, rec["seed_scalars"])
    want = (int(rec["x"]), int(rec["y"]))
    assert capi.msm(bases, sc) == want
    h = capi.scalars_upload(sc)
    tickets = [capi.msm_begin(bases, h, n) for _ in range(3)]
    assert [capi.msm_end(t) for t in tickets] == [want] * 3
    capi.set_window_bits(20)
    try:
        assert capi.msm_resident(bases, h, n) == want
    finally:
        capi.set_window_bits(0)
    h.free()
    bases.free()


def test_config_2_as_worded_2p16_g1_msm_equals_the_naive_loop_golden():
    """BASELINE configs[1]: 'Synthetic 2^16 ... G1 Pippenger MSM only, bit-exact vs bn128.G1 loop'.  The expected point was computed
    offline by oracle/gen_golden_large.py: bases P_i = MulScalar(G, k_i), then acc = Add(acc, MulScalar(P_i, s_i)) over all 2^16
    terms with the C restatement of bn128/g1.go on all host cores (tests/golden/oracle_msm_g1_2p16.json)."""
    import json
    import os
    from gosnark_amd import synth
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_msm_g1_2p16.json")) as f:
        rec = json.load(f)
    n = rec["n"]
    bases = capi.g1_fixed_base(synth.scalars_u64(n, rec["seed_bases"]))
    sc = synth.scalars_u64(n, rec["seed_scalars"])
    want = (int(rec["x"]), int(rec["y"]))
    assert capi.msm(bases, sc) == want
    for c in (13, 16, 18):                                  # and at other window widths
        capi.set_window_bits(c)
        try:
            assert capi.msm(bases, sc) == want, c
        finally:
            capi.set_window_bits(0)
    # the bases themselves: the device's fixed-base batch against the oracle's MulScalar(G, k) on a sample
    pts = capi.g1_download(bases)
    ks = U.u64_rows_to_ints(synth.scalars_u64(n, rec["seed_bases"]))
    for i in (0, 1, n // 2, n - 1):
        a = O.G1.Affine(O.G1.MulScalar(O.G1_GEN, ks[i]))
        assert tuple(U.u64_rows_to_ints(pts[i])) == (a[0], a[1], 1)


@pytest.mark.parametrize("n", [32767, 131072, 131073, 3 * 131072 + 5])
def test_uploads_from_pageable_memory_across_the_staging_pieces(n):
    """Host buffers reach the device through 4 MiB pinned staging pieces filled by several host threads (csrc/hostcopy.h): sizes just
    below the 1 MiB threshold, exactly one piece, one piece + one element, several pieces + a ragged tail; from a deliberately
    misaligned (odd-offset) source.  The resident copy reads back identical, and the host-scalar MSM equals the resident one."""
    from gosnark_amd import synth
    raw = np.zeros(4 * n + 1, dtype=np.uint64)
    sc = raw[1:].reshape(n, 4)                       # 8 bytes off any 16-byte alignment
    sc[:] = synth.scalars_u64(n, 0xC0FFEE + n)
    h = capi.scalars_upload(sc)
    assert np.array_equal(capi.scalars_download(h), sc)
    bases = capi.g1_fixed_base(synth.scalars_u64(n, 17 + n))
    assert capi.msm(bases, sc) == capi.msm_resident(bases, h, n)
    h.free()
    bases.free()


def test_hostile_environment_cannot_change_a_result():
    """every environment variable the library knows, set to a value that round 3's raw atoi() would have used
    (GS_REDUCE_L=3 gave a wrong MSM with status 0), in a FRESH process (the knobs are read once): a 2^12-term G1 MSM, a 2^10-term G2 MSM
    and a pipelined x^3 + x + 5 proof must equal what this process computes, which the tests above pin to the oracle."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import gosnark_amd; from gosnark_amd import capi, synth, groth16; import gpu_util as U\n"
        "capi.init(0)\n"
        "b1 = capi.g1_fixed_base(U.rand_scalars_u64(4096, 611)); k1 = U.rand_scalars_u64(4096, 511)\n"
        "b2 = capi.g2_fixed_base(U.rand_scalars_u64(1024, 701)); k2 = U.rand_scalars_u64(1024, 700)\n"
        "inst = synth.sqchain_setup_instance(256, 99); r, s = synth.field_elems(2, 5)\n"
        "p = groth16.prove_end(groth16.prove_begin(inst.device_pk(), inst.w, inst.px, r, s))\n"
        "q = groth16.prove_end(groth16.prove_host_begin(inst.device_pk(), inst.w_host, capi.scalars_download(inst.px), r, s))\n"
        "print(json.dumps([capi.msm(b1, k1), capi.msm(b2, k2, g2=True), [p.PiA, p.PiB, p.PiC], [q.PiA, q.PiB, q.PiC]]))\n" % (root, os.path.join(root, "tests")))
    hostile = {"GS_REDUCE_L": "3", "GS_CHUNK": "7", "GS_FOLD_MAX": "-5", "GS_AUTO_MAX_C": "99", "GS_SORT_BLOCK": "1", "GS_PART_MIN_R": "0",
               "GS_WINDOW_COST_BUCKET": "nan", "GS_TABLE_PER_ROW": "x", "GS_TAIL_FLIP": "77", "GS_TAIL_PRIORITY": "9", "GS_CHUNK_H": "5", "GS_TAIL_ALONE_LOG2": "99", "GS_COPY_THREADS": "-3", "GS_PLANW_STREAM": "7", "GS_MSM_TICKET_STREAMS": "5",
               "GS_NO_PRIORITY": "1",
               # round 5's switches: staging of host buffers, table policy and background builds, the sparse-B threshold
               "GS_HOST_STAGE": "9", "GS_STAGE_MIB": "0", "GS_STAGE_BUFFERS": "99", "GS_TABLE_POLICY": "7", "GS_TABLE_BG_SLAB_LOG2": "40",
               "GS_TABLE_STREAM_LOW": "-1", "GS_SPLIT_B_PERCENT": "1000", "GS_CHUNK_MODEL": "5", "GS_TABLE_BUDGET_PCT": "-7", "GS_SLOT_STREAMS": "9"}

    def run(env):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    assert run(hostile) == run({})


def test_randomised_degenerate_sums_equal_their_closed_form():
    """tools/stress_msm_random.py for a few seconds (hundreds of random cases): duplicate bases, negated pairs, points at infinity, zeros /
    ones / repeated values among the scalars, G1 and G2, two window widths each, blocking and pipelined -- partial sums that coincide
    or cancel in the tail kernels' memory-operand addition.  Expected values are (sum_i +-k_i b_i) * G, complete by construction (the
    naive reference loop is not: its Add has no P == Q branch)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (GS_STRESS_DUMP: should the tool ever stall, it dumps its Python stack and exits instead of sitting in the pipe until the timeout -- one run of
    #  the suite in round 6 lost this test to a 600 s timeout on a box that was also 5 % slow on every bench line; it could not be reproduced in
    #  three further suite runs nor standalone with this and two other seeds.  One retry, so that a stalled BOX does not fail the suite; a stalled
    #  LIBRARY fails twice and shows where.)
    env = dict(os.environ, GS_STRESS_DUMP="150")
    out = None
    for attempt in range(2):
        out = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(root, "tools", "stress_msm_random.py"), "8", "7"], capture_output=True, text=True,
                             timeout=400, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0 and "all equal the closed form" in out.stdout, (out.stdout[-500:], out.stderr[-2500:])
