"""-m gpu: every Go wrapper's exact C call sequence, run from plain C processes (no Python, no torch) against the
reference's own goldens.  The Go layer (go/) cannot be compiled in this image; these programs are its executable
mirror -- INTEGRATION.md lists the wrapper <-> program pairs."""
import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import capi
import c_util
import golden_util as GU
from oracle import ref_py as O

pytestmark = pytest.mark.gpu


def aff1(p):
    a = O.G1.Affine(GU.g1(p) if not isinstance(p[0], int) else p)
    return [0, 0] if a is None else [a[0], a[1]]


def aff2(p):
    a = O.G2.Affine(GU.g2(p) if not isinstance(p[0][0], int) else p)
    return [0, 0, 0, 0] if a is None else [a[0][0], a[0][1], a[1][0], a[1][1]]


def jac1(p):
    a = aff1(p)
    return a + [1] if any(a) else [0, 0, 0]


def jac2(p):
    a = aff2(p)
    return a + [1, 0] if any(a) else [0] * 6


def words(arr):
    return capi.u64_to_ints(np.ascontiguousarray(arr))


def groth_toxic():
    return tuple(int.from_bytes(bytes((i * k + 7) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13))


def pinocchio_toxic():
    return tuple(int.from_bytes(bytes((i * k + 9) & 0xff for i in range(30)), "big") % O.R for k in (3, 5, 7, 11, 13, 17, 19, 23))


def test_groth16_generateproofs_c_sequence_reproduces_the_reference_proof(tmp_path):
    """go/groth16hip.GenerateProofsWithRS + VerifyProof == tests/c/groth16_generateproofs.c: the x^3 + x + 5 key, witness, px and
    (r, s) as flat limb buffers -> byte for byte the affine form of the proof the reference's compiled prover produced."""
    rec = GU.load("groth_x3")
    blob = c_util.write_groth_instance(tmp_path, rec)
    out = tmp_path / "proof.bin"
    assert c_util.build_and_run("groth16_generateproofs.c", [str(blob), str(out)], tmp_path).startswith("OK")
    raw = c_util.read_words(out)
    assert list(raw[32:36]) == [0, 0, 0, 1]                        # no infinities; right input accepted and wrong one rejected
    assert words(raw[:32]) == aff1(rec["proof"]["PiA"]) + aff2(rec["proof"]["PiB"]) + aff1(rec["proof"]["PiC"])


def test_groth16_setup_prove_verify_c_sequence(tmp_path):
    """go/groth16hip.GenerateTrustedSetup -> GenerateProofs -> VerifyProof == tests/c/groth16_setup_prove_verify.c: the key the
    device builds from the sparse R1CS and the golden's toxic recipe, read back with gs_groth16_pk_export, is the key the
    reference's verifier accepted; the proof made with it is the reference prover's proof."""
    rec = GU.load("groth_x3")
    blob = c_util.write_groth_instance(tmp_path, rec)
    r1cs = c_util.write_r1cs(tmp_path, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C), 1, groth_toxic())
    out = tmp_path / "setup.bin"
    assert c_util.build_and_run("groth16_setup_prove_verify.c", [str(r1cs), str(blob), str(out)], tmp_path).strip() == "OK"
    raw = c_util.read_words(out)
    m, nz, npub = 8, 7, 1
    assert list(raw[32:36]) == [0, 0, 0, 1]
    assert words(raw[:32]) == aff1(rec["proof"]["PiA"]) + aff2(rec["proof"]["PiB"]) + aff1(rec["proof"]["PiC"])
    pos = 36
    svk, opk = rec["setup"]["Vk"], GU.groth_pk(rec["setup"])
    nvk = 84 + 12 * (npub + 1)
    want_vk = jac1(svk["G1"]["Alpha"]) + jac2(svk["G2"]["Beta"]) + jac2(svk["G2"]["Gamma"]) + jac2(svk["G2"]["Delta"])
    for p in svk["IC"]:
        want_vk += jac1(p)
    assert words(raw[pos:pos + nvk]) == want_vk
    pos += nvk
    for arr, jac, w in ((opk.G1_At, jac1, 12), (opk.G1_BACGamma, jac1, 12), (opk.G2_BACGamma, jac2, 24), (opk.BACDelta, jac1, 12),
                        (opk.PowersTauDelta, jac1, 12)):
        want = [x for p in arr for x in jac(p)]
        assert words(raw[pos:pos + w * len(arr)]) == want
        pos += w * len(arr)
    assert len(opk.PowersTauDelta) == nz
    want_single = jac1(opk.G1_Alpha) + jac1(opk.G1_Beta) + jac1(opk.G1_Delta) + jac2(opk.G2_Beta) + jac2(opk.G2_Delta)
    assert words(raw[pos:pos + 84]) == want_single
    pos += 84
    assert words(raw[pos:pos + 4 * nz]) == [z % O.R for z in opk.Z] and m == len(opk.G1_At)


@pytest.mark.parametrize("golden", ["pinocchio_x3_fixture", "pinocchio_rand_m9"])
def test_snark_generateproofs_c_sequence_reproduces_the_reference_proof(tmp_path, golden):
    """go/snarkhip.GenerateProofs + VerifyProof == tests/c/snark_generateproofs.c, on the reference's own wasm/index.js fixture
    (Pinocchio index.js fixture -> byte-identical affine proof) and on a random instance."""
    rec = GU.load(golden)
    public = [x for x in rec["w"][1:1 + rec["circuit"]["NPublic"]]]
    blob = c_util.write_pinocchio_instance(tmp_path, rec, public)
    out = tmp_path / "proof.bin"
    assert c_util.build_and_run("snark_generateproofs.c", [str(blob), str(out)], tmp_path).strip() == "OK"
    raw = c_util.read_words(out)
    pr = rec["proof"]
    want = aff1(pr["PiA"]) + aff1(pr["PiAp"]) + aff2(pr["PiB"]) + aff1(pr["PiBp"]) + aff1(pr["PiC"]) + aff1(pr["PiCp"]) + aff1(pr["PiH"]) + aff1(pr["PiKp"])
    assert words(raw[:72]) == want
    assert list(raw[72:80]) == [1 if not any(aff1(pr[k])) else 0 for k in ("PiA", "PiAp")] + [0] + \
        [1 if not any(aff1(pr[k])) else 0 for k in ("PiBp", "PiC", "PiCp", "PiH", "PiKp")]
    if golden == "pinocchio_x3_fixture":
        assert [v["result"] for v in rec["verify"]] == ["true", "false"]
        assert list(raw[80:82]) == [1, 0] and raw[82] == 0 and raw[83] != 0      # accepted; wrong public input fails a named check


def test_snark_setup_prove_verify_c_sequence(tmp_path):
    """go/snarkhip.GenerateTrustedSetup -> GenerateProofs -> VerifyProof == tests/c/snark_setup_prove_verify.c."""
    rec = GU.load("pinocchio_x3_setup")
    blob = c_util.write_pinocchio_instance(tmp_path, rec)
    r1cs = c_util.write_r1cs(tmp_path, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C), 1, pinocchio_toxic())
    out = tmp_path / "setup.bin"
    assert c_util.build_and_run("snark_setup_prove_verify.c", [str(r1cs), str(blob), str(out)], tmp_path).strip().endswith("OK")    # (RCCL prints its version banner when the rank-mode communicator is created)
    raw = c_util.read_words(out)
    pr, spk, svk = rec["proof"], rec["setup"]["Pk"], rec["setup"]["Vk"]
    want = aff1(pr["PiA"]) + aff1(pr["PiAp"]) + aff2(pr["PiB"]) + aff1(pr["PiBp"]) + aff1(pr["PiC"]) + aff1(pr["PiCp"]) + aff1(pr["PiH"]) + aff1(pr["PiKp"])
    assert words(raw[:72]) == want and list(raw[80:82]) == [1, 0]
    pos = 82
    want_vk = jac2(svk["Vka"]) + jac1(svk["Vkb"]) + jac2(svk["Vkc"]) + jac1(svk["G1Kbg"]) + jac2(svk["G2Kbg"]) + jac2(svk["G2Kg"]) + jac2(svk["Vkz"])
    for p in svk["IC"]:
        want_vk += jac1(p)
    assert words(raw[pos:pos + 4 * len(want_vk)]) == want_vk
    pos += 4 * len(want_vk)
    for name, w in (("A", 12), ("Ap", 12), ("B", 24), ("Bp", 12), ("C", 12), ("Cp", 12), ("Kp", 12), ("G1T", 12)):
        arr = spk[name]
        want = [x for p in arr for x in (jac2(p) if w == 24 else jac1(p))]
        if name in ("A", "Ap"):                                     # infinity for i <= NPublic: what snark.go:265 skips
            want = [0] * 6 + want[6:]                               # two points of three coordinates
        assert words(raw[pos:pos + w * len(arr)]) == want, name
        pos += w * len(arr)
    assert words(raw[pos:pos + 4 * len(spk["Z"])]) == [int(z) % O.R for z in spk["Z"]]


def test_prove_batch_c_sequence_two_logical_devices(tmp_path):
    """go/gosnarkhip.ProveBatch == tests/c/batch_devices.c."""
    blob = c_util.write_groth_instance(tmp_path, GU.load("groth_x3"))
    assert c_util.build_and_run("batch_devices.c", [str(blob)], tmp_path).strip().endswith("OK")


def test_memory_eviction_c_sequence(tmp_path):
    """go/gosnarkhip.MemoryOf / HandleBytes / ReleaseTables / Trim + SetTablePolicy / SetMemoryLimit == tests/c/memory_eviction.c
    (round 5: two keys under a memory cap evict each other's idle window tables instead of failing; a ticket's key is never the victim)."""
    blob = c_util.write_groth_instance(tmp_path, GU.load("groth_x3"))
    assert c_util.build_and_run("memory_eviction.c", [str(blob)], tmp_path).strip().endswith("OK")


def test_stream_host_c_sequence_reproduces_both_reference_proofs(tmp_path):
    """go/gosnarkhip/stream.go == tests/c/stream_host.c: host-buffer tickets (w + px, w alone, both protocols), gs_scalars_update under
    an outstanding ticket, no allocation in a steady lap, a fresh key's first proof table-free under the default policy, gs_build_tables,
    policy never; the proofs written out are the reference prover's (wasm goldens)."""
    g, p = GU.load("groth_x3"), GU.load("pinocchio_x3_setup")
    gblob = c_util.write_groth_instance(tmp_path, g)
    pblob = c_util.write_pinocchio_instance(tmp_path, p)
    r1cs = c_util.write_r1cs(tmp_path, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C), 1, [])
    out = tmp_path / "proofs.bin"
    assert c_util.build_and_run("stream_host.c", [str(r1cs), str(gblob), str(pblob), str(out)], tmp_path).strip() == "OK"
    raw = c_util.read_words(out)
    assert words(raw[:32]) == aff1(g["proof"]["PiA"]) + aff2(g["proof"]["PiB"]) + aff1(g["proof"]["PiC"])
    pr = p["proof"]
    assert words(raw[32:104]) == aff1(pr["PiA"]) + aff1(pr["PiAp"]) + aff2(pr["PiB"]) + aff1(pr["PiBp"]) + aff1(pr["PiC"]) + \
        aff1(pr["PiCp"]) + aff1(pr["PiH"]) + aff1(pr["PiKp"])


def test_witness_to_proof_c_sequence_reproduces_both_reference_proofs(tmp_path):
    """go/groth16hip.GenerateProofsFromWitness + go/snarkhip.GenerateProofsFromWitness == tests/c/witness_to_proof.c: the x^3+x+5
    circuit's sparse R1CS resident, then witness -> proof on the device for both protocols; px equals the reference's
    CombinePolynomials output and the proofs are the reference prover's (wasm goldens)."""
    g, p = GU.load("groth_x3"), GU.load("pinocchio_x3_setup")
    gblob = c_util.write_groth_instance(tmp_path, g)
    pblob = c_util.write_pinocchio_instance(tmp_path, p)
    r1cs = c_util.write_r1cs(tmp_path, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C), 1, [])
    out = tmp_path / "proofs.bin"
    assert c_util.build_and_run("witness_to_proof.c", [str(r1cs), str(gblob), str(pblob), str(out)], tmp_path).strip() == "OK"
    raw = c_util.read_words(out)
    assert words(raw[:32]) == aff1(g["proof"]["PiA"]) + aff2(g["proof"]["PiB"]) + aff1(g["proof"]["PiC"])
    pr = p["proof"]
    assert words(raw[32:104]) == aff1(pr["PiA"]) + aff1(pr["PiAp"]) + aff2(pr["PiB"]) + aff1(pr["PiBp"]) + aff1(pr["PiC"]) + \
        aff1(pr["PiCp"]) + aff1(pr["PiH"]) + aff1(pr["PiKp"])


def test_polynomial_field_c_sequence_equals_the_oracle(tmp_path):
    """go/r1csqaphip.PolynomialField (Mul, Add, Sub, Div, Eval, LagrangeInterpolation, R1CSToQAP's Z, CombinePolynomials) ==
    tests/c/polynomial_field.c, on the reference's own test vectors (r1csqap_test.go:59-112) and the x^3 + x + 5 system; every
    result equals the oracle's restatement of r1csqap.go."""
    rec = GU.load("groth_x3")
    blob = c_util.write_groth_instance(tmp_path, rec)
    r1cs = c_util.write_r1cs(tmp_path, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C), 1, [])
    out = tmp_path / "poly.bin"
    assert c_util.build_and_run("polynomial_field.c", [str(r1cs), str(blob), str(out)], tmp_path).strip() == "OK"
    v = words(c_util.read_words(out))
    a, b = [1, 0, 5], [3, 0, 1]
    pos = 0

    def take(k):
        nonlocal pos
        r = v[pos:pos + k]
        pos += k
        return r
    prod = O.PF.Mul(a, b)
    assert take(5) == prod
    assert take(3) == O.PF.Add(a, b)
    assert take(3) == O.PF.Sub(a, b)
    quo, rem = O.PF.Div(prod, b)
    assert take(3) == quo and take(2) == (rem + [0, 0])[:2]
    assert take(1) == [O.PF.Eval(a, 7)]
    assert take(4) == O.PF.LagrangeInterpolation([0, 0, 0, 5])
    al, be, ga, z = O.PF.R1CSToQAP(O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)
    assert take(len(z)) == z
    ax, bx, cx, px = O.PF.CombinePolynomials(list(O.X3_WITNESS), al, be, ga)
    n = len(O.X3_R1CS_A)
    assert take(n) == ax and take(n) == bx and take(n) == cx and take(2 * n - 1) == px
    assert pos == len(v)


def test_group_ops_c_sequence_equals_the_oracle(tmp_path):
    """go/bn128hip.G1 / G2 (MulScalar, Add, Double), the MSM tickets (begin / cancel / end), the resident MSM and bn128.Pairing ==
    tests/c/group_ops.c; every point equals the affine form of the oracle's restatement of bn128/g1.go / g2.go, the pairing value
    the restatement of bn128.Pairing."""
    from oracle import ref_pairing as RP
    rec = GU.load("groth_x3")
    opk = GU.groth_pk(rec["setup"])
    blob = c_util.write_groth_instance(tmp_path, rec)
    out = tmp_path / "group.bin"
    assert c_util.build_and_run("group_ops.c", [str(blob), str(out)], tmp_path).strip() == "OK"
    v = words(c_util.read_words(out))
    k = 0x123456789abcdef1 | (0x0fedcba987654321 << 64) | (0x1111222233334444 << 128) | (0x0123456789abcdef << 192)
    P, Q = opk.G1_At[2], opk.G1_At[3]
    P2, Q2 = opk.G2_BACGamma[2], opk.G2_BACGamma[3]
    pos = 0
    for want in (O.G1.MulScalar(P, k % O.R), O.G1.Add(P, Q), O.G1.Double(P), O.G1.Add(P, Q), O.G1.Add(P, Q)):
        assert v[pos:pos + 3] == aff1(want) + [0], pos
        pos += 3
    for want in (O.G2.MulScalar(P2, k % O.R), O.G2.Add(P2, Q2)):
        assert v[pos:pos + 5] == aff2(want) + [0], pos
        pos += 5
    e = RP.Pairing(P, P2)
    flat = [c for half in e for pair in half for c in pair]
    assert v[pos:pos + 12] == flat
    assert pos + 12 == len(v)


def test_stream_producer_c_sequence(tmp_path):
    """go/groth16hip.Prover (gosnarkhip.Groth16Prover: Submit / Collect over host-buffer tickets) == tests/c/stream_producer.c: a plain-C
    process builds sqchain(2^12), runs the device setup, computes 6 distinct witnesses with its own Fr arithmetic and streams them from
    host memory with 1 and with 2 producer threads, w alone and w + px; every streamed proof equals the blocking entry point's proof of
    the same witness, each of which gs_groth16_verify accepted for its own public input only.  (At 2^20 with 8 witnesses the same
    program is the measurement of the C ABI's ingest ceiling: profiles/r06_c_producer.txt.)"""
    out = c_util.build_and_run("stream_producer.c", ["12", "6", "0.4"], tmp_path, timeout=300)
    lines = [l for l in out.splitlines() if l.startswith("route ")]
    assert out.strip().endswith("OK") and len(lines) == 4, out
    assert all(" 0 mismatches" in l and " 0 hipMalloc 0 hipFree" in l for l in lines), out


def test_stream_stress_c_threads_of_four_kinds_on_one_device(tmp_path):
    """Round 6 changed two things about concurrency: gs_*_end waits for the device OUTSIDE the context's lock (on its own reference to the
    in-flight record) and blocking entry points work in a FREE ticket slot.  tests/c/stream_stress.c: three producers (host-buffer tickets,
    both routes), two blockers (gs_groth16_prove / _prove_witness_host), one canceller (gs_ticket_cancel) and two bystanders (uploads, gs_r1cs_px,
    downloads, frees, memory queries) hammer one key from eight threads for two seconds; every proof and every px equals its single-threaded
    value, three fresh tickets fit afterwards, and NO thread is locked out (the context's lock is first come, first served: with
    std::mutex the two blockers did 27 434 proofs in a minute against one operation of every other thread)."""
    out = c_util.build_and_run("stream_stress.c", ["11", "6", "2.0", "3", "2", "1", "2"], tmp_path, timeout=300)
    assert out.strip().endswith("OK") and "differ from the single-threaded ones: 0" in out and "starved" not in out, out
