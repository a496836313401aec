#!/usr/bin/env python3
"""ORACLE TOOLING (test infrastructure, not product code).

Generates tests/golden/wasm_*.json by running the REFERENCE'S OWN compiled prover
(/root/reference/wasm/go-snark.wasm, see go_wasm_host.js) on inputs built here.
Run in the build container only (needs /root/reference and node):

    python3 oracle/ref_wasm/gen_golden.py

Each golden file stores the exact JSON texts fed to the reference (circuit / setup / px /
inputs in the utils/base10parsers.go string layouts), the pinned crypto/rand byte stream,
and the proof JSON the reference returned (raw Jacobian) plus its verify verdicts.
Consumers: tests/test_oracle_vs_reference.py (pins oracle/ref_py.py and oracle/gs_oracle.c
bit-for-bit in Jacobian coordinates) and the -m gpu parity tests (affine comparison).
"""
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_py as O  # noqa: E402

WASM = "/root/reference/wasm/go-snark.wasm"
OUT = os.path.join(ROOT, "tests", "golden")


def s3(p):
    return [str(p[0]), str(p[1]), str(p[2])]


def s32(p):
    return [[str(c[0]), str(c[1])] for c in p]


def groth_setup_text(pk, vk):
    """utils/base10parsers.go:401-434 GrothSetupString"""
    d = {
        "Pk": {
            "BACDelta": [s3(p) for p in pk.BACDelta],
            "Z": [str(z) for z in pk.Z],
            "G1": {"Alpha": s3(pk.G1_Alpha), "Beta": s3(pk.G1_Beta), "Delta": s3(pk.G1_Delta),
                   "At": [s3(p) for p in pk.G1_At], "BACGamma": [s3(p) for p in pk.G1_BACGamma]},
            "G2": {"Beta": s32(pk.G2_Beta), "Gamma": s32(pk.G2_Gamma), "Delta": s32(pk.G2_Delta),
                   "BACGamma": [s32(p) for p in pk.G2_BACGamma]},
            "PowersTauDelta": [s3(p) for p in pk.PowersTauDelta],
        },
        "Vk": {
            "IC": [s3(p) for p in vk.IC],
            "G1": {"Alpha": s3(vk.G1_Alpha)},
            "G2": {"Beta": s32(vk.G2_Beta), "Gamma": s32(vk.G2_Gamma), "Delta": s32(vk.G2_Delta)},
        },
    }
    return json.dumps(d)


def circuit_text(nvars, npublic, r1cs=None):
    """utils/base10parsers.go:259-273 CircuitString.  Constraints: [] makes CalculateWitness
    return [1, public..., private...] (circuitcompiler/circuit.go:165-172), i.e. w is injected."""
    names = ["one"] + ["pub%d" % i for i in range(npublic)] + ["s%d" % i for i in range(nvars - 1 - npublic)]
    a, b, c = r1cs if r1cs else ([], [], [])
    d = {"NVars": nvars, "NPublic": npublic, "NSignals": nvars,
         "PrivateInputs": names[1 + npublic:], "PublicInputs": names[1:1 + npublic], "Signals": names,
         "Witness": None, "Constraints": [],
         "R1CS": {"A": [[str(x) for x in row] for row in a], "B": [[str(x) for x in row] for row in b],
                  "C": [[str(x) for x in row] for row in c]}}
    return json.dumps(d)


def inputs_text(w, npublic):
    """circuitcompiler/circuit.go:151-154 Inputs: bare JSON integers (built by hand, exact)."""
    pub = ",".join(str(x) for x in w[1:1 + npublic])
    prv = ",".join(str(x) for x in w[1 + npublic:])
    return '{"Private":[%s],"Public":[%s]}' % (prv, pub)


def zpoly(nvars):
    z = [1]
    for i in range(1, nvars - 1):       # groth16.go:122-131
        z = O.PF.Mul(z, [O.FR.Neg(i), 1])
    return z


def rand_g1(rng, allow_inf=True):
    k = rng.randrange(0, 12)
    if allow_inf and k == 0:
        return O.G1_ZERO
    return O.G1.MulScalar(O.G1_GEN, rng.randrange(1, O.R))


def rand_g2(rng, allow_inf=True):
    k = rng.randrange(0, 12)
    if allow_inf and k == 0:
        return O.G2_ZERO
    return O.G2.MulScalar(O.G2_GEN, rng.randrange(1, O.R))


def rand_stream(seed_mul=37, seed_add=11, n=60):
    return [(seed_mul * i + seed_add) & 0xff for i in range(n)]


def job_groth_x3():
    """SURVEY App. B2 recipe: x^3+x+5, toxic X_k = bytes((i*k+7)&0xff for i<30) mod r for
    k in (3,5,7,11,13); prover randomness byte i = (37 i + 11) & 0xff."""
    toxic = tuple(int.from_bytes(bytes((i * k + 7) & 0xff for i in range(30)), "big") % O.R
                  for k in (3, 5, 7, 11, 13))
    alphas, betas, gammas, _ = O.PF.R1CSToQAP(O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)
    _, _, _, px = O.PF.CombinePolynomials(O.X3_WITNESS, alphas, betas, gammas)
    pk, vk = O.groth16_GenerateTrustedSetup(8, 1, alphas, betas, gammas, toxic)
    return dict(name="groth_x3", kind="groth",
                circuit=circuit_text(8, 1, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)),
                setup=groth_setup_text(pk, vk), px=json.dumps([str(x) for x in px]),
                inputs=inputs_text(O.X3_WITNESS, 1), rand=rand_stream(), verify=["[35]", "[34]"])


def job_groth_rand(m, seed):
    """Random full-width instance: m variables, n = m-1 constraints' worth of px (2n-1 coeffs),
    random Jacobian pk points (some at infinity), inexact division -- exercises every MSM and
    Div of groth16.go:243-275 without going through setup."""
    rng = random.Random(seed)
    n = m - 1
    pk, vk = O.GrothPk(), O.GrothVk()
    pk.Z = zpoly(m)
    pk.G1_Alpha, pk.G1_Beta, pk.G1_Delta = (rand_g1(rng, False) for _ in range(3))
    pk.G2_Beta, pk.G2_Gamma, pk.G2_Delta = (rand_g2(rng, False) for _ in range(3))
    pk.G1_At = [rand_g1(rng) for _ in range(m)]
    pk.G1_BACGamma = [rand_g1(rng) for _ in range(m)]
    pk.G2_BACGamma = [rand_g2(rng) for _ in range(m)]
    pk.BACDelta = [O.G1_ZERO, O.G1_ZERO] + [rand_g1(rng) for _ in range(m - 2)]
    pk.PowersTauDelta = [rand_g1(rng) for _ in range(len(pk.Z))]
    vk.IC = [rand_g1(rng, False) for _ in range(2)]
    vk.G1_Alpha, vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta = pk.G1_Alpha, pk.G2_Beta, pk.G2_Gamma, pk.G2_Delta
    w = [1] + [rng.randrange(0, O.R) for _ in range(m - 1)]
    w[3] = 0                      # a zero scalar and a small one
    w[4] = 1
    px = [rng.randrange(0, O.R) for _ in range(2 * n - 1)]
    return dict(name="groth_rand_m%d" % m, kind="groth", circuit=circuit_text(m, 1),
                setup=groth_setup_text(pk, vk), px=json.dumps([str(x) for x in px]),
                inputs=inputs_text(w, 1), rand=rand_stream(101, 7 + m), verify=[])


def job_pinocchio_rand(m, seed):
    """Random instance for snark.go:254-289 in the SetupString layout (base10parsers.go:135-158)."""
    rng = random.Random(seed)
    n = m - 1
    z = zpoly(m)
    g1s = lambda k: [s3(rand_g1(rng)) for _ in range(k)]  # noqa: E731
    d = {"Pk": {"G1T": g1s(len(z)), "A": g1s(m), "B": [s32(rand_g2(rng)) for _ in range(m)], "C": g1s(m),
                "Kp": g1s(m), "Ap": g1s(m), "Bp": g1s(m), "Cp": g1s(m), "Z": [str(x) for x in z]},
         "Vk": {"Vka": s32(rand_g2(rng, False)), "Vkb": s3(rand_g1(rng, False)), "Vkc": s32(rand_g2(rng, False)),
                "IC": g1s(2), "G1Kbg": s3(rand_g1(rng, False)), "G2Kbg": s32(rand_g2(rng, False)),
                "G2Kg": s32(rand_g2(rng, False)), "Vkz": s32(rand_g2(rng, False))}}
    w = [1] + [rng.randrange(0, O.R) for _ in range(m - 1)]
    px = [rng.randrange(0, O.R) for _ in range(2 * n - 1)]
    return dict(name="pinocchio_rand_m%d" % m, kind="pinocchio", circuit=circuit_text(m, 1),
                setup=json.dumps(d), px=json.dumps([str(x) for x in px]), inputs=inputs_text(w, 1),
                rand=None, verify=[])


def job_pinocchio_x3_setup():
    """snark.GenerateTrustedSetup restated (oracle) on x^3+x+5 with toxic X_k = bytes((i*k+9)&0xff for i<30) mod r,
    k in (3,5,7,11,13,17,19,23); the reference proves with that key and its own VerifyProof must accept [35], reject [34]."""
    toxic = tuple(int.from_bytes(bytes((i * k + 9) & 0xff for i in range(30)), "big") % O.R
                  for k in (3, 5, 7, 11, 13, 17, 19, 23))
    alphas, betas, gammas, _ = O.PF.R1CSToQAP(O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)
    _, _, _, px = O.PF.CombinePolynomials(O.X3_WITNESS, alphas, betas, gammas)
    pk, vk = O.snark_GenerateTrustedSetup(8, 1, alphas, betas, gammas, toxic)
    d = {"Pk": {"G1T": [s3(p) for p in pk.G1T], "A": [s3(p) for p in pk.A], "B": [s32(p) for p in pk.B],
                "C": [s3(p) for p in pk.C], "Kp": [s3(p) for p in pk.Kp], "Ap": [s3(p) for p in pk.Ap],
                "Bp": [s3(p) for p in pk.Bp], "Cp": [s3(p) for p in pk.Cp], "Z": [str(x) for x in pk.Z]},
         "Vk": {"Vka": s32(vk.Vka), "Vkb": s3(vk.Vkb), "Vkc": s32(vk.Vkc), "IC": [s3(p) for p in vk.IC],
                "G1Kbg": s3(vk.G1Kbg), "G2Kbg": s32(vk.G2Kbg), "G2Kg": s32(vk.G2Kg), "Vkz": s32(vk.Vkz)}}
    return dict(name="pinocchio_x3_setup", kind="pinocchio",
                circuit=circuit_text(8, 1, (O.X3_R1CS_A, O.X3_R1CS_B, O.X3_R1CS_C)),
                setup=json.dumps(d), px=json.dumps([str(x) for x in px]), inputs=inputs_text(O.X3_WITNESS, 1),
                rand=None, verify=["[35]", "[34]"], toxic=[str(t) for t in toxic])


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    jobs = [
        dict(name="pinocchio_x3_fixture", kind="fixture", inputs='{"Private":[3],"Public":[35]}',
             verify=["[35]", "[34]"]),
        job_groth_x3(),
        job_groth_rand(9, 1009),
        job_groth_rand(17, 1017),
        job_pinocchio_rand(9, 2009),
        job_pinocchio_x3_setup(),
    ]
    if only:
        jobs = [j for j in jobs if j["name"] in only]
    with tempfile.TemporaryDirectory() as td:
        jp, op = os.path.join(td, "jobs.json"), os.path.join(td, "out.json")
        with open(jp, "w") as f:
            json.dump(jobs, f)
        subprocess.check_call(["node", os.path.join(HERE, "run_jobs.js"), WASM, jp, op])
        with open(op) as f:
            res = json.load(f)
    for rec in res:
        rec["generator"] = "oracle/ref_wasm/gen_golden.py via /root/reference/wasm/go-snark.wasm (go1.12 js/wasm)"
        with open(os.path.join(OUT, "wasm_%s.json" % rec["name"]), "w") as f:
            json.dump(rec, f)
        print("wrote", rec["name"], "proof bytes", len(rec["proof"]))


if __name__ == "__main__":
    main()
