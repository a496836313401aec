"""Loaders for tests/golden/wasm_*.json (outputs of the reference's compiled prover, see
oracle/ref_wasm/gen_golden.py).  Converts the utils/base10parsers.go string layouts into the
oracle's Python structures.  Test infrastructure only."""
import json
import os

from oracle import ref_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def g1(p):
    return tuple(int(x) for x in p)


def g2(p):
    return tuple((int(c[0]), int(c[1])) for c in p)


def load(name):
    with open(os.path.join(GOLDEN, "wasm_%s.json" % name)) as f:
        rec = json.load(f)
    rec["circuit"] = json.loads(rec["circuit"])
    rec["setup"] = json.loads(rec["setup"])
    rec["px"] = [int(x) for x in json.loads(rec["px"])]
    # inputs carry big integers as bare JSON numbers; python's json keeps them exact
    inp = json.loads(rec["inputs"])
    rec["w"] = [1] + [int(x) for x in inp["Public"]] + [int(x) for x in inp["Private"]]
    if rec["circuit"]["Constraints"]:
        # the wasm/index.js demo circuit: CalculateWitness (circuitcompiler/circuit.go:158-185) runs
        # the flat code on inputs 3 / 35 -> circuit_test.go:81
        assert rec["w"] == [1, 35, 3]
        rec["w"] = list(O.X3_WITNESS)
    rec["proof"] = json.loads(rec["proof"])
    return rec


def groth_pk(setup):
    s = setup["Pk"]
    pk = O.GrothPk()
    pk.BACDelta = [g1(p) for p in s["BACDelta"]]
    pk.Z = [int(z) for z in s["Z"]]
    pk.G1_Alpha, pk.G1_Beta, pk.G1_Delta = g1(s["G1"]["Alpha"]), g1(s["G1"]["Beta"]), g1(s["G1"]["Delta"])
    pk.G1_At = [g1(p) for p in s["G1"]["At"]]
    pk.G1_BACGamma = [g1(p) for p in s["G1"]["BACGamma"]]
    pk.G2_Beta, pk.G2_Gamma, pk.G2_Delta = g2(s["G2"]["Beta"]), g2(s["G2"]["Gamma"]), g2(s["G2"]["Delta"])
    pk.G2_BACGamma = [g2(p) for p in s["G2"]["BACGamma"]]
    pk.PowersTauDelta = [g1(p) for p in s["PowersTauDelta"]]
    return pk


def pinocchio_pk(setup):
    s = setup["Pk"]
    pk = O.PinocchioPk()
    for k in ("G1T", "A", "C", "Kp", "Ap", "Bp", "Cp"):
        setattr(pk, k, [g1(p) for p in s[k]])
    pk.B = [g2(p) for p in s["B"]]
    pk.Z = [int(z) for z in s["Z"]]
    return pk


def rs_from_stream(stream):
    """groth16.go:231-238 -> fq.go:116-132: r = first 30 bytes, s = next 30, big-endian mod r."""
    return O.FR.RandFromBytes(bytes(stream[0:30])), O.FR.RandFromBytes(bytes(stream[30:60]))
