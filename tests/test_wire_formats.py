"""the reference's wire formats (utils/base10parsers.go *String mirrors, the CLI's bare-number JSON) and the
binary limb container.  The fixtures are the strings the reference's own compiled wasm consumed and produced
(tests/golden/wasm_*.json), so a lossless round trip here means the reference reads what we write."""
import json
import os

import numpy as np
import pytest

import gosnark_amd  # noqa: F401
from gosnark_amd import groth16, utils

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rec(name):
    with open(os.path.join(GOLDEN, "wasm_%s.json" % name)) as f:
        r = json.load(f)
    return json.loads(r["setup"]), json.loads(r["proof"])


@pytest.mark.parametrize("name", ["groth_x3", "groth_rand_m9", "groth_rand_m17"])
def test_groth_string_round_trip(name):
    setup, proof = rec(name)
    pk, vk = utils.GrothSetupFromString(setup)
    assert utils.GrothSetupToString(pk, vk) == setup                      # byte-for-byte the reference's strings
    assert utils.GrothProofToString(utils.GrothProofFromString(proof)) == proof
    assert isinstance(pk.G1_At[0][0], int) and len(pk.G2_BACGamma[0]) == 3
    vk2 = utils.GrothVkFromString(setup["Vk"])
    assert vk2.IC == vk.IC and vk2.G2_Gamma == vk.G2_Gamma


@pytest.mark.parametrize("name", ["pinocchio_x3_setup", "pinocchio_rand_m9"])
def test_pinocchio_string_round_trip(name):
    setup, proof = rec(name)
    pk, vk = utils.SetupFromString(setup)
    assert utils.SetupToString(pk, vk) == setup
    assert utils.ProofToString(utils.ProofFromString(proof)) == proof


def test_bare_number_json_like_the_cli(tmp_path):
    """cli/main.go:443,508 json.Marshal(*big.Int) -> bare JSON numbers, same nesting, Toxic: nulls."""
    setup, proof = rec("groth_x3")
    pk, vk = utils.GrothSetupFromString(setup)
    p = tmp_path / "trustedsetup.json"
    utils.WriteJSON(str(p), utils.GrothSetupToString(pk, vk, numbers=True))
    text = p.read_text()
    assert '"' + setup["Pk"]["G1"]["At"][1][0] + '"' not in text and setup["Pk"]["G1"]["At"][1][0] in text
    back = utils.ReadJSON(str(p))
    assert back["Toxic"] == {"T": None, "Kalpha": None, "Kbeta": None, "Kgamma": None, "Kdelta": None}
    assert list(back.keys()) == ["Toxic", "Pk", "Vk"]
    pk2, vk2 = utils.GrothSetupFromString(back)
    assert utils.GrothSetupToString(pk2, vk2) == setup
    q = tmp_path / "proofs.json"
    utils.WriteJSON(str(q), utils.GrothProofToString(utils.GrothProofFromString(proof), numbers=True))
    assert utils.GrothProofToString(utils.GrothProofFromString(utils.ReadJSON(str(q)))) == proof
    ps, pp = rec("pinocchio_x3_setup")
    ppk, pvk = utils.SetupFromString(ps)
    s2 = json.loads(json.dumps(utils.SetupToString(ppk, pvk, numbers=True)))
    assert len(s2["Toxic"]) == 9 and utils.SetupToString(*utils.SetupFromString(s2)) == ps
    assert utils.ProofToString(utils.ProofFromString(json.loads(json.dumps(utils.ProofToString(utils.ProofFromString(pp), True))))) == pp


def test_parse_errors_like_the_reference():
    """base10parsers.go returns errors.New("error parsing ...") when SetString fails."""
    with pytest.raises(ValueError, match="error parsing"):
        utils.ArrayStringToBigInt(["12", "0x1f"])
    with pytest.raises(ValueError, match="error parsing"):
        utils.String3ToBigInt(["1", "2"])
    with pytest.raises(ValueError, match="error parsing"):
        utils.String32ToBigInt([["1", "2"], ["3", "x"], ["1", "0"]])
    _, proof = rec("groth_x3")
    proof["PiB"][1][0] = "12a"
    with pytest.raises(ValueError, match="error parsing"):
        utils.GrothProofFromString(proof)
    assert utils.ArrayBigIntToString([0, 5, 2 ** 200]) == ["0", "5", str(2 ** 200)]
    assert utils.ArrayStringToBigInt(["-7"]) == [-7]         # big.Int.SetString takes a sign


def test_binary_container_round_trip(tmp_path):
    setup, _ = rec("groth_rand_m17")
    pk, vk = utils.GrothSetupFromString(setup)
    circ = groth16.Circuit(len(pk.G1_At), 1)
    p = str(tmp_path / "key.gskey")
    utils.GrothSetupToBinary(p, circ, pk, vk)
    proto, nvars, npublic, sec = utils.ReadBinary(p)
    assert (proto, nvars, npublic) == (utils.PROTO_GROTH16, len(pk.G1_At), 1)
    assert sec["G1.At"].shape == (nvars, 12) and sec["G2.BACGamma"].shape == (nvars, 24) and sec["Z"].shape[1] == 4
    assert all(a.dtype == np.dtype("<u8") for a in sec.values())
    circ2, pk2 = utils.GrothPkFromBinary(p)
    assert (circ2.NVars, circ2.NPublic) == (nvars, 1)
    assert utils.GrothSetupToString(pk2, utils.GrothVkFromBinary(p)) == setup
    raw = bytearray(open(p, "rb").read())
    raw[0] ^= 1
    bad = str(tmp_path / "bad.gskey")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError, match="bad magic"):
        utils.ReadBinary(bad)
    open(bad, "wb").write(bytes(raw[:40]))
    with pytest.raises(ValueError, match="error parsing key file"):
        utils.ReadBinary(bad)
    trunc = bytearray(open(p, "rb").read())
    trunc[0:8] = utils.MAGIC
    open(bad, "wb").write(bytes(trunc[:len(trunc) - 64]))
    with pytest.raises(ValueError, match="out of bounds"):
        utils.ReadBinary(bad)


def test_pinocchio_binary_container_round_trip(tmp_path):
    from gosnark_amd import snark
    setup, _ = rec("pinocchio_rand_m9")
    pk, vk = utils.SetupFromString(setup)
    p = str(tmp_path / "pin.gskey")
    utils.SetupToBinary(p, snark.Circuit(len(pk.A), 1), pk, vk)
    circ, pk2, vk2 = utils.SetupFromBinary(p)
    assert (circ.NVars, circ.NPublic) == (len(pk.A), 1)
    assert utils.SetupToString(pk2, vk2) == setup
    with pytest.raises(ValueError, match="not a Groth16 key"):
        utils.GrothPkFromBinary(p)
