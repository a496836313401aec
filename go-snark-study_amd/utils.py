"""Wire and on-disk formats of the reference, host side only.

Three encodings of the same objects (keys, proofs):

* the `*String` mirrors of `utils/base10parsers.go` (decimal strings; what the wasm wrapper and
  `cli --wasm` exchange): ArrayBigIntToString … GrothSetupToString / GrothSetupFromString /
  GrothVkFromString / GrothProofToString / GrothProofFromString (base10parsers.go:401-585) and
  SetupToString / SetupFromString / ProofToString / ProofFromString (:135-273, 338-399);
* the bare-number JSON the CLI writes with `json.Marshal` on `*big.Int` structures —
  `trustedsetup.json`, `proofs.json` (cli/main.go:280, 356, 443, 508): same nesting, JSON numbers
  instead of strings (`numbers=True`; Python's json keeps arbitrary-precision integers exact);
* a binary limb container for large keys (additive: a 2^20-constraint key is ~1 GB of decimal
  text): little-endian u64 limbs in exactly the layout `gs_g1_upload` / `gs_g2_upload` take, so a
  key goes file -> np.memmap -> HBM without ever becoming Python integers.

Error behaviour follows the reference: a malformed decimal raises ValueError("error parsing …")
where base10parsers.go returns `errors.New`.  Points are Jacobian triples exactly as stored (no
normalisation on read or write)."""
import json
import struct

import numpy as np

from . import capi, groth16, snark


# ---------------------------------------------------------------------------------------------
# element helpers (base10parsers.go:13-133)
def _to_s(x):
    return str(int(x))


def _from_s(s, what):
    if isinstance(s, bool):
        raise ValueError("error parsing %s" % what)
    if isinstance(s, int):
        return s
    try:
        return int(s, 10)
    except (TypeError, ValueError):
        raise ValueError("error parsing %s" % what) from None


def ArrayBigIntToString(b):             # :13-18
    return [_to_s(x) for x in b]


def ArrayStringToBigInt(s):             # :20-31
    return [_from_s(x, "ArrayStringToBigInt") for x in s]


def BigInt3ToString(b):                 # :44-51
    return [_to_s(b[0]), _to_s(b[1]), _to_s(b[2])]


def String3ToBigInt(s):                 # :33-42
    if len(s) != 3:
        raise ValueError("error parsing String3ToBigInt")
    return tuple(_from_s(x, "String3ToBigInt") for x in s)


def Array3BigIntToString(b):            # :64-70
    return [BigInt3ToString(p) for p in b]


def Array3StringToBigInt(s):            # :53-62
    return [String3ToBigInt(p) for p in s]


def String2ToBigInt(s):                 # :72-83
    if len(s) != 2:
        raise ValueError("error parsing String2ToBigInt")
    return tuple(_from_s(x, "String2ToBigInt") for x in s)


def BigInt32ToString(b):                # :103-113
    return [[_to_s(c[0]), _to_s(c[1])] for c in b]


def String32ToBigInt(s):                # :85-101
    if len(s) != 3:
        raise ValueError("error parsing String32ToBigInt")
    return tuple(String2ToBigInt(c) for c in s)


def Array32BigIntToString(b):           # :126-132
    return [BigInt32ToString(p) for p in b]


def Array32StringToBigInt(s):           # :115-124
    return [String32ToBigInt(p) for p in s]


def _num3(p):
    return [int(p[0]), int(p[1]), int(p[2])]


def _num32(p):
    return [[int(c[0]), int(c[1])] for c in p]


class _Enc:
    """Encoders for one flavour: decimal strings (the *String structs) or bare JSON numbers."""

    def __init__(self, numbers):
        if numbers:
            self.s = int
            self.g1, self.g2 = _num3, _num32
        else:
            self.s = _to_s
            self.g1, self.g2 = BigInt3ToString, BigInt32ToString

    def arr(self, b):
        return [self.s(x) for x in b]

    def a1(self, b):
        return [self.g1(p) for p in b]

    def a2(self, b):
        return [self.g2(p) for p in b]


# ---------------------------------------------------------------------------------------------
# Groth16 (base10parsers.go:401-585)
def GrothProofToString(p, numbers=False):       # :561-567
    e = _Enc(numbers)
    return {"PiA": e.g1(p.PiA), "PiB": e.g2(p.PiB), "PiC": e.g1(p.PiC)}


def GrothProofFromString(s):                    # :568-585
    return groth16.Proof(PiA=String3ToBigInt(s["PiA"]), PiB=String32ToBigInt(s["PiB"]), PiC=String3ToBigInt(s["PiC"]))


def GrothVkToString(vk, numbers=False):
    e = _Enc(numbers)
    return {"IC": e.a1(vk.IC), "G1": {"Alpha": e.g1(vk.G1_Alpha)},
            "G2": {"Beta": e.g2(vk.G2_Beta), "Gamma": e.g2(vk.G2_Gamma), "Delta": e.g2(vk.G2_Delta)}}


def GrothVkFromString(s):                       # :456-479
    return groth16.Vk(IC=Array3StringToBigInt(s["IC"]), G1_Alpha=String3ToBigInt(s["G1"]["Alpha"]),
                      G2_Beta=String32ToBigInt(s["G2"]["Beta"]), G2_Gamma=String32ToBigInt(s["G2"]["Gamma"]),
                      G2_Delta=String32ToBigInt(s["G2"]["Delta"]))


def GrothPkToString(pk, numbers=False):
    e = _Enc(numbers)
    gamma = pk.G2_Gamma if pk.G2_Gamma is not None else ((0, 0), (0, 0), (0, 0))
    return {"BACDelta": e.a1(pk.BACDelta), "Z": e.arr(pk.Z),
            "G1": {"Alpha": e.g1(pk.G1_Alpha), "Beta": e.g1(pk.G1_Beta), "Delta": e.g1(pk.G1_Delta),
                   "At": e.a1(pk.G1_At), "BACGamma": e.a1(pk.G1_BACGamma)},
            "G2": {"Beta": e.g2(pk.G2_Beta), "Gamma": e.g2(gamma), "Delta": e.g2(pk.G2_Delta),
                   "BACGamma": e.a2(pk.G2_BACGamma)},
            "PowersTauDelta": e.a1(pk.PowersTauDelta)}


def GrothPkFromString(s):
    return groth16.Pk(BACDelta=Array3StringToBigInt(s["BACDelta"]), Z=ArrayStringToBigInt(s["Z"]),
                      G1_Alpha=String3ToBigInt(s["G1"]["Alpha"]), G1_Beta=String3ToBigInt(s["G1"]["Beta"]),
                      G1_Delta=String3ToBigInt(s["G1"]["Delta"]), G1_At=Array3StringToBigInt(s["G1"]["At"]),
                      G1_BACGamma=Array3StringToBigInt(s["G1"]["BACGamma"]),
                      G2_Beta=String32ToBigInt(s["G2"]["Beta"]), G2_Gamma=String32ToBigInt(s["G2"]["Gamma"]),
                      G2_Delta=String32ToBigInt(s["G2"]["Delta"]), G2_BACGamma=Array32StringToBigInt(s["G2"]["BACGamma"]),
                      PowersTauDelta=Array3StringToBigInt(s["PowersTauDelta"]))


def GrothSetupToString(pk, vk, numbers=False):  # :435-455 (Toxic is never serialised by the String mirror)
    out = {"Pk": GrothPkToString(pk, numbers), "Vk": GrothVkToString(vk, numbers)}
    if numbers:         # cli/main.go:436-443 marshals a Setup whose Toxic pointers are nil
        out = {"Toxic": {"T": None, "Kalpha": None, "Kbeta": None, "Kgamma": None, "Kdelta": None}, **out}
    return out


def GrothSetupFromString(s):                    # :481-553 -> (Pk, Vk)
    return GrothPkFromString(s["Pk"]), GrothVkFromString(s["Vk"])


# ---------------------------------------------------------------------------------------------
# Pinocchio (base10parsers.go:135-273, 338-399)
_PIN_G1 = ("G1T", "A", "C", "Kp", "Ap", "Bp", "Cp")


def ProofToString(p, numbers=False):            # :349-360
    e = _Enc(numbers)
    return {k: (e.g2 if k == "PiB" else e.g1)(getattr(p, k)) for k in snark.Proof.FIELDS}


def ProofFromString(s):                         # :361-399
    return snark.Proof(**{k: (String32ToBigInt if k == "PiB" else String3ToBigInt)(s[k]) for k in snark.Proof.FIELDS})


_PIN_VK_G2 = ("Vka", "Vkc", "G2Kbg", "G2Kg", "Vkz")


def SetupToString(pk, vk, numbers=False):       # :160-180
    e = _Enc(numbers)
    spk = {k: e.a1(getattr(pk, k)) for k in _PIN_G1}
    spk["B"] = e.a2(pk.B)
    spk["Z"] = e.arr(pk.Z)
    svk = {k: (e.g2 if k in _PIN_VK_G2 else e.g1)(getattr(vk, k)) for k in snark.Vk.FIELDS}
    svk["IC"] = e.a1(vk.IC)
    out = {"Pk": spk, "Vk": svk}
    if numbers:         # cli/main.go:272-280
        out = {"Toxic": {k: None for k in ("T", "Ka", "Kb", "Kc", "Kbeta", "Kgamma", "RhoA", "RhoB", "RhoC")}, **out}
    return out


def SetupFromString(s):                         # :181-273 -> (Pk, Vk)
    p = s["Pk"]
    pk = snark.Pk(B=Array32StringToBigInt(p["B"]), Z=ArrayStringToBigInt(p["Z"]),
                  **{k: Array3StringToBigInt(p[k]) for k in _PIN_G1})
    v = s["Vk"]
    vk = snark.Vk(IC=Array3StringToBigInt(v["IC"]),
                  **{k: (String32ToBigInt if k in _PIN_VK_G2 else String3ToBigInt)(v[k]) for k in snark.Vk.FIELDS})
    return pk, vk


# ---------------------------------------------------------------------------------------------
# files: trustedsetup.json / proofs.json (bare numbers) and their *String.json twins
def WriteJSON(path, obj):
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))


def ReadJSON(path):
    """Either flavour parses with the *FromString functions (they accept numbers and decimal strings)."""
    with open(path) as f:
        return json.load(f)


# ---------------------------------------------------------------------------------------------
# binary limb container
MAGIC = b"GSKEY\x00\x01\x00"
PROTO_GROTH16, PROTO_PINOCCHIO = 1, 2
_HDR = struct.Struct("<8sIIQQ")               # magic, protocol, nsections, nvars, npublic
_SEC = struct.Struct("<24sIIQQ")              # name, u64 words per element, reserved, count, byte offset
_ALIGN = 64


def WriteBinary(path, protocol, nvars, npublic, sections):
    """sections: {name: uint64 array [count, words]} — G1 points 12 words (Jacobian X, Y, Z), G2 24, scalars 4;
    standard (non-Montgomery) form, the layout of include/gosnark_hip.h."""
    names = list(sections)
    arrs = [np.ascontiguousarray(sections[k], dtype="<u8") for k in names]
    for k, a in zip(names, arrs):
        if a.ndim != 2 or len(k.encode()) > 24:
            raise ValueError("section %r: need a [count, words] array and a name of at most 24 bytes" % k)
    off = _HDR.size + _SEC.size * len(names)
    table, offs = [], []
    for k, a in zip(names, arrs):
        off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        offs.append(off)
        table.append(_SEC.pack(k.encode(), a.shape[1], 0, a.shape[0], off))
        off += a.nbytes
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, protocol, len(names), nvars, npublic))
        f.write(b"".join(table))
        for o, a in zip(offs, arrs):
            f.seek(o)
            f.write(a.tobytes())


def ReadBinary(path):
    """-> (protocol, nvars, npublic, {name: read-only np.memmap [count, words] uint64})."""
    with open(path, "rb") as f:
        head = f.read(_HDR.size)
        if len(head) != _HDR.size:
            raise ValueError("error parsing key file: truncated header")
        magic, protocol, nsec, nvars, npublic = _HDR.unpack(head)
        if magic != MAGIC:
            raise ValueError("error parsing key file: bad magic")
        table = f.read(_SEC.size * nsec)
        f.seek(0, 2)
        size = f.tell()
    if len(table) != _SEC.size * nsec:
        raise ValueError("error parsing key file: truncated section table")
    out = {}
    for i in range(nsec):
        name, words, _, count, off = _SEC.unpack_from(table, i * _SEC.size)
        if words == 0 or off + count * words * 8 > size:
            raise ValueError("error parsing key file: section out of bounds")
        shape = (count, words)
        out[name.rstrip(b"\0").decode()] = (np.memmap(path, dtype="<u8", mode="r", offset=off, shape=shape) if count
                                            else np.zeros(shape, dtype=np.uint64))
    return protocol, nvars, npublic, out


_GROTH_ARRAYS = (("G1.At", 0, 12), ("G1.BACGamma", 1, 12), ("G2.BACGamma", 2, 24), ("BACDelta", 3, 12), ("PowersTauDelta", 4, 12))


def _vk_sections(vk):
    return {"Vk.IC": capi.g1_points_to_u64(vk.IC), "Vk.G1.Alpha": capi.g1_points_to_u64([vk.G1_Alpha]),
            "Vk.G2": capi.g2_points_to_u64([vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta])}


def GrothSetupToBinary(path, circuit, pk, vk):
    """pk: groth16.Pk (host integers) or groth16.DevicePk (resident; read back through gs_groth16_pk_export)."""
    sec = {}
    if isinstance(pk, groth16.DevicePk):
        lib = capi.load_library()
        for name, which, words in _GROTH_ARRAYS:
            count = pk.nvars - 1 if which == 4 else pk.nvars
            a = np.zeros((count, words), dtype=np.uint64)
            capi.check(lib.gs_groth16_pk_export(capi.Handle(pk.handle.h), which, capi.ptr64(a), count))
            sec[name] = a
        singles = np.zeros(84, dtype=np.uint64)
        capi.check(lib.gs_groth16_pk_export(capi.Handle(pk.handle.h), 5, capi.ptr64(singles), 5))
        sec["G1.ABD"] = singles[:36].reshape(3, 12)
        sec["G2.BD"] = singles[36:].reshape(2, 24)
        z = np.zeros((pk.nvars - 1, 4), dtype=np.uint64)
        capi.check(lib.gs_groth16_pk_export(capi.Handle(pk.handle.h), 6, capi.ptr64(z), z.shape[0]))
        sec["Z"] = z
        ne = capi.pk_eval_count(pk.handle)
        if ne:           # optional section: the evaluation-basis copy of PowersTauDelta (only whoever knew tau can produce it)
            e = np.zeros((ne, 12), dtype=np.uint64)
            capi.check(lib.gs_groth16_pk_export(capi.Handle(pk.handle.h), 7, capi.ptr64(e), ne))
            sec["PowersTauDeltaEval"] = e
    else:
        sec["G1.At"] = capi.g1_points_to_u64(pk.G1_At)
        sec["G1.BACGamma"] = capi.g1_points_to_u64(pk.G1_BACGamma)
        sec["G2.BACGamma"] = capi.g2_points_to_u64(pk.G2_BACGamma)
        sec["BACDelta"] = capi.g1_points_to_u64(pk.BACDelta)
        sec["PowersTauDelta"] = capi.g1_points_to_u64(pk.PowersTauDelta)
        sec["G1.ABD"] = capi.g1_points_to_u64([pk.G1_Alpha, pk.G1_Beta, pk.G1_Delta])
        sec["G2.BD"] = capi.g2_points_to_u64([pk.G2_Beta, pk.G2_Delta])
        sec["Z"] = capi.ints_to_u64([z % groth16.R for z in pk.Z])
    if vk is not None:
        sec.update(_vk_sections(vk))
    WriteBinary(path, PROTO_GROTH16, circuit.NVars, circuit.NPublic, sec)


def _g1_tuples(a):
    v = capi.u64_to_ints(a)
    return [(v[3 * i], v[3 * i + 1], v[3 * i + 2]) for i in range(len(v) // 3)]


def _g2_tuples(a):
    v = capi.u64_to_ints(a)
    return [((v[6 * i], v[6 * i + 1]), (v[6 * i + 2], v[6 * i + 3]), (v[6 * i + 4], v[6 * i + 5])) for i in range(len(v) // 6)]


def GrothVkFromBinary(path):
    protocol, _, _, sec = ReadBinary(path)
    if protocol != PROTO_GROTH16 or "Vk.IC" not in sec:
        raise ValueError("error parsing key file: no Groth16 verification key inside")
    g2 = _g2_tuples(sec["Vk.G2"])
    return groth16.Vk(IC=_g1_tuples(sec["Vk.IC"]), G1_Alpha=_g1_tuples(sec["Vk.G1.Alpha"])[0],
                      G2_Beta=g2[0], G2_Gamma=g2[1], G2_Delta=g2[2])


def GrothPkFromBinary(path):
    """Host-integer groth16.Pk (small keys / tests).  Large keys: UploadGrothPkBinary."""
    protocol, nvars, npublic, sec = ReadBinary(path)
    if protocol != PROTO_GROTH16:
        raise ValueError("error parsing key file: not a Groth16 key")
    abd, bd = _g1_tuples(sec["G1.ABD"]), _g2_tuples(sec["G2.BD"])
    gamma = _g2_tuples(sec["Vk.G2"])[1] if "Vk.G2" in sec else None
    pk = groth16.Pk(BACDelta=_g1_tuples(sec["BACDelta"]), Z=capi.u64_to_ints(sec["Z"]), G1_Alpha=abd[0], G1_Beta=abd[1],
                    G1_Delta=abd[2], G1_At=_g1_tuples(sec["G1.At"]), G1_BACGamma=_g1_tuples(sec["G1.BACGamma"]),
                    G2_Beta=bd[0], G2_Delta=bd[1], G2_BACGamma=_g2_tuples(sec["G2.BACGamma"]),
                    PowersTauDelta=_g1_tuples(sec["PowersTauDelta"]), G2_Gamma=gamma)
    return groth16.Circuit(nvars, npublic), pk


def UploadGrothPkBinary(path, shard=None):
    """file -> memmap -> HBM: the arrays never become Python integers.  -> (Circuit, DevicePk).
    shard = (index, count): map and upload ONLY that rank's slices of the five arrays (a key slice for
    groth16.prove_partials / prove_sharded, gs_groth16_pk_create_shard) -- 1/count of the file is read."""
    protocol, nvars, npublic, sec = ReadBinary(path)
    if protocol != PROTO_GROTH16:
        raise ValueError("error parsing key file: not a Groth16 key")
    nptd = sec["PowersTauDelta"].shape[0]
    if shard is None:
        wlo, whi, hlo, hhi = 0, nvars, 0, nptd
    else:
        wlo, whi = groth16._shard_range(nvars, shard[1], shard[0])
        hlo, hhi = groth16._shard_range(nptd, shard[1], shard[0])
    up1 = lambda k, lo, hi: capi.g1_upload(np.ascontiguousarray(sec[k][lo:hi], dtype=np.uint64))     # noqa: E731
    at, b1, cd = up1("G1.At", wlo, whi), up1("G1.BACGamma", wlo, whi), up1("BACDelta", wlo, whi)
    pt = up1("PowersTauDelta", hlo, hhi)
    b2 = capi.g2_upload(np.ascontiguousarray(sec["G2.BACGamma"][wlo:whi], dtype=np.uint64))
    abd, bd = _g1_tuples(sec["G1.ABD"]), _g2_tuples(sec["G2.BD"])
    z = np.ascontiguousarray(sec["Z"], dtype=np.uint64)
    if shard is None:
        dev = groth16.device_pk_from_handles(at, b1, b2, cd, pt, abd[0], abd[1], abd[2], bd[0], bd[1], z, nvars, npublic)
        if "PowersTauDeltaEval" in sec:          # the witness route then runs its h-MSM over H's values (gs_groth16_pk_set_eval)
            e = capi.g1_upload(np.ascontiguousarray(sec["PowersTauDeltaEval"], dtype=np.uint64))
            capi.check(capi.load_library().gs_groth16_pk_set_eval(capi.Handle(dev.handle.h), capi.Handle(e.h)))
            e.free()
    else:
        dev = groth16.device_pk_shard_from_handles(at, b1, b2, cd, pt, abd[0], abd[1], abd[2], bd[0], bd[1], z, nvars, npublic, nptd,
                                                   shard[0], shard[1])
    return groth16.Circuit(nvars, npublic), dev


# ---- Pinocchio keys in the same container ---------------------------------------------------------
_PIN_ARRAYS = (("A", 0, 12), ("Ap", 1, 12), ("B", 2, 24), ("Bp", 3, 12), ("C", 4, 12), ("Cp", 5, 12), ("Kp", 6, 12), ("G1T", 7, 12))


def SetupToBinary(path, circuit, pk, vk=None):
    """snark.Pk (host integers) or snark.DevicePk (resident; gs_pinocchio_pk_export 0..8) [+ snark.Vk] -> binary container.
    Note: a resident key's A / Ap hold infinity for i <= NPublic (what the prover sums, snark.go:265)."""
    sec = {}
    if isinstance(pk, snark.DevicePk):
        lib = capi.load_library()
        for name, which, words in _PIN_ARRAYS:
            count = pk.nvars - 1 if which == 7 else pk.nvars
            a = np.zeros((count, words), dtype=np.uint64)
            capi.check(lib.gs_pinocchio_pk_export(capi.Handle(pk.h), which, capi.ptr64(a), count))
            sec[name] = a
        z = np.zeros((pk.nvars - 1, 4), dtype=np.uint64)
        capi.check(lib.gs_pinocchio_pk_export(capi.Handle(pk.h), 8, capi.ptr64(z), z.shape[0]))
        sec["Z"] = z
        ne = capi.pk_eval_count(pk.handle)
        if ne:           # optional section: the evaluation-basis copy of G1T
            e = np.zeros((ne, 12), dtype=np.uint64)
            capi.check(lib.gs_pinocchio_pk_export(capi.Handle(pk.h), 9, capi.ptr64(e), ne))
            sec["G1TEval"] = e
    else:
        for name, _, words in _PIN_ARRAYS:
            pts = getattr(pk, name)
            sec[name] = capi.g2_points_to_u64(pts) if words == 24 else capi.g1_points_to_u64(pts)
        sec["Z"] = capi.ints_to_u64([z % groth16.R for z in pk.Z])
    if vk is not None:
        sec["Vk.IC"] = capi.g1_points_to_u64(vk.IC)
        sec["Vk.G1"] = capi.g1_points_to_u64([vk.Vkb, vk.G1Kbg])
        sec["Vk.G2"] = capi.g2_points_to_u64([vk.Vka, vk.Vkc, vk.G2Kbg, vk.G2Kg, vk.Vkz])
    WriteBinary(path, PROTO_PINOCCHIO, circuit.NVars, circuit.NPublic, sec)


def SetupFromBinary(path):
    """-> (Circuit, snark.Pk as host integers, snark.Vk or None)."""
    protocol, nvars, npublic, sec = ReadBinary(path)
    if protocol != PROTO_PINOCCHIO:
        raise ValueError("error parsing key file: not a Pinocchio key")
    pk = snark.Pk(B=_g2_tuples(sec["B"]), Z=capi.u64_to_ints(sec["Z"]), **{k: _g1_tuples(sec[k]) for k in _PIN_G1})
    vk = None
    if "Vk.IC" in sec:
        g1, g2 = _g1_tuples(sec["Vk.G1"]), _g2_tuples(sec["Vk.G2"])
        vk = snark.Vk(IC=_g1_tuples(sec["Vk.IC"]), Vkb=g1[0], G1Kbg=g1[1], Vka=g2[0], Vkc=g2[1], G2Kbg=g2[2], G2Kg=g2[3], Vkz=g2[4])
    return snark.Circuit(nvars, npublic), pk, vk


def UploadPkBinary(path):
    """Pinocchio key: file -> memmap -> HBM (gs_g1_upload / gs_g2_upload on the mapped sections).  -> (Circuit, DevicePk)."""
    import ctypes
    protocol, nvars, npublic, sec = ReadBinary(path)
    if protocol != PROTO_PINOCCHIO:
        raise ValueError("error parsing key file: not a Pinocchio key")
    g1 = {k: capi.g1_upload(np.ascontiguousarray(sec[k], dtype=np.uint64)) for k in ("A", "Ap", "Bp", "C", "Cp", "Kp", "G1T")}
    b2 = capi.g2_upload(np.ascontiguousarray(sec["B"], dtype=np.uint64))
    z = np.ascontiguousarray(sec["Z"], dtype=np.uint64)
    h = capi.Handle(0)
    H = lambda x: capi.Handle(x.h)   # noqa: E731
    capi.check(capi.load_library().gs_pinocchio_pk_create(
        H(g1["A"]), H(g1["Ap"]), H(b2), H(g1["Bp"]), H(g1["C"]), H(g1["Cp"]), H(g1["Kp"]), H(g1["G1T"]),
        capi.ptr64(z), z.shape[0], nvars, npublic, ctypes.byref(h)))
    dev = snark.DevicePk(capi.DeviceHandle(h.value), nvars, npublic)
    if "G1TEval" in sec:
        e = capi.g1_upload(np.ascontiguousarray(sec["G1TEval"], dtype=np.uint64))
        capi.check(capi.load_library().gs_pinocchio_pk_set_eval(capi.Handle(dev.h), capi.Handle(e.h)))
        e.free()
    return snark.Circuit(nvars, npublic), dev
