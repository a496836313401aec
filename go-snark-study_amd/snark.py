"""Mirror of the reference's Pinocchio prover interface (snark.go:16-26, 59-69, 254-289)."""
import ctypes

import numpy as np

from . import capi
from .groth16 import Circuit, R   # noqa: F401


class Pk:
    """snark.Pk (snark.go:16-26)."""

    def __init__(self, G1T, A, B, C, Kp, Ap, Bp, Cp, Z):
        self.G1T, self.A, self.B, self.C = G1T, A, B, C
        self.Kp, self.Ap, self.Bp, self.Cp, self.Z = Kp, Ap, Bp, Cp, Z
        self._dev = None


class Proof:
    """snark.Proof (snark.go:59-69)."""
    FIELDS = ("PiA", "PiAp", "PiB", "PiBp", "PiC", "PiCp", "PiH", "PiKp")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, kw[k])


def UploadPk(pk, circuit):
    if pk._dev is not None:
        return pk._dev
    capi.init()
    g1 = {k: capi.g1_upload(capi.g1_points_to_u64(getattr(pk, k))) for k in ("A", "Ap", "Bp", "C", "Cp", "Kp", "G1T")}
    b2 = capi.g2_upload(capi.g2_points_to_u64(pk.B))
    z = capi.ints_to_u64([x % R for x in pk.Z])
    h = capi.Handle(0)
    H = lambda x: capi.Handle(x.h)   # noqa: E731
    capi.check(capi.load_library().gs_pinocchio_pk_create(
        H(g1["A"]), H(g1["Ap"]), H(b2), H(g1["Bp"]), H(g1["C"]), H(g1["Cp"]), H(g1["Kp"]), H(g1["G1T"]),
        capi.ptr64(z), z.shape[0], circuit.NVars, circuit.NPublic, ctypes.byref(h)))
    pk._dev = capi.DeviceHandle(h.value)
    return pk._dev


GS_ERR_BUSY = -6


def _host_scalars(x, what):
    """ints (reduced mod r here, negatives rejected) or an [n, 4] uint64 limb array (as it is) -> contiguous [n, 4] uint64"""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    if any(v < 0 for v in x):
        raise ValueError("negative %s values are not supported" % what)
    return capi.ints_to_u64([v % R for v in x])


def GenerateProofs(circuit, pk, w, px):
    """snark.GenerateProofs(circuit, pk, w, px) (snark.go:254-289).  Deterministic.  Round 6 (as go/snarkhip.GenerateProofs): a host-buffer
    ticket collected at once (gs_pinocchio_prove_host_begin + gs_pinocchio_prove_end); the blocking entry point when all slots are taken."""
    dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
    wa, pa = _host_scalars(w, "witness"), _host_scalars(px, "px")
    try:
        return prove_end(prove_host_begin(dev, wa, pa))
    except capi.GosnarkHipError as e:
        if e.code != GS_ERR_BUSY:
            raise
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove(capi.Handle(dev.h), capi.ptr64(wa), wa.shape[0], capi.ptr64(pa), pa.shape[0],
                                                      capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def GenerateProofsFromWitness(circuit, pk, dev_r1cs, w):
    """go/snarkhip.GenerateProofsFromWitness: witness -> proof against the circuit's resident sparse R1CS, a host-buffer ticket collected at once."""
    dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
    wa = _host_scalars(w, "witness")
    try:
        return prove_end(prove_witness_host_begin(dev, dev_r1cs, wa))
    except capi.GosnarkHipError as e:
        if e.code != GS_ERR_BUSY:
            raise
    return prove_from_witness_host(dev, dev_r1cs, wa)


class Prover:
    """The streaming drop-in (go/snarkhip.Prover; see groth16.Prover): Submit(w[, px]) / Collect(), three proofs in flight."""
    MaxInFlight = 3

    def __init__(self, circuit, pk, dev_r1cs=None):
        self.dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
        self.r1cs = dev_r1cs
        self.tickets, self.done = [], []

    def _collect_oldest(self):
        self.done.append(prove_end(self.tickets.pop(0)))

    def Submit(self, w, px=None):
        if px is None and self.r1cs is None:
            raise ValueError("this prover has no resident R1CS: Submit needs px")
        wa = _host_scalars(w, "witness")
        pa = None if px is None else _host_scalars(px, "px")
        while True:
            if len(self.tickets) >= self.MaxInFlight:
                self._collect_oldest()
            try:
                t = prove_witness_host_begin(self.dev, self.r1cs, wa) if pa is None else prove_host_begin(self.dev, wa, pa)
            except capi.GosnarkHipError as e:
                if e.code == GS_ERR_BUSY and self.tickets:
                    self._collect_oldest()
                    continue
                raise
            self.tickets.append(t)
            return

    def InFlight(self):
        return len(self.tickets) + len(self.done)

    def Collect(self):
        if not self.done:
            if not self.tickets:
                raise ValueError("Collect without a submitted proof")
            self._collect_oldest()
        return self.done.pop(0)

    def Close(self):
        for t in self.tickets:
            capi.ticket_cancel(t)
        self.tickets, self.done = [], []


def NewProver(circuit, pk, dev_r1cs=None):
    return Prover(circuit, pk, dev_r1cs)


def _proof_from_words(out, inf):
    v = capi.u64_to_ints(out)
    res, pos = {}, 0
    for i, k in enumerate(Proof.FIELDS):
        if k == "PiB":
            res[k] = ((0, 0), (0, 0), (0, 0)) if inf[i] else ((v[pos], v[pos + 1]), (v[pos + 2], v[pos + 3]), (1, 0))
            pos += 4
        else:
            res[k] = (0, 0, 0) if inf[i] else (v[pos], v[pos + 1], 1)
            pos += 2
    return Proof(**res)


def prove_resident(dev_pk, w_handle, px_handle):
    """snark.GenerateProofs with the key, w and px already resident in HBM (gs_pinocchio_prove_resident)."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_resident(capi.Handle(dev_pk.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                               capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_from_witness(dev_pk, dev_r1cs, w_handle):
    """Sparse R1CS + resident witness -> proof, H(x) straight from the constraint values (gs_pinocchio_prove_witness): no px."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_witness(capi.Handle(dev_pk.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                              capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_witness_begin(dev_pk, dev_r1cs, w_handle):
    """Enqueue one witness -> proof (gs_pinocchio_prove_witness_begin) -> ticket for prove_end."""
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_pinocchio_prove_witness_begin(capi.Handle(dev_pk.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                                    ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def _u64_rows(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    return capi.ints_to_u64([v % R for v in x])


def prove_host_begin(dev_pk, w, px):
    """snark.GenerateProofs' own call shape, pipelined (gs_pinocchio_prove_host_begin): w and px in host memory -> ticket for prove_end."""
    wa, pa = _u64_rows(w), _u64_rows(px)
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_pinocchio_prove_host_begin(capi.Handle(dev_pk.h), capi.ptr64(wa), wa.shape[0], capi.ptr64(pa), pa.shape[0],
                                                                 ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_witness_host_begin(dev_pk, dev_r1cs, w):
    """A fresh host witness against the resident sparse R1CS (gs_pinocchio_prove_witness_host_begin) -> ticket for prove_end."""
    wa = _u64_rows(w)
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_pinocchio_prove_witness_host_begin(capi.Handle(dev_pk.h), capi.Handle(dev_r1cs.handle.h), capi.ptr64(wa),
                                                                         wa.shape[0], ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_from_witness_host(dev_pk, dev_r1cs, w):
    """Blocking: host witness -> proof (gs_pinocchio_prove_witness_host)."""
    wa = _u64_rows(w)
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_witness_host(capi.Handle(dev_pk.h), capi.Handle(dev_r1cs.handle.h), capi.ptr64(wa), wa.shape[0],
                                                                   capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def SetEvalBasis(dev_pk, points):
    """Attach an evaluation-basis copy of G1T (n Jacobian int triples) to a resident key: gs_pinocchio_pk_set_eval."""
    arr = capi.ints_to_u64([c for p in points for c in p]).reshape(-1, 12)
    b = capi.g1_upload(arr)
    capi.check(capi.load_library().gs_pinocchio_pk_set_eval(capi.Handle(dev_pk.h), capi.Handle(b.h)))


def prove_begin(dev_pk, w_handle, px_handle):
    """Enqueue one Pinocchio proof (gs_pinocchio_prove_begin) -> ticket.  Up to three operations may be outstanding."""
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_pinocchio_prove_begin(capi.Handle(dev_pk.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                            ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_end(ticket):
    """Collect the proof of a ticket (gs_pinocchio_prove_end)."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_end(ticket, capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


# ---- several GPUs (SURVEY 8e applied to snark.go:254-289): a proof is the sum of the ranks' eight partial points ----------------
def ShardPk(dev_pk, shard_index, shard_count, target_device=None):
    """Slice `shard_index` of `shard_count` of a resident full key (gs_pinocchio_pk_shard), on logical device `target_device` if given
    (gs_pinocchio_pk_shard_to).  -> DevicePk holding 1 / shard_count of every array."""
    h = capi.Handle(0)
    if target_device is None:
        capi.check(capi.load_library().gs_pinocchio_pk_shard(capi.Handle(dev_pk.h), shard_index, shard_count, ctypes.byref(h)))
    else:
        capi.check(capi.load_library().gs_pinocchio_pk_shard_to(capi.Handle(dev_pk.h), shard_index, shard_count, int(target_device), ctypes.byref(h)))
    return DevicePk(capi.DeviceHandle(h.value), dev_pk.nvars, dev_pk.npublic)


def _sums(out, inf):
    return np.array(out, dtype=np.uint64), [int(x) for x in inf]


def prove_partials(dev_pk, w_handle, px_handle, shard_index, shard_count):
    """The eight sums over shard `shard_index` of the term ranges (gs_pinocchio_prove_partials) -> (72 words, 8 infinity flags),
    the layout of a proof; add the ranks' records with combine()."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_partials(capi.Handle(dev_pk.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                               shard_index, shard_count, capi.ptr64(out), inf))
    return _sums(out, inf)


def witness_values(dev_pk, dev_r1cs, w_handle, hv_handle=None):
    """The proof owner's polynomial stage (gs_pinocchio_witness_values) -> (handle of the n values H(n+1..2n), violated)."""
    h = capi.Handle(hv_handle.h if hv_handle is not None else 0)
    bad = ctypes.c_uint32(0)
    capi.check(capi.load_library().gs_pinocchio_witness_values(capi.Handle(dev_pk.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                               ctypes.byref(h), ctypes.byref(bad)))
    return (hv_handle if hv_handle is not None else capi.DeviceHandle(h.value)), int(bad.value)


def prove_partials_values(dev_pk, w_handle, hv_slice, shard_index, shard_count):
    """gs_pinocchio_prove_partials_values: the eight sums with PiH over this rank's slice of H's values."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_prove_partials_values(capi.Handle(dev_pk.h), capi.Handle(w_handle.h), capi.Handle(hv_slice.h),
                                                                      shard_index, shard_count, capi.ptr64(out), inf))
    return _sums(out, inf)


def combine(records):
    """gs_pinocchio_combine: [(72 words, 8 flags)] of every rank -> Proof."""
    n = len(records)
    sums = np.ascontiguousarray(np.concatenate([np.asarray(r[0], dtype=np.uint64).reshape(72) for r in records]))
    fl = (ctypes.c_int * (8 * n))(*[int(x) for r in records for x in r[1]])
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    capi.check(capi.load_library().gs_pinocchio_combine(capi.ptr64(sums), fl, n, capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_multi(dev_pks, w_handles, third_handles, values=False):
    """One proof over len(dev_pks) logical devices of THIS process (gs_pinocchio_prove_multi, or _multi_values when `third_handles`
    are the devices' slices of H's values instead of replicas of px).  -> (Proof, used_rccl)."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    used = ctypes.c_int(0)
    lib = capi.load_library()
    fn = lib.gs_pinocchio_prove_multi_values if values else lib.gs_pinocchio_prove_multi
    capi.check(fn(capi._harr([k.handle for k in dev_pks]), capi._harr(w_handles), capi._harr(third_handles), len(dev_pks), capi.ptr64(out), inf,
                  ctypes.byref(used)))
    return _proof_from_words(out, inf), bool(used.value)


def prove_sharded_rccl(dev_pk, w_handle, third_handle, values=False):
    """One process per GPU, records gathered INSIDE the library over the communicator of capi.comm_init_rank
    (gs_pinocchio_prove_sharded / _sharded_values).  Every rank returns the same Proof."""
    out = np.zeros(72, dtype=np.uint64)
    inf = (ctypes.c_int * 8)()
    lib = capi.load_library()
    fn = lib.gs_pinocchio_prove_sharded_values if values else lib.gs_pinocchio_prove_sharded
    capi.check(fn(capi.Handle(dev_pk.h), capi.Handle(w_handle.h), capi.Handle(third_handle.h), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_batch(pk_of_device, w_handles, px_handles):
    """A batch of independent proofs round-robined over logical devices (gs_pinocchio_prove_batch): proof i runs where w_handles[i]
    lives, with pk_of_device[that device] (None for unused devices).  No collective."""
    n = len(w_handles)
    out = np.zeros((max(n, 1), 72), dtype=np.uint64)
    inf = (ctypes.c_int * (8 * max(n, 1)))()
    pks = capi._harr([(k.handle if k is not None else 0) for k in pk_of_device])
    capi.check(capi.load_library().gs_pinocchio_prove_batch(pks, len(pk_of_device), capi._harr(w_handles), capi._harr(px_handles), n,
                                                            capi.ptr64(out), inf))
    return [_proof_from_words(out[i], inf[8 * i:8 * i + 8]) for i in range(n)]


class Vk:
    """snark.Vk (snark.go:28-38): affine Jacobian tuples."""
    FIELDS = ("Vka", "Vkb", "Vkc", "G1Kbg", "G2Kbg", "G2Kg", "Vkz")

    def __init__(self, IC, **kw):
        self.IC = IC
        for k in self.FIELDS:
            setattr(self, k, kw[k])


PK_ARRAYS = {"A": 0, "Ap": 1, "B": 2, "Bp": 3, "C": 4, "Cp": 5, "Kp": 6, "G1T": 7, "G1TEval": 9}   # G1TEval: evaluation-basis copy of G1T


class DevicePk:
    def __init__(self, handle, nvars, npublic):
        self.h, self.handle, self.nvars, self.npublic = handle.h, handle, nvars, npublic


def GenerateTrustedSetupSparse(n, nvars, npublic, a_csr, b_csr, c_csr, toxic):
    """snark.GenerateTrustedSetup (snark.go:98-251) on a sparse R1CS, toxic = (T, Ka, Kb, Kc, Kbeta, Kgamma, RhoA, RhoB)
    injected instead of drawn at :114-148; runs on the device (gs_pinocchio_setup).  -> (DevicePk, Vk)."""
    capi.init()
    args = []
    for rp, cl, vl in (a_csr, b_csr, c_csr):
        rp = np.ascontiguousarray(rp, dtype=np.uint32)
        cl = np.ascontiguousarray(cl, dtype=np.uint32)
        vl = np.ascontiguousarray(vl, dtype=np.uint64).reshape(-1, 4)
        if cl.size == 0:
            cl, vl = np.zeros(1, dtype=np.uint32), np.zeros((1, 4), dtype=np.uint64)
        args += [rp, cl, vl]
    tox = capi.ints_to_u64([t % R for t in toxic]).reshape(-1)
    vk = np.zeros(4 * (72 + 6 * (npublic + 1)), dtype=np.uint64)          # 288 u32 words + IC, as u64 limbs
    h = capi.Handle(0)
    capi.check(capi.load_library().gs_pinocchio_setup(
        n, nvars, npublic, capi.ptr32(args[0]), capi.ptr32(args[1]), capi.ptr64(args[2]), capi.ptr32(args[3]), capi.ptr32(args[4]),
        capi.ptr64(args[5]), capi.ptr32(args[6]), capi.ptr32(args[7]), capi.ptr64(args[8]), capi.ptr64(tox), ctypes.byref(h), capi.ptr64(vk)))
    v = capi.u64_to_ints(vk)
    g1 = lambda o: (v[o], v[o + 1], v[o + 2])                                           # noqa: E731
    g2 = lambda o: ((v[o], v[o + 1]), (v[o + 2], v[o + 3]), (v[o + 4], v[o + 5]))       # noqa: E731
    vkey = Vk(IC=[g1(36 + 3 * i) for i in range(npublic + 1)], Vka=g2(0), Vkb=g1(6), Vkc=g2(9), G1Kbg=g1(15), G2Kbg=g2(18), G2Kg=g2(24),
              Vkz=g2(30))
    return DevicePk(capi.DeviceHandle(h.value), nvars, npublic), vkey


def ExportPkArray(dev_pk, name):
    which = PK_ARRAYS[name]
    count = capi.pk_eval_count(dev_pk.handle) if which == 9 else dev_pk.nvars - 1 if which == 7 else dev_pk.nvars
    if count == 0:
        return []
    words = 24 if which == 2 else 12
    out = np.zeros((count, words), dtype=np.uint64)
    capi.check(capi.load_library().gs_pinocchio_pk_export(capi.Handle(dev_pk.h), which, capi.ptr64(out), count))
    v = capi.u64_to_ints(out)
    if which == 2:
        return [((v[6 * i], v[6 * i + 1]), (v[6 * i + 2], v[6 * i + 3]), (v[6 * i + 4], v[6 * i + 5])) for i in range(count)]
    return [(v[3 * i], v[3 * i + 1], v[3 * i + 2]) for i in range(count)]


_CHECKS = ("e(piA, Va) == e(piA', g2), valid knowledge commitment for A",
           "e(Vb, piB) == e(piB', g2), valid knowledge commitment for B",
           "e(piC, Vc) == e(piC', g2), valid knowledge commitment for C",
           "e(Vkx+piA, piB) == e(piH, Vkz) * e(piC, g2), QAP disibility checked",
           "e(Vkx+piA+piC, g2KbetaKgamma) * e(g1KbetaKgamma, piB) == e(piK, g2Kgamma)")


def VerifyProof(vk, proof, publicSignals, debug=False):
    """snark.VerifyProof(vk, proof, publicSignals, debug) (snark.go:292-368) -> bool: the five pairing equations in the
    reference's order (gs_pinocchio_verify, host side, no device needed)."""
    if len(vk.IC) < len(publicSignals) + 1:
        raise IndexError("index out of range: %d public signals, vk.IC has %d points" % (len(publicSignals), len(vk.IC)))
    ic = capi.g1_points_to_u64(vk.IC)
    pub = capi.ints_to_u64([int(x) % R for x in publicSignals]) if publicSignals else np.zeros((1, 4), dtype=np.uint64)
    g1 = capi.g1_points_to_u64([vk.Vkb, vk.G1Kbg])
    g2 = capi.g2_points_to_u64([vk.Vka, vk.Vkc, vk.G2Kbg, vk.G2Kg, vk.Vkz])
    words = np.concatenate([capi.g1_points_to_u64([proof.PiA, proof.PiAp]).reshape(-1), capi.g2_points_to_u64([proof.PiB]).reshape(-1),
                            capi.g1_points_to_u64([proof.PiBp, proof.PiC, proof.PiCp, proof.PiH, proof.PiKp]).reshape(-1)])
    words = np.ascontiguousarray(words, dtype=np.uint64)
    ok, bad = ctypes.c_int(0), ctypes.c_int(0)
    capi.check(capi.load_library().gs_pinocchio_verify(capi.ptr64(g2[0]), capi.ptr64(g1[0]), capi.ptr64(g2[1]), capi.ptr64(g1[1]),
                                                       capi.ptr64(g2[2]), capi.ptr64(g2[3]), capi.ptr64(g2[4]), capi.ptr64(ic), len(vk.IC),
                                                       capi.ptr64(pub), len(publicSignals), capi.ptr64(words), ctypes.byref(ok),
                                                       ctypes.byref(bad)))
    if debug:
        for i, text in enumerate(_CHECKS):
            if bad.value and i + 1 == bad.value:
                print("❌ " + text)
                break
            print("✓ " + text)
    return bool(ok.value)
