"""Mirror of the reference's groth16 prover interface (groth16/groth16.go) on the HIP library.

    pk = groth16.Pk(...)                      # same fields as groth16.go:15-32
    proof = groth16.GenerateProofs(circuit, pk, w, px)            # groth16.go:225
    proof = groth16.GenerateProofsWithRS(circuit, pk, w, px, r, s)   # randomness injected

Values are Python ints / tuples shaped like the reference's big.Int structures.  Proof elements
are returned in the affine normal form [x, y, 1] (infinity = all zero)."""
import os

import numpy as np

from . import capi

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class Circuit:
    """The fields of circuitcompiler.Circuit the prover reads (circuit.go:12-26; groth16.go:243,248)."""

    def __init__(self, NVars, NPublic):
        self.NVars = NVars
        self.NPublic = NPublic


class Pk:
    """groth16.Pk (groth16.go:15-32)."""

    def __init__(self, BACDelta, Z, G1_Alpha, G1_Beta, G1_Delta, G1_At, G1_BACGamma,
                 G2_Beta, G2_Delta, G2_BACGamma, PowersTauDelta, G2_Gamma=None):
        self.BACDelta, self.Z = BACDelta, Z
        self.G1_Alpha, self.G1_Beta, self.G1_Delta = G1_Alpha, G1_Beta, G1_Delta
        self.G1_At, self.G1_BACGamma = G1_At, G1_BACGamma
        self.G2_Beta, self.G2_Gamma, self.G2_Delta = G2_Beta, G2_Gamma, G2_Delta
        self.G2_BACGamma = G2_BACGamma
        self.PowersTauDelta = PowersTauDelta
        self._dev = None


class Proof:
    """groth16.Proof (groth16.go:61-65)."""

    def __init__(self, PiA, PiB, PiC):
        self.PiA, self.PiB, self.PiC = PiA, PiB, PiC


class DevicePk:
    """Proving key resident in HBM (upload + affine-normalise once per circuit, SURVEY hard part 4)."""

    def __init__(self, handle, nvars, npublic, keep):
        self.handle, self.nvars, self.npublic, self._keep = handle, nvars, npublic, keep


def device_pk_from_handles(at, bacgamma1, bacgamma2, bacdelta, ptd, alpha, beta, delta, beta2, delta2, z_u64, nvars, npublic):
    """Assemble a DevicePk from already-resident base arrays (capi.DeviceHandle) and Jacobian int tuples."""
    import ctypes
    capi.init()
    h = capi.Handle(0)
    a = capi.g1_points_to_u64([alpha, beta, delta])
    b = capi.g2_points_to_u64([beta2, delta2])
    z = np.ascontiguousarray(z_u64, dtype=np.uint64).reshape(-1, 4)
    capi.check(capi.load_library().gs_groth16_pk_create(
        capi.Handle(at.h), capi.Handle(bacgamma1.h), capi.Handle(bacgamma2.h), capi.Handle(bacdelta.h), capi.Handle(ptd.h),
        capi.ptr64(a[0]), capi.ptr64(a[1]), capi.ptr64(a[2]), capi.ptr64(b[0]), capi.ptr64(b[1]),
        capi.ptr64(z), z.shape[0], nvars, npublic, ctypes.byref(h)))
    return DevicePk(capi.DeviceHandle(h.value), nvars, npublic, None)


class Vk:
    """groth16.Vk (groth16.go:33-43): affine Jacobian tuples."""

    def __init__(self, IC, G1_Alpha, G2_Beta, G2_Gamma, G2_Delta):
        self.IC, self.G1_Alpha, self.G2_Beta, self.G2_Gamma, self.G2_Delta = IC, G1_Alpha, G2_Beta, G2_Gamma, G2_Delta


def GenerateTrustedSetupSparse(n, nvars, npublic, a_csr, b_csr, c_csr, toxic):
    """groth16.GenerateTrustedSetup (groth16.go:94-222) on a sparse R1CS with the toxic scalars
    (T, Kalpha, Kbeta, Kgamma, Kdelta) injected instead of drawn at :99-119.  Everything heavy runs on the device
    (gs_groth16_setup); returns (DevicePk resident in HBM, Vk)."""
    import ctypes
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
    vk = np.zeros(12 + 72 + 12 * (npublic + 1), dtype=np.uint64)
    h = capi.Handle(0)
    capi.check(capi.load_library().gs_groth16_setup(
        n, nvars, npublic, capi.ptr32(args[0]), capi.ptr32(args[1]), capi.ptr64(args[2]), capi.ptr32(args[3]), capi.ptr32(args[4]),
        capi.ptr64(args[5]), capi.ptr32(args[6]), capi.ptr32(args[7]), capi.ptr64(args[8]), capi.ptr64(tox), ctypes.byref(h), capi.ptr64(vk)))
    v = capi.u64_to_ints(vk)
    g1 = lambda o: (v[o], v[o + 1], v[o + 2])                               # noqa: E731
    g2 = lambda o: ((v[o], v[o + 1]), (v[o + 2], v[o + 3]), (v[o + 4], v[o + 5]))   # noqa: E731
    vkey = Vk(IC=[g1(21 + 3 * i) for i in range(npublic + 1)], G1_Alpha=g1(0), G2_Beta=g2(3), G2_Gamma=g2(9), G2_Delta=g2(15))
    return DevicePk(capi.DeviceHandle(h.value), nvars, npublic, None), vkey


# "PowersTauDeltaEval": the evaluation-basis copy of PowersTauDelta (include/gosnark_hip.h, gs_groth16_pk_set_eval) -- not a field
# of the reference's Pk; keys built by gs_groth16_setup carry it, the binary key container stores it as an extra section.
PK_ARRAYS = {"G1_At": 0, "G1_BACGamma": 1, "G2_BACGamma": 2, "BACDelta": 3, "PowersTauDelta": 4, "PowersTauDeltaEval": 7}


def ExportPkArray(dev_pk, name):
    """One array of a resident key as affine Jacobian int tuples (testing / serialisation)."""
    which = PK_ARRAYS[name]
    count = capi.pk_eval_count(dev_pk.handle) if which == 7 else dev_pk.nvars if which != 4 else dev_pk.nvars - 1
    if count == 0:
        return []
    words = 24 if which == 2 else 12
    out = np.zeros((count, words), dtype=np.uint64)
    capi.check(capi.load_library().gs_groth16_pk_export(capi.Handle(dev_pk.handle.h), which, capi.ptr64(out), count))
    v = capi.u64_to_ints(out)
    if which == 2:
        return [((v[6 * i], v[6 * i + 1]), (v[6 * i + 2], v[6 * i + 3]), (v[6 * i + 4], v[6 * i + 5])) for i in range(count)]
    return [(v[3 * i], v[3 * i + 1], v[3 * i + 2]) for i in range(count)]


def UploadPk(pk, circuit):
    """Additive extension: make pk resident.  Cached on the Pk object."""
    if pk._dev is not None:
        return pk._dev
    at = capi.g1_upload(capi.g1_points_to_u64(pk.G1_At))
    b1 = capi.g1_upload(capi.g1_points_to_u64(pk.G1_BACGamma))
    b2 = capi.g2_upload(capi.g2_points_to_u64(pk.G2_BACGamma))
    cd = capi.g1_upload(capi.g1_points_to_u64(pk.BACDelta))
    pt = capi.g1_upload(capi.g1_points_to_u64(pk.PowersTauDelta))
    pk._dev = device_pk_from_handles(at, b1, b2, cd, pt, pk.G1_Alpha, pk.G1_Beta, pk.G1_Delta, pk.G2_Beta, pk.G2_Delta,
                                     capi.ints_to_u64([z % R for z in pk.Z]), circuit.NVars, circuit.NPublic)
    return pk._dev


def _shard_range(n, count, index):
    q, rem = divmod(n, count)
    lo = index * q + min(index, rem)
    return lo, lo + q + (1 if index < rem else 0)


def ShardPk(dev_pk, shard_index, shard_count):
    """The slice of a resident full key that rank `shard_index` of `shard_count` needs for prove_partials / prove_sharded
    (gs_groth16_pk_shard).  Free the full key afterwards (dev_pk.handle.free()) to keep only 1/shard_count of it in HBM."""
    import ctypes
    h = capi.Handle(0)
    capi.check(capi.load_library().gs_groth16_pk_shard(capi.Handle(dev_pk.handle.h), shard_index, shard_count, ctypes.byref(h)))
    return DevicePk(capi.DeviceHandle(h.value), dev_pk.nvars, dev_pk.npublic, None)


def ShardPkTo(dev_pk, shard_index, shard_count, target_device):
    """The same slice, created on logical device `target_device` (gs_groth16_pk_shard_to; the copies cross xGMI when the two
    are different GPUs)."""
    import ctypes
    h = capi.Handle(0)
    capi.check(capi.load_library().gs_groth16_pk_shard_to(capi.Handle(dev_pk.handle.h), shard_index, shard_count, int(target_device),
                                                          ctypes.byref(h)))
    return DevicePk(capi.DeviceHandle(h.value), dev_pk.nvars, dev_pk.npublic, None)


def UploadPkShard(pk, circuit, shard_index, shard_count):
    """Upload ONLY this rank's slice of a host key (gs_groth16_pk_create_shard): arrays cut with the split prove_partials uses."""
    import ctypes
    capi.init()
    wlo, whi = _shard_range(circuit.NVars, shard_count, shard_index)
    hlo, hhi = _shard_range(len(pk.PowersTauDelta), shard_count, shard_index)
    empty1, empty2 = np.zeros((0, 12), dtype=np.uint64), np.zeros((0, 24), dtype=np.uint64)
    up1 = lambda pts: capi.g1_upload(capi.g1_points_to_u64(pts) if pts else empty1)     # noqa: E731
    at, b1, cd = up1(pk.G1_At[wlo:whi]), up1(pk.G1_BACGamma[wlo:whi]), up1(pk.BACDelta[wlo:whi])
    pt = up1(pk.PowersTauDelta[hlo:hhi])
    sl2 = pk.G2_BACGamma[wlo:whi]
    b2 = capi.g2_upload(capi.g2_points_to_u64(sl2) if sl2 else empty2)
    return device_pk_shard_from_handles(at, b1, b2, cd, pt, pk.G1_Alpha, pk.G1_Beta, pk.G1_Delta, pk.G2_Beta, pk.G2_Delta,
                                        capi.ints_to_u64([z % R for z in pk.Z]), circuit.NVars, circuit.NPublic, len(pk.PowersTauDelta),
                                        shard_index, shard_count)


def device_pk_shard_from_handles(at, bacgamma1, bacgamma2, bacdelta, ptd, alpha, beta, delta, beta2, delta2, z_u64, nvars, npublic,
                                 nptd_total, shard_index, shard_count):
    import ctypes
    h = capi.Handle(0)
    a = capi.g1_points_to_u64([alpha, beta, delta])
    b = capi.g2_points_to_u64([beta2, delta2])
    z = np.ascontiguousarray(z_u64, dtype=np.uint64).reshape(-1, 4)
    capi.check(capi.load_library().gs_groth16_pk_create_shard(
        capi.Handle(at.h), capi.Handle(bacgamma1.h), capi.Handle(bacgamma2.h), capi.Handle(bacdelta.h), capi.Handle(ptd.h),
        capi.ptr64(a[0]), capi.ptr64(a[1]), capi.ptr64(a[2]), capi.ptr64(b[0]), capi.ptr64(b[1]),
        capi.ptr64(z), z.shape[0], nvars, npublic, nptd_total, shard_index, shard_count, ctypes.byref(h)))
    return DevicePk(capi.DeviceHandle(h.value), nvars, npublic, None)


def _proof_from_words(out, inf):
    v = capi.u64_to_ints(out)
    PiA = (0, 0, 0) if inf[0] else (v[0], v[1], 1)
    PiB = ((0, 0), (0, 0), (0, 0)) if inf[1] else ((v[2], v[3]), (v[4], v[5]), (1, 0))
    PiC = (0, 0, 0) if inf[2] else (v[6], v[7], 1)
    return Proof(PiA, PiB, PiC)


def FqRRand():
    """Utils.FqR.Rand (fields/fq.go:116-132): 30 random bytes, big-endian, mod r."""
    return int.from_bytes(os.urandom(30), "big") % R


GS_ERR_BUSY = -6


def _host_scalars(x, what):
    """The reference's []*big.Int (Python ints: reduced mod r here, negatives rejected -- the reference drops the sign, fq.go:138-140) or an
    [n, 4] uint64 limb array (taken as it is: the device reduces any value < 2^256) -> contiguous [n, 4] uint64."""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    if any(v < 0 for v in x):
        raise ValueError("negative %s values are not supported (the reference drops the sign, fq.go:138-140)" % what)
    return capi.ints_to_u64([v % R for v in x])


def GenerateProofsWithRS(circuit, pk, w, px, r, s):
    """groth16.go:225-278 with r, s given instead of drawn at :231-238.  Round 6 (as go/groth16hip.GenerateProofsWithRS): w and px travel
    as a HOST-BUFFER TICKET collected at once (gs_groth16_prove_host_begin + gs_groth16_prove_end: staged into the slot's own device
    buffers, nothing allocated per proof, concurrent callers pipeline); when all three slots are taken, the blocking entry point."""
    import ctypes
    dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
    wa, pa = _host_scalars(w, "witness"), _host_scalars(px, "px")
    try:
        return prove_end(prove_host_begin(dev, wa, pa, r, s))
    except capi.GosnarkHipError as e:
        if e.code != GS_ERR_BUSY:
            raise
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove(capi.Handle(dev.handle.h), capi.ptr64(wa), wa.shape[0], capi.ptr64(pa), pa.shape[0],
                                                    capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def GenerateProofs(circuit, pk, w, px):
    """groth16.GenerateProofs(circuit, pk, w, px) (groth16.go:225)."""
    return GenerateProofsWithRS(circuit, pk, w, px, FqRRand(), FqRRand())


def GenerateProofsFromWitnessWithRS(circuit, pk, dev_r1cs, w, r, s):
    """go/groth16hip.GenerateProofsFromWitnessWithRS: the callers' R1CSToQAP -> CombinePolynomials -> GenerateProofs chain (cli/main.go:480-501)
    from the witness alone, against the circuit's resident sparse R1CS (r1csqap.DeviceR1CS); a host-buffer ticket collected at once."""
    dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
    wa = _host_scalars(w, "witness")
    try:
        return prove_end(prove_witness_host_begin(dev, dev_r1cs, wa, r, s))
    except capi.GosnarkHipError as e:
        if e.code != GS_ERR_BUSY:
            raise
    return prove_from_witness_host(dev, dev_r1cs, wa, r, s)


class Prover:
    """The streaming drop-in (go/groth16hip.Prover, tests/c/stream_producer.c): one resident key, a NEW witness per Submit, up to three
    proofs in flight, proofs back in submission order.
        p = groth16.NewProver(circuit, pk, dev_r1cs)          # dev_r1cs = None: every Submit brings px
        for w in witnesses:
            p.Submit(w)                                       # or p.Submit(w, px)
            if p.InFlight() == 3: proof = p.Collect()
        while p.InFlight(): proof = p.Collect()
    A Submit on a full pipeline first collects the oldest ticket into a done-queue (it never fails with GS_ERR_BUSY)."""
    MaxInFlight = 3

    def __init__(self, circuit, pk, dev_r1cs=None):
        self.dev = pk if isinstance(pk, DevicePk) else UploadPk(pk, circuit)
        self.r1cs = dev_r1cs
        self.tickets, self.done = [], []

    def _collect_oldest(self):
        self.done.append(prove_end(self.tickets.pop(0)))

    def SubmitWithRS(self, w, px, r, s):
        if px is None and self.r1cs is None:
            raise ValueError("this prover has no resident R1CS: Submit needs px")
        wa = _host_scalars(w, "witness")
        pa = None if px is None else _host_scalars(px, "px")
        while True:
            if len(self.tickets) >= self.MaxInFlight:
                self._collect_oldest()
            try:
                t = prove_witness_host_begin(self.dev, self.r1cs, wa, r, s) if pa is None else prove_host_begin(self.dev, wa, pa, r, s)
            except capi.GosnarkHipError as e:
                if e.code == GS_ERR_BUSY and self.tickets:       # another prover shares the device's slots: make room and retry
                    self._collect_oldest()
                    continue
                raise
            self.tickets.append(t)
            return

    def Submit(self, w, px=None):
        self.SubmitWithRS(w, px, FqRRand(), FqRRand())

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


def prove_resident(dev_pk, w_handle, px_handle, r, s):
    """Inputs already resident in HBM (what bench.py times)."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove_resident(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                             capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_from_r1cs(dev_pk, dev_r1cs, w_handle, r, s, px_handle=None):
    """Sparse R1CS + resident witness -> proof in one call (gs_groth16_prove_r1cs): px is computed behind the accumulations
    over w.  Returns (Proof, px_handle); pass the previous px_handle to overwrite it instead of allocating."""
    import ctypes
    rs = capi.ints_to_u64([r % R, s % R])
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    h = capi.Handle(px_handle.h if px_handle is not None else 0)
    capi.check(capi.load_library().gs_groth16_prove_r1cs(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                         ctypes.byref(h), capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf), (px_handle if px_handle is not None else capi.DeviceHandle(h.value))


def prove_from_witness(dev_pk, dev_r1cs, w_handle, r, s):
    """Sparse R1CS + resident witness -> proof, H(x) straight from the constraint values (gs_groth16_prove_witness): no px."""
    import ctypes
    rs = capi.ints_to_u64([r % R, s % R])
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    capi.check(capi.load_library().gs_groth16_prove_witness(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                            capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_witness_begin(dev_pk, dev_r1cs, w_handle, r, s):
    """Enqueue one witness -> proof (gs_groth16_prove_witness_begin) -> ticket for prove_end.  With an evaluation-basis key the
    call never waits for the device."""
    import ctypes
    rs = capi.ints_to_u64([r % R, s % R])
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_groth16_prove_witness_begin(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                                  capi.ptr64(rs[0]), capi.ptr64(rs[1]), ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def _u64_rows(x):
    """ints or an [n, 4] uint64 array -> contiguous [n, 4] uint64 (standard form)"""
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
    return capi.ints_to_u64([v % R for v in x])


def prove_host_begin(dev_pk, w, px, r, s):
    """groth16.GenerateProofs' own call shape at the pipelined rate (gs_groth16_prove_host_begin): w and px in HOST memory (ints or
    [n, 4] uint64 arrays), new ones every call, staged into the ticket slot's own device buffers -> ticket for prove_end."""
    import ctypes
    wa, pa = _u64_rows(w), _u64_rows(px)
    rs = capi.ints_to_u64([r % R, s % R])
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_groth16_prove_host_begin(capi.Handle(dev_pk.handle.h), capi.ptr64(wa), wa.shape[0], capi.ptr64(pa), pa.shape[0],
                                                               capi.ptr64(rs[0]), capi.ptr64(rs[1]), ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_witness_host_begin(dev_pk, dev_r1cs, w, r, s):
    """A fresh witness in HOST memory against the resident sparse R1CS (gs_groth16_prove_witness_host_begin) -> ticket for prove_end."""
    import ctypes
    wa = _u64_rows(w)
    rs = capi.ints_to_u64([r % R, s % R])
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_groth16_prove_witness_host_begin(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.ptr64(wa),
                                                                       wa.shape[0], capi.ptr64(rs[0]), capi.ptr64(rs[1]),
                                                                       ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_from_witness_host(dev_pk, dev_r1cs, w, r, s):
    """Blocking: host witness -> proof (gs_groth16_prove_witness_host)."""
    import ctypes
    wa = _u64_rows(w)
    rs = capi.ints_to_u64([r % R, s % R])
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    capi.check(capi.load_library().gs_groth16_prove_witness_host(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.ptr64(wa), wa.shape[0],
                                                                 capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def SetEvalBasis(dev_pk, points):
    """Attach an evaluation-basis copy of PowersTauDelta (n Jacobian int triples, e.g. read from a key file) to a resident key:
    gs_groth16_pk_set_eval.  The witness route then runs its h-MSM over H's values (no interpolation)."""
    arr = capi.ints_to_u64([c for p in points for c in p]).reshape(-1, 12)
    b = capi.g1_upload(arr)
    capi.check(capi.load_library().gs_groth16_pk_set_eval(capi.Handle(dev_pk.handle.h), capi.Handle(b.h)))


def prove_partials(dev_pk, w_handle, px_handle, shard_index, shard_count):
    """This rank's five raw MSM sums (gs_groth16_prove_partials): [At, G1.BACGamma, G2.BACGamma, BACDelta, h.PTD] as affine
    points / None, plus the g2 flags parallel.allgather_points wants."""
    import ctypes
    out = np.zeros(48, dtype=np.uint64)
    inf = (ctypes.c_int * 5)()
    capi.check(capi.load_library().gs_groth16_prove_partials(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                             shard_index, shard_count, capi.ptr64(out), inf))
    v = capi.u64_to_ints(out)
    pts = [None if inf[0] else (v[0], v[1]), None if inf[1] else (v[2], v[3]),
           None if inf[2] else ((v[4], v[5]), (v[6], v[7])), None if inf[3] else (v[8], v[9]), None if inf[4] else (v[10], v[11])]
    return pts, SUM_IS_G2


SUM_IS_G2 = [False, False, True, False, False]


def finish(dev_pk, sums, r, s):
    """gs_groth16_finish: the O(1) tail of groth16.go:253-275 on the (combined) five sums."""
    import ctypes
    flat, infs = [], []
    for p, g2 in zip(sums, SUM_IS_G2):
        words = 4 if g2 else 2
        if p is None:
            flat += [0] * words
            infs.append(1)
        else:
            flat += ([p[0][0], p[0][1], p[1][0], p[1][1]] if g2 else [p[0], p[1]])
            infs.append(0)
    arr = capi.ints_to_u64(flat).reshape(-1)
    ia = (ctypes.c_int * 5)(*infs)
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_finish(capi.Handle(dev_pk.handle.h), capi.ptr64(arr), ia, capi.ptr64(rs[0]), capi.ptr64(rs[1]),
                                                     capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_sharded(dev_pk, w_handle, px_handle, r, s, group=None):
    """One proof over all ranks of `group` (one process per GPU): local sums over this rank's term ranges -> ONE all-gather of
    the 5 partial points per rank -> local combination -> tail.  Every rank returns the same Proof."""
    import torch.distributed as dist
    from . import parallel
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    pts, flags = prove_partials(dev_pk, w_handle, px_handle, rank, world)
    per_rank = parallel.allgather_points(pts, flags, group)
    return finish(dev_pk, parallel.combine_partials(per_rank, flags), r, s)


def prove_multi(dev_pks, w_handles, px_handles, r, s):
    """One proof over len(dev_pks) logical devices of THIS process (gs_groth16_prove_multi): dev_pks[d] is the full key or
    slice d on logical device d, w_handles[d] / px_handles[d] replicas there.  Returns (Proof, used_rccl)."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    used = ctypes.c_int(0)
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove_multi(capi._harr([k.handle for k in dev_pks]), capi._harr(w_handles), capi._harr(px_handles),
                                                          len(dev_pks), capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf,
                                                          ctypes.byref(used)))
    return _proof_from_words(out, inf), bool(used.value)


def witness_values(dev_pk, dev_r1cs, w_handle, hv_handle=None):
    """The proof owner's polynomial stage (gs_groth16_witness_values): resident sparse R1CS + witness -> the n values H(n+1..2n) as a
    resident scalar vector.  Returns (hv_handle, violated); violated != 0 means the witness breaks a constraint and the values are void."""
    import ctypes
    h = capi.Handle(hv_handle.h if hv_handle is not None else 0)
    bad = ctypes.c_uint32(0)
    capi.check(capi.load_library().gs_groth16_witness_values(capi.Handle(dev_pk.handle.h), capi.Handle(dev_r1cs.handle.h), capi.Handle(w_handle.h),
                                                             ctypes.byref(h), ctypes.byref(bad)))
    return (hv_handle if hv_handle is not None else capi.DeviceHandle(h.value)), int(bad.value)


def prove_partials_values(dev_pk, w_handle, hv_slice, shard_index, shard_count):
    """gs_groth16_prove_partials_values: this rank's five sums, the fifth over its slice of H's values (no polynomial work here)."""
    import ctypes
    out = np.zeros(48, dtype=np.uint64)
    inf = (ctypes.c_int * 5)()
    capi.check(capi.load_library().gs_groth16_prove_partials_values(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(hv_slice.h),
                                                                    shard_index, shard_count, capi.ptr64(out), inf))
    v = capi.u64_to_ints(out)
    pts = [None if inf[0] else (v[0], v[1]), None if inf[1] else (v[2], v[3]),
           None if inf[2] else ((v[4], v[5]), (v[6], v[7])), None if inf[3] else (v[8], v[9]), None if inf[4] else (v[10], v[11])]
    return pts, SUM_IS_G2


def partials_values_begin(dev_pk, w_handle, hv_slice, shard_index, shard_count):
    """gs_groth16_partials_values_begin -> ticket (collect with partials_end)."""
    import ctypes
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_groth16_partials_values_begin(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(hv_slice.h),
                                                                    shard_index, shard_count, ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def partials_end(ticket):
    """gs_groth16_partials_end -> the five sums as prove_partials returns them."""
    import ctypes
    out = np.zeros(48, dtype=np.uint64)
    inf = (ctypes.c_int * 5)()
    capi.check(capi.load_library().gs_groth16_partials_end(ctypes.c_uint64(ticket), capi.ptr64(out), inf))
    v = capi.u64_to_ints(out)
    return [None if inf[0] else (v[0], v[1]), None if inf[1] else (v[2], v[3]),
            None if inf[2] else ((v[4], v[5]), (v[6], v[7])), None if inf[3] else (v[8], v[9]), None if inf[4] else (v[10], v[11])]


def scatter_values(hv_handle, ndev):
    """The owner's scatter between the logical devices of this process: slice d of the contiguous split of H's values -> device d."""
    n = len(hv_handle)
    out = []
    for d in range(ndev):
        lo, hi = _shard_range(n, ndev, d)
        out.append(capi.scalars_clone(hv_handle, d, lo, hi - lo))
    return out


def prove_multi_values(dev_pks, w_handles, hv_slices, r, s):
    """One proof over the logical devices of this process with the polynomial stage done ONCE (gs_groth16_prove_multi_values):
    hv_slices[d] = device d's slice of H's values (scatter_values).  Returns (Proof, used_rccl)."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    used = ctypes.c_int(0)
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove_multi_values(capi._harr([k.handle for k in dev_pks]), capi._harr(w_handles), capi._harr(hv_slices),
                                                                 len(dev_pks), capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf,
                                                                 ctypes.byref(used)))
    return _proof_from_words(out, inf), bool(used.value)


def prove_sharded_values_rccl(dev_pk, w_handle, hv_slice, r, s):
    """One process per GPU, values route (gs_groth16_prove_sharded_values): this rank's slice of H's values came from the owner
    through capi.scalars_scatter; the 416-byte records are gathered inside the library."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove_sharded_values(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(hv_slice.h),
                                                                   capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_sharded_rccl(dev_pk, w_handle, px_handle, r, s):
    """One process per GPU, gathered INSIDE the library over the communicator of capi.comm_init_rank
    (gs_groth16_prove_sharded).  Every rank returns the same Proof."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    rs = capi.ints_to_u64([r % R, s % R])
    capi.check(capi.load_library().gs_groth16_prove_sharded(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                            capi.ptr64(rs[0]), capi.ptr64(rs[1]), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def prove_batch(pk_of_device, w_handles, px_handles, rs_pairs):
    """A batch of independent proofs round-robined over logical devices (gs_groth16_prove_batch, BASELINE configs[4]): proof i
    runs where w_handles[i] lives, with pk_of_device[that device] (None for unused devices).  No collective."""
    import ctypes
    n = len(w_handles)
    out = np.zeros((max(n, 1), 32), dtype=np.uint64)
    inf = (ctypes.c_int * (3 * max(n, 1)))()
    ra = capi.ints_to_u64([r % R for r, _ in rs_pairs]) if n else np.zeros((1, 4), dtype=np.uint64)
    sa = capi.ints_to_u64([s % R for _, s in rs_pairs]) if n else np.zeros((1, 4), dtype=np.uint64)
    pks = capi._harr([(k.handle if k is not None else 0) for k in pk_of_device])
    capi.check(capi.load_library().gs_groth16_prove_batch(pks, len(pk_of_device), capi._harr(w_handles), capi._harr(px_handles), n,
                                                          capi.ptr64(ra), capi.ptr64(sa), capi.ptr64(out), inf))
    return [_proof_from_words(out[i], inf[3 * i:3 * i + 3]) for i in range(n)]


def prove_begin(dev_pk, w_handle, px_handle, r, s):
    """Enqueue one proof (gs_groth16_prove_begin) -> ticket.  At most three may be outstanding."""
    import ctypes
    rs = capi.ints_to_u64([r % R, s % R])
    t = ctypes.c_uint64(0)
    capi.check(capi.load_library().gs_groth16_prove_begin(capi.Handle(dev_pk.handle.h), capi.Handle(w_handle.h), capi.Handle(px_handle.h),
                                                          capi.ptr64(rs[0]), capi.ptr64(rs[1]), ctypes.cast(ctypes.byref(t), capi.u64p)))
    return t.value


def prove_end(ticket):
    """Wait for that proof and return it (gs_groth16_prove_end)."""
    import ctypes
    out = np.zeros(32, dtype=np.uint64)
    inf = (ctypes.c_int * 3)()
    capi.check(capi.load_library().gs_groth16_prove_end(ctypes.c_uint64(ticket), capi.ptr64(out), inf))
    return _proof_from_words(out, inf)


def VerifyProof(vk, proof, publicSignals, debug=False):
    """groth16.VerifyProof(vk, proof, publicSignals, debug) (groth16.go:281-305) -> bool.  Host side
    (gs_groth16_verify: one 4-pair multi-pairing with a shared final exponentiation); needs no device."""
    import ctypes
    if len(vk.IC) < len(publicSignals) + 1:
        raise IndexError("index out of range: %d public signals, vk.IC has %d points" % (len(publicSignals), len(vk.IC)))
    ic = capi.g1_points_to_u64(vk.IC)
    pub = capi.ints_to_u64([int(x) % R for x in publicSignals]) if publicSignals else np.zeros((1, 4), dtype=np.uint64)
    g1 = capi.g1_points_to_u64([vk.G1_Alpha, proof.PiA, proof.PiC])
    g2 = capi.g2_points_to_u64([vk.G2_Beta, vk.G2_Gamma, vk.G2_Delta, proof.PiB])
    ok = ctypes.c_int(0)
    capi.check(capi.load_library().gs_groth16_verify(capi.ptr64(g1[0]), capi.ptr64(g2[0]), capi.ptr64(g2[1]), capi.ptr64(g2[2]),
                                                     capi.ptr64(ic), len(vk.IC), capi.ptr64(pub), len(publicSignals),
                                                     capi.ptr64(g1[1]), capi.ptr64(g2[3]), capi.ptr64(g1[2]), ctypes.byref(ok)))
    if debug:
        print("✓ groth16 verification passed" if ok.value else "❌ groth16 verification not passed")
    return bool(ok.value)
