"""ctypes binding of include/gosnark_hip.h (the same C ABI a cgo binding would use)."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LOCK = threading.Lock()
_INIT_DEVICE = None

u64p = ctypes.POINTER(ctypes.c_uint64)
u32p = ctypes.POINTER(ctypes.c_uint32)
intp = ctypes.POINTER(ctypes.c_int)
Handle = ctypes.c_uint64


class GosnarkHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgosnark_hip: status %d: %s" % (code, msg))
        self.code = code


class Timing(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_float) for n in
                 ("total_ms", "plan_ms", "accumulate_ms", "reduce_ms", "poly_ms", "h2d_ms", "acc_g1_ms", "acc_g2_ms")]
                + [("acc_g1_launches", ctypes.c_uint32), ("acc_g2_launches", ctypes.c_uint32),
                   ("acc_g1_terms", ctypes.c_uint64), ("acc_g2_terms", ctypes.c_uint64),
                   ("acc_g1_adds", ctypes.c_uint64), ("acc_g2_adds", ctypes.c_uint64),
                   ("window_bits", ctypes.c_uint32), ("fallbacks", ctypes.c_uint32),
                   ("plan_digits", ctypes.c_uint64), ("plan_entries", ctypes.c_uint64), ("heavy_buckets", ctypes.c_uint32), ("reserved", ctypes.c_uint32)])


class Memory(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("device_total_bytes", "device_free_bytes", "library_bytes", "object_bytes", "table_bytes",
                                               "workspace_bytes", "objects", "evictions")]


def lib_path():
    # GS_LIB: development aid for A/B runs of two builds inside one GPU session
    return os.environ.get("GS_LIB") or os.path.join(_HERE, "libgosnark_hip.so")


# name -> (argtypes); every function returns int status unless listed in _NONSTATUS
_SIGS = {
    "gs_init": [intp, ctypes.c_int],
    "gs_free": [Handle],
    "gs_len": [Handle, ctypes.POINTER(ctypes.c_size_t)],
    "gs_g1_upload": [u64p, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_g2_upload": [u64p, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_g1_download": [Handle, u64p, ctypes.c_size_t],
    "gs_g2_download": [Handle, u64p, ctypes.c_size_t],
    "gs_g1_fixed_base": [u64p, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_g2_fixed_base": [u64p, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_scalars_upload": [u64p, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_scalars_download": [Handle, u64p, ctypes.c_size_t],
    "gs_scalars_update": [Handle, u64p, ctypes.c_size_t],
    "gs_msm_g1": [Handle, u64p, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_msm_g2": [Handle, u64p, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_msm_g1_resident": [Handle, ctypes.c_size_t, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_msm_g2_resident": [Handle, ctypes.c_size_t, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_msm_g1_begin": [Handle, ctypes.c_size_t, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p],
    "gs_msm_g2_begin": [Handle, ctypes.c_size_t, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p],
    "gs_msm_end": [ctypes.c_uint64, u64p, intp],
    "gs_g1_sum_affine": [u64p, intp, ctypes.c_size_t, u64p, intp],
    "gs_g2_sum_affine": [u64p, intp, ctypes.c_size_t, u64p, intp],
    "gs_poly_mul": [u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p],
    "gs_poly_div": [u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, u64p],
    "gs_poly_add": [u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p],
    "gs_poly_sub": [u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p],
    "gs_poly_eval": [u64p, ctypes.c_size_t, u64p, u64p],
    "gs_lagrange_interpolation": [u64p, ctypes.c_size_t, u64p],
    "gs_zpoly": [ctypes.c_size_t, u64p],
    "gs_r1cs_to_px": [ctypes.c_size_t, ctypes.c_size_t, u32p, u32p, u64p, u32p, u32p, u64p, u32p, u32p, u64p,
                      u64p, u64p, u64p, u64p, u64p],
    "gs_r1cs_upload": [ctypes.c_size_t, ctypes.c_size_t, u32p, u32p, u64p, u32p, u32p, u64p, u32p, u32p, u64p, ctypes.POINTER(Handle)],
    "gs_r1cs_px": [Handle, Handle, ctypes.POINTER(Handle)],
    "gs_groth16_pk_create": [Handle, Handle, Handle, Handle, Handle, u64p, u64p, u64p, u64p, u64p,
                             u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_groth16_prove": [Handle, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, u64p, u64p, intp],
    "gs_groth16_prove_resident": [Handle, Handle, Handle, u64p, u64p, u64p, intp],
    "gs_groth16_prove_r1cs": [Handle, Handle, Handle, ctypes.POINTER(Handle), u64p, u64p, u64p, intp],
    "gs_groth16_prove_witness": [Handle, Handle, Handle, u64p, u64p, u64p, intp],
    "gs_groth16_prove_witness_begin": [Handle, Handle, Handle, u64p, u64p, u64p],
    "gs_groth16_pk_set_eval": [Handle, Handle],
    "gs_pk_eval_count": [Handle, ctypes.POINTER(ctypes.c_size_t)],
    "gs_pinocchio_pk_set_eval": [Handle, Handle],
    "gs_pinocchio_prove_witness_begin": [Handle, Handle, Handle, u64p],
    "gs_groth16_prove_begin": [Handle, Handle, Handle, u64p, u64p, u64p],
    "gs_groth16_prove_end": [ctypes.c_uint64, u64p, intp],
    "gs_groth16_prove_host_begin": [Handle, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, u64p, u64p],
    "gs_groth16_prove_witness_host_begin": [Handle, Handle, u64p, ctypes.c_size_t, u64p, u64p, u64p],
    "gs_groth16_prove_witness_host": [Handle, Handle, u64p, ctypes.c_size_t, u64p, u64p, u64p, intp],
    "gs_pinocchio_prove_host_begin": [Handle, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p],
    "gs_pinocchio_prove_witness_host_begin": [Handle, Handle, u64p, ctypes.c_size_t, u64p],
    "gs_pinocchio_prove_witness_host": [Handle, Handle, u64p, ctypes.c_size_t, u64p, intp],
    "gs_ticket_cancel": [ctypes.c_uint64],
    "gs_groth16_pk_create_shard": [Handle, Handle, Handle, Handle, Handle, u64p, u64p, u64p, u64p, u64p, u64p, ctypes.c_size_t,
                                   ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_groth16_pk_shard": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_groth16_prove_partials": [Handle, Handle, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_groth16_finish": [Handle, u64p, intp, u64p, u64p, u64p, intp],
    "gs_groth16_setup": [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, u32p, u32p, u64p, u32p, u32p, u64p, u32p, u32p, u64p,
                         u64p, ctypes.POINTER(Handle), u64p],
    "gs_groth16_pk_export": [Handle, ctypes.c_int, u64p, ctypes.c_size_t],
    "gs_pinocchio_setup": [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, u32p, u32p, u64p, u32p, u32p, u64p, u32p, u32p, u64p,
                           u64p, ctypes.POINTER(Handle), u64p],
    "gs_pinocchio_pk_export": [Handle, ctypes.c_int, u64p, ctypes.c_size_t],
    "gs_pinocchio_pk_create": [Handle, Handle, Handle, Handle, Handle, Handle, Handle, Handle, u64p,
                               ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_pinocchio_prove": [Handle, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, intp],
    "gs_pinocchio_prove_resident": [Handle, Handle, Handle, u64p, intp],
    "gs_pinocchio_prove_witness": [Handle, Handle, Handle, u64p, intp],
    "gs_pinocchio_prove_begin": [Handle, Handle, Handle, u64p],
    "gs_pinocchio_prove_end": [ctypes.c_uint64, u64p, intp],
    "gs_device_count": [],
    "gs_set_device": [ctypes.c_int],
    "gs_get_device": [],
    "gs_handle_device": [Handle],
    "gs_scalars_clone": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_g1_clone": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_g2_clone": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_groth16_pk_shard_to": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_comm_unique_id": [ctypes.POINTER(ctypes.c_uint8)],
    "gs_comm_init_rank": [ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.c_int],
    "gs_comm_init_local": [],
    "gs_comm_info": [intp, intp, intp, u64p],
    "gs_comm_allgather": [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p],
    "gs_msm_g1_multi": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, intp, intp],
    "gs_msm_g2_multi": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, intp, intp],
    "gs_groth16_prove_multi": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, u64p, u64p, intp, intp],
    "gs_groth16_prove_sharded": [Handle, Handle, Handle, u64p, u64p, u64p, intp],
    "gs_groth16_witness_values": [Handle, Handle, Handle, ctypes.POINTER(Handle), ctypes.POINTER(ctypes.c_uint32)],
    "gs_groth16_prove_partials_values": [Handle, Handle, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_groth16_partials_values_begin": [Handle, Handle, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p],
    "gs_groth16_partials_end": [ctypes.c_uint64, u64p, intp],
    "gs_groth16_prove_multi_values": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, u64p, u64p, intp, intp],
    "gs_groth16_prove_sharded_values": [Handle, Handle, Handle, u64p, u64p, u64p, intp],
    "gs_scalars_scatter": [Handle, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_msm_g1_sharded": [Handle, Handle, u64p, intp],
    "gs_msm_g2_sharded": [Handle, Handle, u64p, intp],
    "gs_groth16_prove_batch": [ctypes.POINTER(Handle), ctypes.c_int, ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_size_t,
                               u64p, u64p, u64p, intp],
    "gs_pinocchio_pk_shard": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(Handle)],
    "gs_pinocchio_pk_shard_to": [Handle, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(Handle)],
    "gs_pinocchio_prove_partials": [Handle, Handle, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_pinocchio_witness_values": [Handle, Handle, Handle, ctypes.POINTER(Handle), ctypes.POINTER(ctypes.c_uint32)],
    "gs_pinocchio_prove_partials_values": [Handle, Handle, Handle, ctypes.c_size_t, ctypes.c_size_t, u64p, intp],
    "gs_pinocchio_combine": [u64p, intp, ctypes.c_size_t, u64p, intp],
    "gs_pinocchio_prove_multi": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, intp, intp],
    "gs_pinocchio_prove_multi_values": [ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_int, u64p, intp, intp],
    "gs_pinocchio_prove_sharded": [Handle, Handle, Handle, u64p, intp],
    "gs_pinocchio_prove_sharded_values": [Handle, Handle, Handle, u64p, intp],
    "gs_pinocchio_prove_batch": [ctypes.POINTER(Handle), ctypes.c_int, ctypes.POINTER(Handle), ctypes.POINTER(Handle), ctypes.c_size_t, u64p, intp],
    "gs_last_timing": [ctypes.POINTER(Timing)],
    "gs_device_timing": [ctypes.c_int, ctypes.POINTER(Timing)],
    "gs_set_window_bits": [ctypes.c_int],
    "gs_set_eval_basis": [ctypes.c_int],
    "gs_memory_query": [ctypes.POINTER(Memory)],
    "gs_handle_bytes": [Handle, u64p, u64p],
    "gs_release_tables": [Handle],
    "gs_set_table_policy": [ctypes.c_int],
    "gs_build_tables": [Handle, ctypes.c_int],
    "gs_set_memory_limit": [ctypes.c_uint64],
    "gs_alloc_counters": [u64p, u64p],
    "gs_abi_sizes": [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)],
    "gs_trim": [],
    "gs_verify_set_strict": [ctypes.c_int],
    "gs_pairing": [u64p, u64p, u64p],
    "gs_pairing_check": [u64p, u64p, ctypes.c_size_t, intp],
    "gs_groth16_verify": [u64p, u64p, u64p, u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, u64p, u64p, intp],
    "gs_pinocchio_verify": [u64p, u64p, u64p, u64p, u64p, u64p, u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p, intp, intp],
}
EXPORTS = sorted(list(_SIGS) + ["gs_shutdown", "gs_last_error", "gs_version", "gs_comm_destroy"])


def load_library():
    """dlopen libgosnark_hip.so (built by `make -C go-snark-study_amd/csrc` / __graft_entry__.build()).
    Fails loudly when it is missing: there is no other implementation to fall back to."""
    global _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        path = lib_path()
        try:
            # torch ships its own libamdhip64/libhsa-runtime64 (same sonames as /opt/rocm's): when a process
            # uses both torch and this library (bench.py, smoke()), torch's copies must be the ones that get
            # loaded first, otherwise two HSA runtimes end up in the process and device discovery fails.
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(path):
            raise GosnarkHipError(-1, "%s not found: build it with __graft_entry__.build() "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        lib = ctypes.CDLL(path)
        missing = [n for n in EXPORTS if not hasattr(lib, n)]
        if missing:
            raise GosnarkHipError(-1, "%s lacks symbols declared in include/gosnark_hip.h: %s" % (path, missing))
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        lib.gs_last_error.restype = ctypes.c_char_p
        lib.gs_version.restype = ctypes.c_char_p
        lib.gs_shutdown.restype = None
        lib.gs_comm_destroy.restype = None
        # gs_timing / gs_memory are written through our pointers: a library built from another revision of the header must not
        # be handed these structs (gs_timing grew by 24 bytes in round 4 without anything noticing)
        tb, mb = ctypes.c_size_t(0), ctypes.c_size_t(0)
        lib.gs_abi_sizes(ctypes.byref(tb), ctypes.byref(mb))
        if tb.value != ctypes.sizeof(Timing) or mb.value != ctypes.sizeof(Memory):
            raise GosnarkHipError(-1, "%s writes gs_timing / gs_memory of %d / %d bytes, this binding expects %d / %d: rebuild the library"
                                  % (path, tb.value, mb.value, ctypes.sizeof(Timing), ctypes.sizeof(Memory)))
        _LIB = lib
        return lib


def check(status):
    if status != 0:
        raise GosnarkHipError(status, load_library().gs_last_error().decode("utf-8", "replace"))


def init(device=None):
    """gs_init on `device` (default: LOCAL_RANK or 0): one process drives one GPU.  `device` may also be a list of HIP
    ordinals -- one logical device (context) per entry, the same ordinal may repeat (see include/gosnark_hip.h)."""
    global _INIT_DEVICE
    lib = load_library()
    if device is None:
        if _INIT_DEVICE is not None:          # already initialised: keep it
            return
        device = int(os.environ.get("LOCAL_RANK", "0"))
    devices = tuple(int(d) for d in device) if isinstance(device, (list, tuple)) else (int(device),)
    if _INIT_DEVICE == devices:
        return
    arr = (ctypes.c_int * len(devices))(*devices)
    check(lib.gs_init(arr, len(devices)))
    _INIT_DEVICE = devices


def device_count():
    return load_library().gs_device_count()


def set_device(logical):
    """Objects created by this host thread from now on live on logical device `logical`."""
    check(load_library().gs_set_device(int(logical)))


def get_device():
    return load_library().gs_get_device()


def handle_device(handle):
    return load_library().gs_handle_device(Handle(handle.h))


def shutdown():
    """gs_shutdown: every handle dies, streams / pinned buffers / workspaces are returned; init() may be called again."""
    global _INIT_DEVICE
    load_library().gs_shutdown()
    _INIT_DEVICE = None


def version():
    return load_library().gs_version().decode()


# ---- integer <-> limb array helpers -------------------------------------------------------------
def ints_to_u64(vals, words=4):
    """list of non-negative Python ints -> np.uint64 array [len, words] (little-endian limbs)."""
    nbytes = 8 * words
    buf = b"".join(int(v).to_bytes(nbytes, "little") for v in vals)
    return np.frombuffer(buf, dtype="<u8").reshape(len(vals), words).copy()


def u64_to_ints(arr, words=4):
    a = np.ascontiguousarray(arr, dtype="<u8").reshape(-1, words)
    raw = a.tobytes()
    nbytes = 8 * words
    return [int.from_bytes(raw[i * nbytes:(i + 1) * nbytes], "little") for i in range(a.shape[0])]


def ptr64(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def ptr32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def g1_points_to_u64(points):
    """[(X, Y, Z), ...] Jacobian ints -> [n, 12] uint64"""
    flat = [c for p in points for c in p]
    return ints_to_u64(flat).reshape(len(points), 12)


def g2_points_to_u64(points):
    """[((X0,X1),(Y0,Y1),(Z0,Z1)), ...] -> [n, 24] uint64"""
    flat = [c for p in points for xy in p for c in xy]
    return ints_to_u64(flat).reshape(len(points), 24)


class DeviceHandle:
    """RAII wrapper of a gs_handle."""

    def __init__(self, h):
        self.h = int(h)

    def free(self):
        """gs_free.  Safe while tickets that read the object are outstanding (the library defers the release); if the call
        fails all the same (foreign handle) the handle is kept so that the failure is not silent."""
        if self.h:
            lib = load_library()
            rc = lib.gs_free(Handle(self.h))
            if rc == 0 or lib.gs_device_count() == 0:     # after gs_shutdown every handle is already gone
                self.h = 0
            else:
                check(rc)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self):
        n = ctypes.c_size_t(0)
        check(load_library().gs_len(Handle(self.h), ctypes.byref(n)))
        return n.value


def _upload(fname, arr, n):
    init()
    h = Handle(0)
    check(getattr(load_library(), fname)(ptr64(arr), n, ctypes.byref(h)))
    return DeviceHandle(h.value)


def pk_eval_count(pk_handle):
    """gs_pk_eval_count: evaluation-basis points a resident Groth16 / Pinocchio key holds (0 = none)."""
    n = ctypes.c_size_t(0)
    check(load_library().gs_pk_eval_count(Handle(_raw(pk_handle)), ctypes.byref(n)))
    return int(n.value)


def g1_upload(points_u64):
    a = np.ascontiguousarray(points_u64, dtype=np.uint64).reshape(-1, 12)
    return _upload("gs_g1_upload", a, a.shape[0])


def g2_upload(points_u64):
    a = np.ascontiguousarray(points_u64, dtype=np.uint64).reshape(-1, 24)
    return _upload("gs_g2_upload", a, a.shape[0])


def scalars_upload(s_u64):
    a = np.ascontiguousarray(s_u64, dtype=np.uint64).reshape(-1, 4)
    return _upload("gs_scalars_upload", a, a.shape[0])


def g1_fixed_base(s_u64):
    a = np.ascontiguousarray(s_u64, dtype=np.uint64).reshape(-1, 4)
    return _upload("gs_g1_fixed_base", a, a.shape[0])


def g2_fixed_base(s_u64):
    a = np.ascontiguousarray(s_u64, dtype=np.uint64).reshape(-1, 4)
    return _upload("gs_g2_fixed_base", a, a.shape[0])


def g1_download(handle):
    n = len(handle)
    out = np.zeros((n, 12), dtype=np.uint64)
    check(load_library().gs_g1_download(Handle(handle.h), ptr64(out), n))
    return out


def g2_download(handle):
    n = len(handle)
    out = np.zeros((n, 24), dtype=np.uint64)
    check(load_library().gs_g2_download(Handle(handle.h), ptr64(out), n))
    return out


def scalars_download(handle):
    n = len(handle)
    out = np.zeros((n, 4), dtype=np.uint64)
    check(load_library().gs_scalars_download(Handle(handle.h), ptr64(out), n))
    return out


def _affine_result(out, inf, g2):
    if inf.value:
        return None
    v = u64_to_ints(out)
    return ((v[0], v[1]), (v[2], v[3])) if g2 else (v[0], v[1])


def msm(bases, scalars_u64, off=0, g2=False):
    """sum_i scalars[i] * bases[off+i] -> affine (x, y) / ((x0,x1),(y0,y1)) ints, None = infinity."""
    s = np.ascontiguousarray(scalars_u64, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    fn = load_library().gs_msm_g2 if g2 else load_library().gs_msm_g1
    check(fn(Handle(bases.h), ptr64(s), off, s.shape[0], ptr64(out), ctypes.byref(inf)))
    return _affine_result(out, inf, g2)


def msm_resident(bases, scalars, n, off=0, soff=0, g2=False):
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    fn = load_library().gs_msm_g2_resident if g2 else load_library().gs_msm_g1_resident
    check(fn(Handle(bases.h), off, Handle(scalars.h), soff, n, ptr64(out), ctypes.byref(inf)))
    return _affine_result(out, inf, g2)


def sum_affine(points, g2=False):
    """points: list of affine tuples or None (infinity) -> affine sum."""
    n = len(points)
    words = 4 if g2 else 2
    flat, infs = [], []
    for p in points:
        if p is None:
            flat += [0] * words
            infs.append(1)
        else:
            flat += ([p[0][0], p[0][1], p[1][0], p[1][1]] if g2 else [p[0], p[1]])
            infs.append(0)
    arr = ints_to_u64(flat).reshape(n, 4 * words) if n else np.zeros((0, 4 * words), dtype=np.uint64)
    ia = (ctypes.c_int * max(n, 1))(*infs)
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    fn = load_library().gs_g2_sum_affine if g2 else load_library().gs_g1_sum_affine
    check(fn(ptr64(arr), ia, n, ptr64(out), ctypes.byref(inf)))
    return _affine_result(out, inf, g2)


def last_timing():
    t = Timing()
    check(load_library().gs_last_timing(ctypes.byref(t)))
    return {n: getattr(t, n) for n, _ in Timing._fields_}


def _raw(h):
    return int(h.h) if hasattr(h, "h") else int(h)


def memory_query():
    """gs_memory_query: what the current logical device's GPU and this library hold (bytes)."""
    init()
    m = Memory()
    check(load_library().gs_memory_query(ctypes.byref(m)))
    return {n: int(getattr(m, n)) for n, _ in Memory._fields_ if n != "reserved"}


def handle_bytes(h):
    """gs_handle_bytes -> (object bytes, window-table bytes) of one handle."""
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load_library().gs_handle_bytes(Handle(_raw(h)), ctypes.cast(ctypes.byref(a), u64p), ctypes.cast(ctypes.byref(b), u64p)))
    return int(a.value), int(b.value)


def release_tables(h):
    """gs_release_tables: drop the window tables of a key / base array (rebuilt on its next use)."""
    check(load_library().gs_release_tables(Handle(_raw(h))))


TABLE_POLICY = {"auto": 0, "always": 1, "never": 2}


def set_table_policy(policy):
    """gs_set_table_policy: "auto" (table-free until a base array's second use, then a build in instalments paid by the calls
    that follow), "always" (build inside the first call), "never" (table-free only).  Results never depend on it."""
    init()
    check(load_library().gs_set_table_policy(TABLE_POLICY[policy] if isinstance(policy, str) else int(policy)))


def build_tables(h, route=0):
    """gs_build_tables: build the window tables of a key / base array now (blocking).  route: 0 all, 1 px routes only, 2 witness
    routes only."""
    check(load_library().gs_build_tables(Handle(_raw(h)), int(route)))


def set_memory_limit(nbytes):
    """gs_set_memory_limit (development / test hook): cap on the device bytes the library may hold, 0 = none."""
    check(load_library().gs_set_memory_limit(ctypes.c_uint64(int(nbytes))))


def alloc_counters():
    """gs_alloc_counters -> (hipMalloc calls, hipFree calls) the library has made so far."""
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load_library().gs_alloc_counters(ctypes.cast(ctypes.byref(a), u64p), ctypes.cast(ctypes.byref(b), u64p)))
    return int(a.value), int(b.value)


def scalars_update(handle, s_u64):
    """gs_scalars_update: overwrite a resident scalar vector in place (same length) -- no allocation, no device-wide sync."""
    a = np.ascontiguousarray(s_u64, dtype=np.uint64).reshape(-1, 4)
    check(load_library().gs_scalars_update(Handle(_raw(handle)), ptr64(a), a.shape[0]))


def trim():
    """gs_trim: drop the cached workspaces of the current logical device."""
    check(load_library().gs_trim())


def set_eval_basis(on):
    init()
    check(load_library().gs_set_eval_basis(1 if on else 0))


def set_window_bits(c):
    init()
    check(load_library().gs_set_window_bits(int(c)))


def zpoly(deg):
    """Z(x) = prod_{i=1}^{deg} (x - i) as a [deg+1, 4] uint64 array (gs_zpoly)."""
    init()
    out = np.zeros((deg + 1, 4), dtype=np.uint64)
    check(load_library().gs_zpoly(deg, ptr64(out)))
    return out


def msm_begin(bases, scalars, n, off=0, soff=0, g2=False):
    """Enqueue one resident MSM (gs_msm_g1_begin / gs_msm_g2_begin) -> ticket; at most three operations outstanding per logical device."""
    t = ctypes.c_uint64(0)
    fn = load_library().gs_msm_g2_begin if g2 else load_library().gs_msm_g1_begin
    check(fn(Handle(bases.h), off, Handle(scalars.h), soff, n, ctypes.cast(ctypes.byref(t), u64p)))
    return (t.value, g2)


def ticket_cancel(ticket):
    """gs_ticket_cancel: abandon a pipelined operation (proof ticket, or the integer of an MSM ticket) without its result."""
    check(load_library().gs_ticket_cancel(ctypes.c_uint64(ticket[0] if isinstance(ticket, tuple) else ticket)))


def msm_end(ticket):
    t, g2 = ticket
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    check(load_library().gs_msm_end(ctypes.c_uint64(t), ptr64(out), ctypes.byref(inf)))
    return _affine_result(out, inf, g2)


# ---- several logical devices / communicator (multi.hip) -------------------------------------------------
def _clone(fname, handle, off, n, target):
    h = Handle(0)
    check(getattr(load_library(), fname)(Handle(handle.h), off, n, int(target), ctypes.byref(h)))
    return DeviceHandle(h.value)


def scalars_clone(handle, target, off=0, n=None):
    return _clone("gs_scalars_clone", handle, off, len(handle) - off if n is None else n, target)


def g1_clone(handle, target, off=0, n=None):
    return _clone("gs_g1_clone", handle, off, len(handle) - off if n is None else n, target)


def g2_clone(handle, target, off=0, n=None):
    return _clone("gs_g2_clone", handle, off, len(handle) - off if n is None else n, target)


def scalars_scatter(full_handle, total, root, slice_handle=None):
    """gs_scalars_scatter (one process per GPU, communicator of comm_init_rank): the root's `total` scalars -> every rank's slice of
    the contiguous split.  full_handle is ignored on the other ranks (pass None)."""
    h = Handle(slice_handle.h if slice_handle is not None else 0)
    check(load_library().gs_scalars_scatter(Handle(full_handle.h if full_handle is not None else 0), int(total), int(root), ctypes.byref(h)))
    return slice_handle if slice_handle is not None else DeviceHandle(h.value)


def comm_unique_id():
    buf = (ctypes.c_uint8 * 128)()
    check(load_library().gs_comm_unique_id(buf))
    return bytes(buf)


def comm_init_rank(uid, nranks, rank):
    buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
    check(load_library().gs_comm_init_rank(buf, int(nranks), int(rank)))


def comm_init_local():
    check(load_library().gs_comm_init_local())


def comm_destroy():
    load_library().gs_comm_destroy()


def comm_info():
    nr, rk, loc, cnt = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_uint64(0)
    check(load_library().gs_comm_info(ctypes.byref(nr), ctypes.byref(rk), ctypes.byref(loc), ctypes.cast(ctypes.byref(cnt), u64p)))
    return {"nranks": nr.value, "rank": rk.value, "local": bool(loc.value), "collectives": cnt.value}


def comm_allgather(block, nblocks_out, local_blocks=1):
    """bytes `block` (local_blocks consecutive blocks in local mode) -> bytes of all ranks' blocks."""
    per = len(block) // local_blocks
    send = (ctypes.c_uint8 * len(block)).from_buffer_copy(block)
    recv = (ctypes.c_uint8 * (per * nblocks_out))()
    check(load_library().gs_comm_allgather(send, per, recv))
    return bytes(recv)


def _harr(handles):
    return (Handle * len(handles))(*[Handle(h.h if isinstance(h, DeviceHandle) else int(h)) for h in handles])


def msm_multi(bases, scalars, g2=False):
    """One MSM over len(bases) logical devices (gs_msm_g1_multi / gs_msm_g2_multi) -> (affine point, used_rccl)."""
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf, used = ctypes.c_int(0), ctypes.c_int(0)
    fn = load_library().gs_msm_g2_multi if g2 else load_library().gs_msm_g1_multi
    check(fn(_harr(bases), _harr(scalars), len(bases), ptr64(out), ctypes.byref(inf), ctypes.byref(used)))
    return _affine_result(out, inf, g2), bool(used.value)


def msm_sharded(bases, scalars, g2=False):
    out = np.zeros(16 if g2 else 8, dtype=np.uint64)
    inf = ctypes.c_int(0)
    fn = load_library().gs_msm_g2_sharded if g2 else load_library().gs_msm_g1_sharded
    check(fn(Handle(bases.h), Handle(scalars.h), ptr64(out), ctypes.byref(inf)))
    return _affine_result(out, inf, g2)


def device_timing(logical):
    t = Timing()
    check(load_library().gs_device_timing(int(logical), ctypes.byref(t)))
    return {n: getattr(t, n) for n, _ in Timing._fields_}
