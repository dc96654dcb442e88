"""ctypes binding of libfasta_hip.so (C ABI: include/fasta_hip.h).

This is the only module that talks to the GPU, and nothing in it falls back: if the shared library is missing or
no MI355X is visible, `load_library()` / `HipContext()` raise.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $FASTA_HIP_LIB names another build of the SAME C ABI (the test job of the experimental stencil forms points it at
# libfasta_hip_experimental.so, csrc/fh_experimental.h); it is never a fallback: a path that does not exist is an error
LIB_PATH = os.environ.get("FASTA_HIP_LIB") or os.path.join(_HERE, "libfasta_hip.so")

# enums mirrored from include/fasta_hip.h ---------------------------------------------------------
PROX_IDENTITY, PROX_SHRINK, PROX_NONNEG, PROX_LINF, PROX_L1BALL, PROX_TVBALL, PROX_BOX = range(7)
(VEC_X0, VEC_G0, VEC_XHAT, VEC_XPROX, VEC_X1, VEC_G1, VEC_BEST, VEC_B, VEC_Z,
 VEC_T0, VEC_T1, VEC_T2, VEC_T3) = range(13)
(S_FSQ, S_DXG0, S_DX2, S_XH2, S_G02, S_GSUM, S_GMAX, S_RDOT, S_DXDG, S_DG2, S_FSQ_ADJ, S_XH2_ADJ,
 S_GSUM_ADJ, S_GMAX_ADJ, S_ALPHA) = range(15)
NSCALARS = 16
K_FWD, K_ADJ, K_AUX, K_COMM, K_FUSED, K_HOST_ISSUE, K_LEVEL = range(7)
(TUNE_FWD_ROWS, TUNE_FWD_GRID_CAP, TUNE_ADJ_SLAB_ROWS, TUNE_ADJ_CPT, TUNE_LD_PAD, TUNE_NT_LOADS,
 TUNE_TV_U, TUNE_TV_ROWS, TUNE_TV_NT, TUNE_FUSED_VARIANT, TUNE_TV_ZFREE, TUNE_TV_PIPE, TUNE_TV_XCD, TUNE_TV_LDS_PAD,
 TUNE_TV_RING, TUNE_TV_SLOTS, TUNE_FUSED_CUS, TUNE_RUN_MAX_N, TUNE_SEQ_POLL, TUNE_RUN_CHAIN, TUNE_ADJ_CYCLIC) = range(21)
# keys 10, 13, 14, 15 (TUNE_TV_ZFREE / _LDS_PAD / _RING / _SLOTS) are NOT in include/fasta_hip.h: experimental forms of the stencil sweep,
# accepted only by libfasta_hip_experimental.so (csrc/fh_experimental.h); the shipped library answers FH_E_ARG
EXPERIMENTAL_KEYS = (TUNE_TV_ZFREE, TUNE_TV_LDS_PAD, TUNE_TV_RING, TUNE_TV_SLOTS)
TUNE_TEST_HOOKS = 0x7E57             # tests only (csrc/fh_experimental.h): 1 = a withheld team partial, 2 = the co-residency probe says no,
#                                      4 = the level search's last workgroup withholds its record, 8 = ... and its fall-back is off,
#                                      attempt << 8 = fh_run's last workgroup stays away from that attempt's first grid barrier
HOOK_WITHHOLD_PARTIAL, HOOK_PROBE_SAYS_NO, HOOK_LEVEL_WITHHOLD, HOOK_LEVEL_NO_FALLBACK = 1, 2, 4, 8
HOOK_RUN_ATTEMPT_SHIFT = 8
E_ARG, E_STATE, E_RCCL, E_TIMEOUT = 10001, 10002, 10003, 10004
LAUNCH_SEPARATE, LAUNCH_ONEPASS_ALWAYS, LAUNCH_ONEPASS_SPECULATIVE, LAUNCH_PAIR = range(4)     # fh_run_opts.launch_mode (fh_iterate)
LAUNCH_MODES = {None: LAUNCH_SEPARATE, "always": LAUNCH_ONEPASS_ALWAYS, "speculative": LAUNCH_ONEPASS_SPECULATIVE, "pair": LAUNCH_PAIR}
RECOVERED_LEVEL_FALLBACK, RECOVERED_LEVEL_FAILED, RECOVERED_RUN_TIMEOUT = range(3)
UNIQUE_ID_BYTES = 128
DTYPE_F64, DTYPE_F32_STORAGE = 0, 1
CREATE_RCCL_SHELL = 0x100          # or'ed into the dtype of fh_create_ex: the multi-device (RCCL) form even for a single device
STORAGE = {"f64": DTYPE_F64, "f32": DTYPE_F32_STORAGE}

_u64, _i32, _dbl = C.c_uint64, C.c_int, C.c_double
_ctx = C.c_void_p
_pd = C.POINTER(C.c_double)

RUN_HIST, RUN_WINDOW_MAX = 8, 64
STOP_RULES = ("residual", "norm_residual", "ratio_residual", "hybrid_residual")      # fh_run_opts.stop_rule = index (fasta/stopping.py:6-51)


class RunOpts(C.Structure):                 # fh_run_opts
    _fields_ = [(k, _i32) for k in ("adaptive", "accelerate", "backtrack", "restart", "evaluate_objective", "stop_rule", "window",
                                    "max_backtracks")] + [("stepsize_shrink", _dbl), ("tolerance", _dbl), ("launch_mode", _i32), ("reserved", _i32)]


class RunState(C.Structure):                # fh_run_state
    _fields_ = [("tau_next", _dbl), ("alpha1", _dbl), ("max_residual", _dbl), ("best_quality", _dbl), ("iteration", _u64), ("backtracks", _u64),
                ("stopped", _i32), ("reserved", _i32), ("f_window", _dbl * RUN_WINDOW_MAX),
                ("spec_cooldown", _i32), ("onepass_backoff", _i32), ("onepass_off_until", C.c_int64),
                ("onepass_launches", _u64), ("pair_launches", _u64), ("onepass_timeouts", _u64)]



# every exported symbol with its signature; tests assert the .so exports exactly these ------------
SIGNATURES = {
    "fh_last_error": (C.c_char_p, []),
    "fh_device_count": (_i32, [C.POINTER(_i32)]),
    "fh_create": (_i32, [_i32, C.POINTER(_ctx)]),
    "fh_create_ex": (_i32, [_i32, C.POINTER(_i32), _i32, C.POINTER(_ctx)]),
    "fh_shard_count": (_i32, [_ctx, C.POINTER(_i32)]),
    "fh_shard": (_i32, [_ctx, _i32, C.POINTER(_ctx), C.POINTER(_u64), C.POINTER(_u64)]),
    "fh_destroy": (_i32, [_ctx]),
    "fh_sync": (_i32, [_ctx]),
    "fh_set_tuning": (_i32, [_ctx, _i32, C.c_longlong]),
    "fh_set_matrix": (_i32, [_ctx, _pd, _u64, _u64, _u64]),
    "fh_set_matrix_f32": (_i32, [_ctx, C.POINTER(C.c_float), _u64, _u64, _u64]),
    "fh_generate_matrix": (_i32, [_ctx, _u64, _u64, _u64, _u64, _dbl]),
    "fh_get_matrix_rows": (_i32, [_ctx, _u64, _u64, _pd]),
    "fh_set_stencil": (_i32, [_ctx, _u64, _u64]),
    "fh_shape": (_i32, [_ctx, C.POINTER(_u64), C.POINTER(_u64)]),
    "fh_set_loss_lsq": (_i32, [_ctx, _pd, _u64]),
    "fh_set_loss_logistic": (_i32, [_ctx, _pd, _u64]),
    "fh_set_prox": (_i32, [_ctx, _i32, _dbl, _dbl, _dbl]),
    "fh_set_vector": (_i32, [_ctx, _i32, _pd, _u64]),
    "fh_get_vector": (_i32, [_ctx, _i32, _pd, _u64]),
    "fh_init": (_i32, [_ctx, _pd]),
    "fh_setup": (_i32, [_ctx, _pd]),
    "fh_gradient_at": (_i32, [_ctx, _i32, _i32]),
    "fh_diff_norm": (_i32, [_ctx, _i32, _i32, _pd]),
    "fh_fwd": (_i32, [_ctx, _dbl, _pd]),
    "fh_adj": (_i32, [_ctx, _dbl, _i32, _dbl, _pd]),
    "fh_commit": (_i32, [_ctx, _i32]),
    "fh_abi_sizes": (_i32, [C.POINTER(_u64)]),
    "fh_run_supported": (_i32, [_ctx, C.POINTER(_i32)]),
    "fh_run": (_i32, [_ctx, _i32, C.POINTER(RunOpts), C.POINTER(RunState), _pd, C.POINTER(_i32)]),
    "fh_iterate": (_i32, [_ctx, _i32, C.POINTER(RunOpts), C.POINTER(RunState), _pd, C.POINTER(_i32)]),
    "fh_recovered_count": (_i32, [_ctx, _i32, C.POINTER(_u64)]),
    "fh_fused_supported": (_i32, [_ctx, C.POINTER(_i32)]),
    "fh_fused_agree": (_i32, [_ctx, C.POINTER(_i32)]),
    "fh_coresident_probe": (_i32, [_ctx, _i32, C.POINTER(_i32)]),
    "fh_fused_shape": (_i32, [_u64, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "fh_fwd_adj": (_i32, [_ctx, _dbl, _pd]),
    "fh_step": (_i32, [_ctx, _dbl, _pd]),
    "fh_step_begin": (_i32, [_ctx, _dbl]),
    "fh_step_end": (_i32, [_ctx, _pd]),
    "fh_step_accel": (_i32, [_ctx, _dbl, _dbl, _i32, _pd]),
    "fh_apply": (_i32, [_ctx, _i32, _pd, _pd]),
    "fh_comm_unique_id": (_i32, [C.c_void_p]),
    "fh_comm_init": (_i32, [_ctx, _i32, _i32, C.c_void_p]),
    "fh_comm_count": (_i32, [_ctx, C.POINTER(_i32)]),
    "fh_comm_destroy": (_i32, [_ctx]),
    "fh_comm_library": (C.c_char_p, []),
    "fh_comm_version": (_i32, [C.POINTER(_i32)]),
    "fh_alloc_settle": (_i32, [_i32]),
    "fh_alloc_settle_waited": (_i32, [C.POINTER(_dbl)]),
    "fh_alloc_cache": (_i32, [_i32]),
    "fh_release_cached": (_i32, [_i32]),
    "fh_alloc_cache_hits": (_i32, [C.POINTER(_u64)]),
    "fh_comm_selftest": (_i32, [_ctx, _u64, _pd, C.POINTER(_i32)]),
    "fh_cu_count": (_i32, [_ctx, C.POINTER(_i32), C.POINTER(_i32)]),
    "fh_timing_enable": (_i32, [_ctx, _i32]),
    "fh_timing_get": (_i32, [_ctx, _i32, _pd, C.POINTER(_u64)]),
    "fh_timing_reset": (_i32, [_ctx]),
    "fh_timing_overlap": (_i32, [_ctx, _ctx, _i32, _pd, _pd, _pd]),
    "fh_stream_read_ms": (_i32, [_ctx, _i32, _pd, C.POINTER(_u64)]),
}

_lib = None


class HipError(RuntimeError):
    """Non-zero status from libfasta_hip.so (message = fh_last_error())."""


class HipTimeout(HipError):
    """A bounded in-launch hand-off ran out: the one-pass kernel's spins (scalar word 15 of the launch: the launch itself succeeded and
    left the solver state untouched, the caller may fall back to K-fwd / K-adj), or status FH_E_TIMEOUT -- the clipping-level search of the
    l-infinity prox / l1-ball projection found no level even alone (the step's outputs are NaN; a second failure on the two-launch path
    propagates out of fasta()).  The ONLY HipError a solver may recover from: every other non-zero status (device fault, RCCL error,
    bad state) must propagate.  `partial`: history records of the iterations an fh_iterate call completed before it failed."""
    partial = None


def load_library(path=None):
    """dlopen libfasta_hip.so and bind every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise HipError(f"{path} is missing -- build it with `python __graft_entry__.py` "
                       "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    # the structs that cross the boundary by pointer: this binding's layouts must be the library's (a stale .so next to a newer hip.py, or the
    # other way round, would otherwise read and write past each other's fields)
    sizes = (_u64 * 4)()
    _check(lib, lib.fh_abi_sizes(sizes))
    if tuple(sizes) != (C.sizeof(RunOpts), C.sizeof(RunState), RUN_HIST, RUN_WINDOW_MAX):
        raise HipError(f"{path}: fh_run_opts / fh_run_state / history layout {tuple(sizes)} does not match this binding's "
                       f"{(C.sizeof(RunOpts), C.sizeof(RunState), RUN_HIST, RUN_WINDOW_MAX)} -- rebuild the library (python __graft_entry__.py)")
    _lib = lib
    return lib


def _check(lib, status):
    if status != 0:
        raise (HipTimeout if status == E_TIMEOUT else HipError)(f"[{status}] " + lib.fh_last_error().decode(errors="replace"))


def _as_f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_pd)


def device_count():
    lib = load_library()
    n = _i32(0)
    _check(lib, lib.fh_device_count(C.byref(n)))
    return n.value


def device_cus(device=0):
    """Compute units device `device` reports (a throw-away context asks the runtime)."""
    with HipContext(int(device)) as c:
        return c.cu_count()[0]


def fused_shape(n, storage="f64", variant=2, ncu=256):
    """((pieces per lane, posting distance, team members, x slice in LDS, row buffers), instantiated) of the one-pass kernel for
    rows of n columns; all zeros = no one-pass kernel for that width.  Host-only (works without a GPU)."""
    lib = load_library()
    shape = (_i32 * 5)()
    inst = _i32(0)
    _check(lib, lib.fh_fused_shape(int(n), STORAGE[storage], int(variant), int(ncu), shape, C.byref(inst)))
    return tuple(shape), bool(inst.value)


def comm_library():
    """Path of the library the collectives come from ("" before the first communicator): the system's RCCL or $FASTA_RCCL_LIB."""
    return load_library().fh_comm_library().decode(errors="replace")


def comm_version():
    """RCCL's version code (e.g. 22203), or -1 before the first communicator / when the loaded library does not export it."""
    lib = load_library()
    v = _i32(-1)
    _check(lib, lib.fh_comm_version(C.byref(v)))
    return int(v.value)


def alloc_settle(enable=True):
    """Process-wide: wait with a large (>= 1 GiB) matrix allocation that no kept block serves until the device's earlier large frees have
    been cleared by the driver (default OFF since the one-pass kernel deals its rows cyclically and no longer depends on the mapping; see
    include/fasta_hip.h, profiles/r06_alloc_settle.txt and profiles/r06_placement.txt)."""
    lib = load_library()
    _check(lib, lib.fh_alloc_settle(1 if enable else 0))


def alloc_cache(enable=True):
    """Process-wide: keep the matrix block (>= 1 GiB) a context gives up -- one per device -- for the next matrix on that device that fits it
    (default on: no clearing, no new mapping, no waiting; include/fasta_hip.h).  False also returns the kept blocks to the driver."""
    lib = load_library()
    _check(lib, lib.fh_alloc_cache(1 if enable else 0))


def release_cached(device=-1):
    """Return the kept matrix block of `device` (-1: of every device) to the driver."""
    lib = load_library()
    _check(lib, lib.fh_release_cached(int(device)))


def alloc_cache_hits():
    lib = load_library()
    n = _u64(0)
    _check(lib, lib.fh_alloc_cache_hits(C.byref(n)))
    return int(n.value)


def alloc_settle_waited():
    """Seconds the library has spent in that wait so far (process-wide)."""
    lib = load_library()
    v = _dbl(0.0)
    _check(lib, lib.fh_alloc_settle_waited(C.byref(v)))
    return float(v.value)


def comm_unique_id():
    lib = load_library()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib, lib.fh_comm_unique_id(buf))
    return buf.raw


class HipContext:
    """One context (stream, device-resident A, vectors, workspace) on one device -- or, with `devices=[...]`, one context over
    several row blocks of A driven from this process (fh_create_ex, ndev > 1).  Not thread-safe."""

    def __init__(self, device=0, storage="f64", devices=None, rccl_shell=False):
        """storage: "f64" (default) or "f32" -- the device copy of a dense A in float32 (opt-in throughput mode; vectors,
        accumulation and scalars stay float64).
        devices: list of device ids, one per row block (in-process row sharding): all different = one GPU each, sums over RCCL;
        all equal = every block on that one GPU, sums by an in-library kernel (what a one-GPU box can run)."""
        self.lib = load_library()
        self._h = _ctx()
        if storage not in STORAGE:
            raise ValueError('storage must be "f64" or "f32"')
        if devices is not None:
            devices = [int(d) for d in devices]
            if not devices:
                raise ValueError("devices must name at least one device")
            ids = (_i32 * len(devices))(*devices)
            flags = CREATE_RCCL_SHELL if rccl_shell else 0      # tests: the RCCL branch with one device / a repeated device id
            _check(self.lib, self.lib.fh_create_ex(len(devices), ids, STORAGE[storage] | flags, C.byref(self._h)))
            device = devices[0]
        elif storage == "f64":
            _check(self.lib, self.lib.fh_create(int(device), C.byref(self._h)))
        else:
            ids = (_i32 * 1)(int(device))
            _check(self.lib, self.lib.fh_create_ex(1, ids, STORAGE[storage], C.byref(self._h)))
        self.device = int(device)
        self.devices = devices
        self.storage = storage
        self._scal = np.zeros(NSCALARS)
        self._scal_p = self._scal.ctypes.data_as(_pd)
        self.sharded = False            # True once a multi-rank communicator is attached (comm_init)

    @classmethod
    def _borrowed(cls, lib, handle, device, storage):
        """A view of a shard of a multi-device context (fh_shard): same methods, never destroyed from here."""
        self = cls.__new__(cls)
        self.lib, self._h, self.device, self.devices, self.storage = lib, handle, device, None, storage
        self._scal = np.zeros(NSCALARS)
        self._scal_p = self._scal.ctypes.data_as(_pd)
        self.sharded = False
        self._view = True
        return self

    def shard_count(self):
        n = _i32(0)
        self._call("fh_shard_count", C.byref(n))
        return int(n.value)

    def shard(self, k):
        """(context view, row0, rows) of row block k; the view is valid while this context lives."""
        h, r0, rows = _ctx(), _u64(0), _u64(0)
        self._call("fh_shard", int(k), C.byref(h), C.byref(r0), C.byref(rows))
        dev = self.devices[k] if self.devices else self.device
        return HipContext._borrowed(self.lib, h, dev, self.storage), int(r0.value), int(rows.value)

    # ---- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_view", False):           # a borrowed shard: owned by its multi-device context
            self._h = _ctx()
            return
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.fh_destroy(self._h)
            self._h = _ctx()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        _check(self.lib, getattr(self.lib, name)(self._h, *args))

    # ---- operator / problem data -----------------------------------------------------------------
    def set_tuning(self, key, value):
        self._call("fh_set_tuning", int(key), int(value))

    def set_matrix(self, A):
        assert A.ndim == 2
        if self.storage == "f32" and A.dtype == np.float32:           # float32 data into float32 storage: no float64 detour
            A = np.ascontiguousarray(A)
            m, n = A.shape
            self._call("fh_set_matrix_f32", A.ctypes.data_as(C.POINTER(C.c_float)), m, n, n)
            return
        if A.dtype != np.float64 or not A.flags.c_contiguous:
            A = np.ascontiguousarray(A, dtype=np.float64)
        m, n = A.shape
        self._call("fh_set_matrix", A.ctypes.data_as(_pd), m, n, n)

    def generate_matrix(self, m, n, row0, seed, coef):
        self._call("fh_generate_matrix", int(m), int(n), int(row0), int(seed), float(coef))

    def get_matrix_rows(self, row0, nrows):
        m, n = self.shape()
        out = np.empty((nrows, n))
        self._call("fh_get_matrix_rows", int(row0), int(nrows), out.ctypes.data_as(_pd))
        return out

    def set_stencil(self, H, W):
        self._call("fh_set_stencil", int(H), int(W))

    def shape(self):
        m, n = _u64(0), _u64(0)
        self._call("fh_shape", C.byref(m), C.byref(n))
        return m.value, n.value

    def set_loss_lsq(self, b):
        b, p = _as_f64(np.ravel(b))
        self._call("fh_set_loss_lsq", p, b.size)

    def set_loss_logistic(self, labels):
        labels, p = _as_f64(np.ravel(labels))
        self._call("fh_set_loss_logistic", p, labels.size)

    def set_prox(self, kind, mu=0.0, lo=0.0, hi=0.0):
        self._call("fh_set_prox", int(kind), float(mu), float(lo), float(hi))

    def set_vector(self, which, v):
        v, p = _as_f64(np.ravel(v))
        self._call("fh_set_vector", int(which), p, v.size)

    def get_vector(self, which, length):
        out = np.empty(int(length))
        self._call("fh_get_vector", int(which), out.ctypes.data_as(_pd), out.size)
        return out

    # ---- solver steps (scalars come back as a fresh copy of the FH_S_* block) ---------------------
    def init(self):
        self._call("fh_init", self._scal_p)
        return self._scal.copy()

    def setup(self):
        """fh_setup: the two Lipschitz probes (in VEC_T0 / VEC_T1) and init() in one call -- one read of a dense least-squares A where the
        one-read kernel has a shape.  Scalars as init(), plus S_DG2 = ||grad(T0) - grad(T1)||^2 and S_DX2 = ||T0 - T1||^2 (VEC_T2 / T3: scratch)."""
        self._call("fh_setup", self._scal_p)
        return self._scal.copy()

    def gradient_at(self, src, dst):
        self._call("fh_gradient_at", int(src), int(dst))

    def diff_norm(self, a, b):
        out = _dbl(0.0)
        self._call("fh_diff_norm", int(a), int(b), C.byref(out))
        return out.value

    def fwd(self, tau):
        self._call("fh_fwd", float(tau), self._scal_p)
        return self._scal.copy()

    def adj(self, tau, accel=False, coef=0.0):
        self._call("fh_adj", float(tau), 1 if accel else 0, float(coef), self._scal_p)
        return self._scal.copy()

    def fwd_adj(self, tau):
        """K-fwd + K-adj (no acceleration) under one synchronisation: both halves of the scalar block."""
        self._call("fh_fwd_adj", float(tau), self._scal_p)
        return self._scal.copy()

    def fused_supported(self):
        """0 = none; 1 = dense one-pass kernel, recommended; 3 = dense, available but not recommended (small matrix: n < 16384 and fewer than 8 Mi elements);
        2 = stencil (one sweep replaces both launches)."""
        yes = _i32(0)
        self._call("fh_fused_supported", C.byref(yes))
        return int(yes.value)

    def fused_agree(self):
        """The one-pass verdict of fused_supported() settled over the ranks of the attached communicator (any 0 wins, else any 3):
        a COLLECTIVE call when the context has a communicator -- every rank calls it at the same point; the local query otherwise."""
        yes = _i32(0)
        self._call("fh_fused_agree", C.byref(yes))
        return int(yes.value)

    def coresident_probe(self, workgroups):
        """True if `workgroups` whole-CU workgroups run side by side on this device (what the dense one-pass kernel needs of #CUs)."""
        ok = _i32(0)
        self._call("fh_coresident_probe", int(workgroups), C.byref(ok))
        return bool(ok.value)

    def step(self, tau):
        """One-pass K-fwd + K-adj (no acceleration).  Raises HipTimeout if the bounded spins timed out."""
        self._call("fh_step", float(tau), self._scal_p)
        if self._scal[15] != 0.0:
            raise HipTimeout("fused one-pass kernel: team hand-off timed out (workgroups not co-resident?)")
        return self._scal.copy()

    def step_begin(self, tau):
        """Issue step(tau) without waiting for it: the host may drive other contexts until step_end()."""
        self._call("fh_step_begin", float(tau))

    def step_end(self):
        """Wait for the step issued by step_begin and return its scalars (HipTimeout as step())."""
        self._call("fh_step_end", self._scal_p)
        if self._scal[15] != 0.0:
            raise HipTimeout("fused one-pass kernel: team hand-off timed out (workgroups not co-resident?)")
        return self._scal.copy()

    def step_accel(self, tau, coef, restart):
        """One-pass K-fwd + K-adj with FISTA extrapolation: `coef` applies unless `restart` and this step's restart
        dot (returned in S_RDOT) exceeds 1e-30.  Dense operator."""
        self._call("fh_step_accel", float(tau), float(coef), 1 if restart else 0, self._scal_p)
        if self._scal[15] != 0.0:
            raise HipTimeout("fused one-pass kernel: team hand-off timed out (workgroups not co-resident?)")
        return self._scal.copy()

    def run_supported(self):
        """True if fh_run (the loop on the device, csrc/fh_run.h) has a kernel for this context's operator, loss and prox."""
        yes = _i32(0)
        self._call("fh_run_supported", C.byref(yes))
        return bool(yes.value)

    def run(self, max_steps, opts, state):
        """Up to `max_steps` FBS iterations in one persistent launch.  `opts`: RunOpts, `state`: RunState (updated in place).
        Returns the (steps, RUN_HIST) history block: residual, norm_residual, stepsize, f, objective, backtracks, alpha0,
        became-best (+ 2: acceleration restarted).  A grid-barrier timeout of the launch (FH_E_TIMEOUT) is NOT raised: the block then holds
        the iterations completed before it, `state.stopped == 3`, and context and state are those of the last completed iteration --
        the caller carries on with `iterate()` / `step()`."""
        hist = np.empty((int(max_steps), RUN_HIST))
        done = _i32(0)
        status = self.lib.fh_run(self._h, int(max_steps), C.byref(opts), C.byref(state), hist.ctypes.data_as(_pd), C.byref(done))
        if status != 0 and not (status == E_TIMEOUT and state.stopped == 3):
            _check(self.lib, status)
        return hist[:done.value]

    def iterate(self, max_steps, opts, state):
        """Up to `max_steps` FBS iterations driven by the library's host-side loop (fh_iterate, csrc/fh_host_iterate.h): every operator,
        loss, prox and sharding form; same arguments and history block as run().  If an iteration fails, the exception carries the records
        of the iterations completed before it as `.partial` (state and context are those of the last completed iteration)."""
        hist = np.empty((int(max_steps), RUN_HIST))
        done = _i32(0)
        status = self.lib.fh_iterate(self._h, int(max_steps), C.byref(opts), C.byref(state), hist.ctypes.data_as(_pd), C.byref(done))
        if status != 0:
            try:
                _check(self.lib, status)
            except HipError as exc:
                exc.partial = hist[:done.value]
                raise
        return hist[:done.value]

    def recovered_count(self, what):
        """In-launch timeouts this context got over: RECOVERED_LEVEL_FALLBACK / RECOVERED_LEVEL_FAILED / RECOVERED_RUN_TIMEOUT."""
        n = _u64(0)
        self._call("fh_recovered_count", int(what), C.byref(n))
        return int(n.value)

    def commit(self, save_best=False):
        self._call("fh_commit", 1 if save_best else 0)

    def apply(self, v, adjoint=False):
        m, n = self.shape()
        v, p = _as_f64(np.ravel(v))
        assert v.size == (m if adjoint else n)
        out = np.empty(n if adjoint else m)
        self._call("fh_apply", 1 if adjoint else 0, p, out.ctypes.data_as(_pd))
        return out

    def sync(self):
        self._call("fh_sync")

    # ---- row sharding ---------------------------------------------------------------------------
    def comm_init(self, nranks, rank, unique_id):
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        self._call("fh_comm_init", int(nranks), int(rank), buf)
        self.sharded = True

    def comm_count(self):
        """Ranks in the attached communicator as RCCL reports them (1 without one)."""
        n = _i32(0)
        self._call("fh_comm_count", C.byref(n))
        return int(n.value)

    def comm_selftest(self, count):
        """(max |error|, blocks summed over) of this context's exchange on a known pattern of `count` doubles (collective with a communicator)."""
        err, nb = _dbl(-1.0), _i32(0)
        self._call("fh_comm_selftest", int(count), C.byref(err), C.byref(nb))
        return err.value, int(nb.value)

    def comm_destroy(self):
        self._call("fh_comm_destroy")
        self.sharded = False

    def cu_count(self):
        """(CUs the device reports, CUs the dense one-pass kernel is launched on -- TUNE_FUSED_CUS)."""
        dev, used = _i32(0), _i32(0)
        self._call("fh_cu_count", C.byref(dev), C.byref(used))
        return int(dev.value), int(used.value)

    # ---- measurement ----------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._call("fh_timing_enable", 1 if on else 0)

    def timing_reset(self):
        self._call("fh_timing_reset")

    def timing_get(self, kernel_id):
        ms, cnt = _dbl(0.0), _u64(0)
        self._call("fh_timing_get", int(kernel_id), C.byref(ms), C.byref(cnt))
        return ms.value, cnt.value

    def timing_overlap(self, other, kernel_id):
        """(ms this context's latest timed launch of `kernel_id` ran, ms `other`'s ran, ms BOTH were running; <= 0: one after the other)."""
        a, b, both = _dbl(0.0), _dbl(0.0), _dbl(0.0)
        _check(self.lib, self.lib.fh_timing_overlap(self._h, other._h, int(kernel_id), C.byref(a), C.byref(b), C.byref(both)))
        return a.value, b.value, both.value

    def stream_read_ms(self, reps=3):
        ms, nbytes = _dbl(0.0), _u64(0)
        self._call("fh_stream_read_ms", int(reps), C.byref(ms), C.byref(nbytes))
        return ms.value, nbytes.value
