"""ctypes binding of libgrape_hip.so (include/grape_hip.h) -- the only compute path.

There is deliberately no CPU fallback here: if the HIP library is missing or no gfx950
device is visible, construction raises GrapeError.  (The CPU restatement under oracle/ is
test infrastructure and is never imported from this package.)

This is the Python twin of julia/GrapeHIP.jl: both pack the operators once
(what init_ensemble produces, /root/reference/src/tools.jl:42-53), create a context
(init_GRAPE, src/grape_tools.jl:4-16) and then forward every call of the (F, G, x) closure
(src/solve.jl:164-196) to grape_eval.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYS_TYPE_CODES = {"UnitaryGate": 0, "StateTransfer": 1, "CoherenceTransfer": 2}
FLAG_KEEP_COSTATES = 1
FLAG_TIME_KERNELS = 2
FLAG_PHASE_STAMPS = 4
FLAG_FORCE_GENERAL = 8
FLAG_MEMBER_RESULTS = 16
FLAG_FORCE_COLLECTIVE = 32
FLAG_TIME_SAMPLED = 64
FLAG_GROUP_PEER_SUM = 128
MAX_DEVICES = 8
ABI_VERSION = 6

STATUS = {0: "GRAPE_OK", -1: "GRAPE_ERR_INVALID_ARG", -2: "GRAPE_ERR_UNSUPPORTED",
          -3: "GRAPE_ERR_NO_DEVICE", -4: "GRAPE_ERR_HIP", -5: "GRAPE_ERR_NOT_READY",
          -6: "GRAPE_ERR_ALLOC", -7: "GRAPE_ERR_TIMEOUT", -8: "GRAPE_ERR_COMM"}

# every symbol include/grape_hip.h declares
EXPORTS = ["grape_abi_version", "grape_create", "grape_destroy", "grape_set_operators",
           "grape_comm_unique_id", "grape_comm_attach", "grape_ipc_export", "grape_ipc_attach",
           "grape_eval", "grape_eval_device", "grape_eval_batch", "grape_eval_batch_device", "grape_lbfgs", "grape_lbfgs_get_trace",
           "grape_get_member_results", "grape_get_trajectory",
           "grape_get_kernel_time", "grape_get_kernel_samples", "grape_get_kernel_names", "grape_get_group_timing", "grape_get_phase_stamps",
           "grape_get_info",
           "grape_last_error"]


class GrapeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS.get(status, status)}: {message}")
        self.status = status


class GrapeConfig(C.Structure):
    _fields_ = [("sys_type", C.c_int32), ("variant", C.c_int32), ("n", C.c_int32),
                ("n_controls", C.c_int32), ("n_slices", C.c_int32), ("n_ensemble", C.c_int32),
                ("duration", C.c_double), ("device", C.c_int32), ("flags", C.c_int32),
                ("slices_per_lane", C.c_int32), ("waves_per_member", C.c_int32),
                ("expm_squarings", C.c_int32), ("max_batch", C.c_int32),
                ("n_state_cols", C.c_int32), ("n_devices", C.c_int32), ("device_ids", C.c_int32 * MAX_DEVICES),
                ("gradient", C.c_int32), ("objective", C.c_int32)]


class GrapeInfo(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("compute_units", C.c_int32),
                ("slices_per_lane", C.c_int32), ("waves_per_member", C.c_int32),
                ("expm_squarings", C.c_int32), ("kernel_family", C.c_int32), ("unitary_flow", C.c_int32),
                ("expm_theta", C.c_double), ("workspace_bytes", C.c_uint64), ("arch", C.c_char * 32),
                ("n_devices", C.c_int32), ("comm_size", C.c_int32), ("comm_rank", C.c_int32),
                ("members_first_device", C.c_int32), ("lane_pair", C.c_int32),
                ("states_stored", C.c_int32), ("rank_one_chain", C.c_int32),
                ("sparse_controls", C.c_int32), ("fused_forward", C.c_int32),
                ("time_chunks", C.c_int32), ("hoisted_controls", C.c_int32),
                ("expm_action", C.c_int32), ("prop_chain", C.c_int32), ("member_chunk", C.c_int32), ("reserved0", C.c_int32),
                ("workspace_budget_bytes", C.c_uint64), ("scaled_controls", C.c_int32), ("propagator_blocks", C.c_int32)]


class GrapeLbfgsOptions(C.Structure):
    _fields_ = [("memory", C.c_int32), ("max_iterations", C.c_int32), ("g_tol", C.c_double), ("f_tol", C.c_double),
                ("max_linesearch", C.c_int32), ("probes", C.c_int32), ("line_search", C.c_int32), ("reserved", C.c_int32)]


class GrapeLbfgsResult(C.Structure):
    _fields_ = [("minimum", C.c_double), ("g_norm", C.c_double), ("seconds", C.c_double), ("iterations", C.c_int32),
                ("evaluations", C.c_int32), ("status", C.c_int32), ("probes", C.c_int32),
                ("line_search", C.c_int32), ("ladder_fallbacks", C.c_int32)]


class GrapeCommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


def library_path():
    """libgrape_hip.so next to this file; GRAPE_HIP_LIB points diagnostics (tools/ablate.sh) at another build."""
    return os.environ.get("GRAPE_HIP_LIB") or os.path.join(_HERE, "libgrape_hip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ into libgrape_hip.so (in-tree)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return library_path()


def load_library():
    """dlopen libgrape_hip.so and declare the prototypes of include/grape_hip.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        try:                                   # a fresh checkout: build in-tree once (hipcc, gfx950)
            build_library()
        except Exception as exc:               # no hipcc / build failure: fail loudly, no fallback
            raise GrapeError(-3, f"{path} is not built and building it failed: {exc}") from exc
    L = C.CDLL(path)
    vp, dp, i32 = C.c_void_p, C.POINTER(C.c_double), C.c_int32
    L.grape_abi_version.restype = C.c_int
    L.grape_create.argtypes = [C.POINTER(GrapeConfig), C.POINTER(vp)]
    L.grape_destroy.argtypes = [vp]
    L.grape_set_operators.argtypes = [vp] * 6
    L.grape_comm_unique_id.argtypes = [C.POINTER(GrapeCommId)]
    L.grape_comm_attach.argtypes = [vp, C.POINTER(GrapeCommId), i32, i32]
    L.grape_ipc_export.argtypes = [vp, i32, vp]
    L.grape_ipc_attach.argtypes = [vp, vp, i32, i32]
    L.grape_eval.argtypes = [vp, vp, dp, vp]
    L.grape_eval_device.argtypes = [vp, vp, vp, vp]
    L.grape_eval_batch.argtypes = [vp, i32, vp, vp, vp]
    L.grape_eval_batch_device.argtypes = [vp, i32, vp, vp, vp]
    L.grape_lbfgs.argtypes = [vp, vp, C.POINTER(GrapeLbfgsOptions), vp, C.POINTER(GrapeLbfgsResult)]
    L.grape_get_member_results.argtypes = [vp, vp, vp]
    L.grape_get_trajectory.argtypes = [vp, i32, vp, vp, vp]
    L.grape_get_kernel_time.argtypes = [vp, dp, C.POINTER(C.c_int64), i32]
    L.grape_get_kernel_samples.argtypes = [vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.grape_get_kernel_names.argtypes = [vp, C.c_char_p, C.c_int32]
    L.grape_lbfgs_get_trace.argtypes = [vp, vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.grape_get_group_timing.argtypes = [vp, vp, i32]
    L.grape_get_phase_stamps.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.grape_get_info.argtypes = [vp, C.POINTER(GrapeInfo)]
    L.grape_last_error.argtypes = [vp]
    L.grape_last_error.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("grape_last_error",):
            getattr(L, name).restype = C.c_int
    _LIB = L
    return L


def _cm(M):
    """[..., i, j] complex -> contiguous buffer with each matrix column-major (Julia layout)."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(M, dtype=np.complex128), -1, -2))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class GrapeEngine:
    """One context = one device + one shard of the ensemble.

    A (E,n,n), B (E,K,n,n), Xi/Xt (E,n,n), wts (E,) -- natural numpy matrices A[k][i,j].
    eval(x) -> (F, G) with x, G of shape (K, N) (x[j, i] as in the reference)."""

    def __init__(self, sys_type, A, B, Xi, Xt, wts, T, n_slices, variant=0, device=-1, flags=0,
                 slices_per_lane=0, waves_per_member=0, expm_squarings=-1, member_results=False, max_batch=1,
                 devices=None, force_collective=False, gradient="reference", objective="fom"):
        """devices: list of HIP ordinals -> the library shards the ensemble over them itself
        (grape_config.n_devices / device_ids) and all-reduces [G, F] with RCCL once per evaluation.
        gradient: "reference" (the first-order grad_func!) or "exact" (derivative of the objective, 2 <= n <= 64);
        objective: "fom" (fom_func) or "c1" (the ADGRAPE functional C1(Xt, U Xi [U']); needs gradient="exact")."""
        self._h = None
        self._lib = load_library()
        A = np.asarray(A, dtype=np.complex128)
        B = np.asarray(B, dtype=np.complex128)
        if A.ndim != 3 or B.ndim != 4 or A.shape[1] != A.shape[2]:
            raise ValueError("A must be (E,n,n) and B (E,K,n,n)")
        E, n = A.shape[0], A.shape[1]
        K = B.shape[1]
        Xi = np.asarray(Xi, dtype=np.complex128)
        Xt = np.asarray(Xt, dtype=np.complex128)
        if Xi.ndim != 3 or Xi.shape[:2] != (E, n) or Xt.shape != Xi.shape or not 1 <= Xi.shape[2] <= n or B.shape != (E, K, n, n):
            raise ValueError("operator shapes disagree (B (E,K,n,n); Xi, Xt (E,n,m) with 1 <= m <= n)")
        m = Xi.shape[2]
        wts = np.ascontiguousarray(wts, dtype=np.float64)
        if wts.shape != (E,):
            raise ValueError("wts must have one weight per member")
        if member_results:
            flags |= FLAG_MEMBER_RESULTS
        if force_collective:
            flags |= FLAG_FORCE_COLLECTIVE
        devices = list(devices) if devices is not None else []
        if len(devices) > MAX_DEVICES:
            raise ValueError(f"at most {MAX_DEVICES} devices")
        code = SYS_TYPE_CODES[sys_type] if isinstance(sys_type, str) else int(sys_type)
        self.sys_type, self.n, self.K, self.N, self.E, self.T = sys_type, n, K, int(n_slices), E, float(T)
        self.m = m
        ids = (C.c_int32 * MAX_DEVICES)(*(devices + [0] * (MAX_DEVICES - len(devices))))
        if len(devices) == 1:
            device = devices[0]
        cfg = GrapeConfig(code, int(variant), n, K, int(n_slices), E, float(T), int(device), int(flags),
                          int(slices_per_lane), int(waves_per_member), int(expm_squarings), int(max_batch),
                          0 if m == n else m, len(devices) if len(devices) > 1 else 0, ids,
                          {"reference": 0, "exact": 1}[gradient], {"fom": 0, "c1": 1}[objective])
        self.max_batch = max(1, int(max_batch))
        h = C.c_void_p()
        rc = self._lib.grape_create(C.byref(cfg), C.byref(h))
        if rc:
            raise GrapeError(rc, self._lib.grape_last_error(None).decode())
        self._h = h
        self._F = C.c_double()
        self._check(self._lib.grape_set_operators(h, _p(_cm(A)), _p(_cm(B)), _p(_cm(Xi)), _p(_cm(Xt)),
                                                  _p(wts)))

    def set_operators(self, A, B, Xi, Xt, wts):
        """grape_set_operators again on the same context (same shapes): new members' operators, new states."""
        A, B = np.asarray(A, np.complex128), np.asarray(B, np.complex128)
        Xi, Xt = np.asarray(Xi, np.complex128), np.asarray(Xt, np.complex128)
        wts = np.ascontiguousarray(wts, dtype=np.float64)
        if A.shape != (self.E, self.n, self.n) or B.shape != (self.E, self.K, self.n, self.n) or \
                Xi.shape != (self.E, self.n, self.m) or Xt.shape != Xi.shape or wts.shape != (self.E,):
            raise ValueError("operator shapes must match the context")
        self._check(self._lib.grape_set_operators(self._h, _p(_cm(A)), _p(_cm(B)), _p(_cm(Xi)), _p(_cm(Xt)), _p(wts)))

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc:
            raise GrapeError(rc, self._lib.grape_last_error(self._h).decode())

    def close(self):
        if self._h is not None:
            self._lib.grape_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def info(self):
        inf = GrapeInfo()
        self._check(self._lib.grape_get_info(self._h, C.byref(inf)))
        return {f: (getattr(inf, f).decode() if f == "arch" else getattr(inf, f)) for f, _ in inf._fields_}

    # ------------------------------------------------------------------ one process per GPU
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL bootstrap token (rank 0 creates it, every rank passes it to comm_attach)."""
        lib = load_library()
        cid = GrapeCommId()
        rc = lib.grape_comm_unique_id(C.byref(cid))
        if rc:
            raise GrapeError(rc, lib.grape_last_error(None).decode())
        return bytes(bytearray(cid)[:128])

    def comm_attach(self, token, rank, n_ranks):
        """Join the communicator: from now on every eval()/eval_device() of this context (this rank's
        member shard) ends in the single all-reduce of [G, F] over the ranks, inside the library."""
        cid = GrapeCommId.from_buffer_copy(bytes(token))
        self._check(self._lib.grape_comm_attach(self._h, C.byref(cid), int(rank), int(n_ranks)))

    def ipc_export(self, n_ranks):
        """64 opaque bytes naming this rank's exchange mailbox (grape_ipc_export): all-gather them, then ipc_attach."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.grape_ipc_export(self._h, int(n_ranks), buf))
        return buf.raw

    def ipc_attach(self, handles, rank, n_ranks):
        """handles: the n_ranks exported byte strings in rank order.  From now on every eval()/eval_device()/lbfgs() of this
        context ends in the mailbox all-reduce of [G, F] over the ranks (no RCCL)."""
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != 64 * int(n_ranks):
            raise ValueError("ipc_attach: need n_ranks handles of 64 bytes")
        self._check(self._lib.grape_ipc_attach(self._h, C.create_string_buffer(blob, len(blob)), int(rank), int(n_ranks)))

    # ------------------------------------------------------------------ evaluation
    def eval(self, x, want_F=True, want_G=True):
        """grape_eval: host x (K,N) -> (F, G); either may be skipped like Optim's only_fg!."""
        x = np.asarray(x, dtype=np.float64)
        if x.shape != (self.K, self.N):
            raise ValueError(f"x must be ({self.K},{self.N})")
        xf = np.ascontiguousarray(x.T)
        F = C.c_double()
        G = np.empty((self.N, self.K)) if want_G else None
        self._check(self._lib.grape_eval(self._h, _p(xf), C.byref(F) if want_F else None, _p(G)))
        return (F.value if want_F else None), (np.ascontiguousarray(G.T) if want_G else None)

    def eval_cm(self, xf, G_out=None):
        """grape_eval on caller-owned buffers in the library's own layout, no copies: xf is x as (K,N) COLUMN-major
        memory, i.e. a C-contiguous float64 array of shape (N, K); G_out (same shape, or None to skip G) receives
        the gradient in that layout.  Returns F.  This is what a compiled caller (the Julia ccall) does per
        optimiser step; eval() is the convenience form with natural (K, N) arrays."""
        if xf.dtype != np.float64 or not xf.flags.c_contiguous or xf.shape != (self.N, self.K):
            raise ValueError(f"xf must be a C-contiguous float64 array of shape ({self.N},{self.K})")
        if G_out is not None and (G_out.dtype != np.float64 or not G_out.flags.c_contiguous or G_out.shape != xf.shape):
            raise ValueError("G_out must match xf")
        rc = self._lib.grape_eval(self._h, xf.ctypes.data, C.byref(self._F), G_out.ctypes.data if G_out is not None else None)
        if rc:
            self._check(rc)
        return self._F.value

    def bind_eval(self, xf, G_out):
        """Pre-bound grape_eval on fixed caller buffers (layout as eval_cm): returns a zero-argument callable that
        runs one evaluation and returns F.  The ctypes argument objects are built once, so a call costs what the
        foreign-function call itself costs -- the closest Python gets to the Julia `ccall` in julia/GrapeHIP.jl."""
        if xf.dtype != np.float64 or not xf.flags.c_contiguous or xf.shape != (self.N, self.K):
            raise ValueError(f"xf must be a C-contiguous float64 array of shape ({self.N},{self.K})")
        if G_out.dtype != np.float64 or not G_out.flags.c_contiguous or G_out.shape != xf.shape:
            raise ValueError("G_out must match xf")
        fn, h, px, pg, F = self._lib.grape_eval, self._h, C.c_void_p(xf.ctypes.data), C.c_void_p(G_out.ctypes.data), self._F
        pF = C.byref(F)
        check = self._check

        def call():
            rc = fn(h, px, pF, pg)
            if rc:
                check(rc)
            return F.value
        call.buffers = (xf, G_out)              # keep them alive as long as the callable lives
        return call

    LBFGS_STATUS = {0: "g_tol reached", 1: "f_tol reached", 2: "max iterations", 3: "line search failed",
                    4: "zero step (Optim: x converged)"}

    LINE_SEARCH = {"hagerzhang": 0, "hz": 0, "optim": 1, "hagerzhang_strict": 1, "ladder": 2}

    def lbfgs(self, x0, memory=0, iterations=0, g_tol=-1.0, f_tol=0.0, max_linesearch=0, probes=0, line_search="hagerzhang"):
        """grape_lbfgs: device-resident L-BFGS from x0 (K,N) -> (x_min (K,N), result dict).  Optim LBFGS()
        defaults when the options are left at 0 / negative.  line_search: "hagerzhang" (Optim's line search, the
        initial step accepted when it satisfies the Wolfe conditions), "optim" (Hager-Zhang exactly as Optim runs it
        behind InitialStatic) or "ladder" (`probes` step lengths per batched launch; needs max_batch >= probes)."""
        x0 = np.asarray(x0, dtype=np.float64)
        if x0.shape != (self.K, self.N):
            raise ValueError(f"x0 must be ({self.K},{self.N})")
        xf = np.ascontiguousarray(x0.T)
        out = np.empty_like(xf)
        opts = GrapeLbfgsOptions(int(memory), int(iterations), float(g_tol), float(f_tol), int(max_linesearch), int(probes),
                                 self.LINE_SEARCH[line_search] if isinstance(line_search, str) else int(line_search), 0)
        res = GrapeLbfgsResult()
        self._check(self._lib.grape_lbfgs(self._h, _p(xf), C.byref(opts), _p(out), C.byref(res)))
        info = {f: getattr(res, f) for f, _ in res._fields_}
        info["message"] = self.LBFGS_STATUS.get(res.status, "?")
        return np.ascontiguousarray(out.T), info

    def lbfgs_trace(self):
        """grape_lbfgs_get_trace: (alphas, evaluations) of the last lbfgs() run, one entry per iteration."""
        cnt = C.c_int32()
        self._check(self._lib.grape_lbfgs_get_trace(self._h, None, None, 0, C.byref(cnt)))
        al, ev = np.empty(cnt.value), np.empty(cnt.value, dtype=np.int32)
        self._check(self._lib.grape_lbfgs_get_trace(self._h, _p(al), _p(ev), cnt.value, C.byref(cnt)))
        return al, ev

    def eval_batch(self, X):
        """grape_eval_batch: X (n_x, K, N) control arrays -> (F (n_x,), G (n_x, K, N)); entry b equals
        eval(X[b]).  An extension for multi-start optimisation; needs max_batch >= n_x."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 3 or X.shape[1:] != (self.K, self.N):
            raise ValueError(f"X must be (n_x, {self.K}, {self.N})")
        n_x = X.shape[0]
        xf = np.ascontiguousarray(np.swapaxes(X, 1, 2))            # each (K,N) column-major
        F = np.empty(n_x)
        G = np.empty((n_x, self.N, self.K))
        self._check(self._lib.grape_eval_batch(self._h, n_x, _p(xf), _p(F), _p(G)))
        return F, np.ascontiguousarray(np.swapaxes(G, 1, 2))

    def eval_device(self, d_x_ptr, d_fg_ptr, stream=0):
        """grape_eval_device with raw device pointers (e.g. torch tensor .data_ptr())."""
        self._check(self._lib.grape_eval_device(self._h, C.c_void_p(d_x_ptr), C.c_void_p(d_fg_ptr),
                                                C.c_void_p(stream)))

    def eval_batch_device(self, n_x, d_x_ptr, d_fg_ptr, stream=0):
        """grape_eval_batch_device with raw device pointers: d_x (K,N,n_x) f64, d_fg n_x blocks of (K*N+1) f64."""
        self._check(self._lib.grape_eval_batch_device(self._h, int(n_x), C.c_void_p(d_x_ptr), C.c_void_p(d_fg_ptr),
                                                      C.c_void_p(stream)))

    def member_results(self):
        foms = np.empty(self.E)
        grads = np.empty((self.E, self.N, self.K))
        self._check(self._lib.grape_get_member_results(self._h, _p(foms), _p(grads)))
        return foms, np.ascontiguousarray(np.swapaxes(grads, 1, 2))

    def trajectory(self, member, costates=False, states=True):
        n, N, m = self.n, self.N, self.m
        P = np.empty((N, n, n), np.complex128)
        X = np.empty((N + 1, m, n), np.complex128) if states else None       # column-major n x m each
        Lc = np.empty((N + 1, m, n), np.complex128) if costates else None
        self._check(self._lib.grape_get_trajectory(self._h, int(member), _p(P), _p(X), _p(Lc)))
        sw = lambda a: None if a is None else np.ascontiguousarray(np.swapaxes(a, -1, -2))
        return (sw(P), sw(X), sw(Lc)) if costates else (sw(P), sw(X))

    def phase_stamps(self):
        """(E*W, 8) uint64 stamps of the last evaluation (FLAG_PHASE_STAMPS)."""
        cnt = C.c_int64()
        self._check(self._lib.grape_get_phase_stamps(self._h, None, 0, C.byref(cnt)))
        out = np.empty(cnt.value, dtype=np.uint64)
        self._check(self._lib.grape_get_phase_stamps(self._h, _p(out), cnt.value, C.byref(cnt)))
        return out.reshape(-1, 8)

    def group_timing(self, reset=False):
        """multi-device contexts: mean host-side microseconds per grape_eval since the last reset."""
        out = np.zeros(6)
        self._check(self._lib.grape_get_group_timing(self._h, _p(out), int(reset)))
        return dict(zip(("evaluations", "stage_x_us", "issue_skew_us", "sum_issue_us", "wait_us", "total_us"), out.tolist()))

    def kernel_samples(self, capacity=65536):
        """(total_ms, first_ms): per-evaluation kernel durations since the last kernel_time(reset=True) (FLAG_TIME_KERNELS);
        first_ms = the expm part of the n = 5..32 family (0 for n <= 4)."""
        cnt = C.c_int64()
        self._check(self._lib.grape_get_kernel_samples(self._h, None, None, 0, C.byref(cnt)))
        n = min(int(cnt.value), int(capacity))
        tot, first = np.empty(n), np.empty(n)
        if n:
            self._check(self._lib.grape_get_kernel_samples(self._h, _p(tot), _p(first), n, C.byref(cnt)))
        return tot, first

    def kernel_names(self):
        """The kernels the last evaluation launched, in launch order (grape_get_kernel_names): the names a rocprofv3
        kernel trace shows, without namespace and template arguments."""
        need = self._lib.grape_get_kernel_names(self._h, None, 0)
        if need < 0:
            self._check(need)
        buf = C.create_string_buffer(max(int(need), 1))
        self._check(min(self._lib.grape_get_kernel_names(self._h, buf, len(buf)), 0))
        return [k for k in buf.value.decode().split(";") if k]

    def kernel_time(self, reset=False):
        ms = C.c_double()
        cnt = C.c_int64()
        self._check(self._lib.grape_get_kernel_time(self._h, C.byref(ms), C.byref(cnt), int(reset)))
        return ms.value, cnt.value
