"""ctypes loader for the C oracle (oracle/grape_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of grape_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package never
imports this module.

Python-side conventions (shared with the product's host layer so the same arrays can be
fed to both):
  A   (E, n, n) complex128, A[k][i, j]            B   (E, K, n, n)
  Xi  (E, n, n)                                    Xt  (E, n, n)
  wts (E,) float64                                 x   (K, N) float64  (x[j, i], like Julia)
The C side wants Julia's column-major matrices; `pack_cm` does that transposition.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYS_TYPES = {"UnitaryGate": 0, "StateTransfer": 1, "CoherenceTransfer": 2}


def build(force=False):
    so = os.path.join(_HERE, "libgrape_oracle.so")
    src = os.path.join(_HERE, "grape_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libgrape_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp, vp = C.POINTER(C.c_double), C.c_void_p
        L.oracle_expm.argtypes = [C.c_int, vp, vp]
        L.oracle_expm.restype = C.c_int
        L.oracle_member_eval.argtypes = [C.c_int] * 5 + [C.c_double] + [vp] * 4 + [vp, dp, vp, vp, vp, vp]
        L.oracle_member_eval.restype = C.c_int
        L.oracle_ensemble_eval.argtypes = [C.c_int] * 6 + [C.c_double] + [vp] * 6 + [dp, vp, vp, vp, C.c_int]
        L.oracle_ensemble_eval.restype = C.c_int
        L.oracle_member_eval_rect.argtypes = [C.c_int] * 5 + [C.c_double] + [vp] * 4 + [vp, dp, vp, vp, vp, vp]
        L.oracle_member_eval_rect.restype = C.c_int
        L.oracle_member_exact.argtypes = [C.c_int] * 6 + [C.c_double] + [vp] * 4 + [vp, dp, vp]
        L.oracle_member_exact.restype = C.c_int
        L.oracle_C1.argtypes = [C.c_int, vp, vp]
        L.oracle_C1.restype = C.c_double
        _LIB = L
    return _LIB


def pack_cm(M):
    """[..., i, j] -> contiguous memory with each trailing matrix column-major."""
    M = np.asarray(M, dtype=np.complex128)
    return np.ascontiguousarray(np.swapaxes(M, -1, -2))


def unpack_cm(buf, shape):
    """inverse of pack_cm for a buffer of `shape` = (..., n, n) logical matrices."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(buf).reshape(shape), -1, -2))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def expm(A):
    A = np.asarray(A, dtype=np.complex128)
    n = A.shape[0]
    a = pack_cm(A)
    out = np.empty_like(a)
    rc = lib().oracle_expm(n, _p(a), _p(out))
    if rc:
        raise RuntimeError(f"oracle_expm failed rc={rc}")
    return unpack_cm(out, (n, n))


def C1(KT, KN):
    KT = pack_cm(KT)
    KN = pack_cm(KN)
    return lib().oracle_C1(KT.shape[0], _p(KT), _p(KN))


def member_eval(sys_type, A, B, Xi, Xt, x, T, variant=0, trajectory=False):
    """One member: returns (F, G[K,N]) or (F, G, props[N,n,n], states[N+1,n,n], costates[N+1,n,n])."""
    st = SYS_TYPES[sys_type] if isinstance(sys_type, str) else int(sys_type)
    if np.asarray(Xi).shape[-1] != np.asarray(A).shape[-1]:
        return member_eval_rect(A, B, Xi, Xt, x, T, variant, trajectory, st)
    A = pack_cm(A)
    n = A.shape[0]
    B = pack_cm(B)
    K = B.shape[0]
    Xi, Xt = pack_cm(Xi), pack_cm(Xt)
    x = np.asarray(x, dtype=np.float64)
    assert x.shape[0] == K
    N = x.shape[1]
    xf = np.ascontiguousarray(x.T)          # memory: x[j + i*K]
    grad = np.empty((N, K))
    fom = C.c_double()
    props = np.empty((N, n, n), np.complex128) if trajectory else None
    sts = np.empty((N + 1, n, n), np.complex128) if trajectory else None
    cos = np.empty((N + 1, n, n), np.complex128) if trajectory else None
    rc = lib().oracle_member_eval(
        st, variant, n, K, N, float(T), _p(A), _p(B), _p(Xi), _p(Xt), _p(xf), C.byref(fom),
        _p(grad), _p(props) if trajectory else None, _p(sts) if trajectory else None,
        _p(cos) if trajectory else None)
    if rc:
        raise RuntimeError(f"oracle_member_eval failed rc={rc}")
    G = np.ascontiguousarray(grad.T)
    if trajectory:
        sw = lambda a: np.ascontiguousarray(np.swapaxes(a, -1, -2))
        return fom.value, G, sw(props), sw(sts), sw(cos)
    return fom.value, G


def member_exact(sys_type, A, B, Xi, Xt, x, T, variant=0, objective=0):
    """Exact gradient of objective 0 (GRAPE fom_func) or 1 (the ADGRAPE C1 functional): (F, G[K,N])."""
    st = SYS_TYPES[sys_type] if isinstance(sys_type, str) else int(sys_type)
    A, B, Xi, Xt = pack_cm(A), pack_cm(B), pack_cm(Xi), pack_cm(Xt)
    n, K = A.shape[0], B.shape[0]
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[1]
    xf = np.ascontiguousarray(x.T)
    grad = np.empty((N, K))
    fom = C.c_double()
    rc = lib().oracle_member_exact(int(objective), st, int(variant), n, K, N, float(T), _p(A), _p(B), _p(Xi), _p(Xt),
                                   _p(xf), C.byref(fom), _p(grad))
    if rc:
        raise RuntimeError(f"oracle_member_exact failed rc={rc}")
    return fom.value, np.ascontiguousarray(grad.T)


def ensemble_exact(sys_type, A, B, Xi, Xt, wts, x, T, variant=0, objective=0, per_member=False):
    """sum_k w_k (F_k, g_k) of member_exact, k ascending (the ADGRAPE ensemble functionals, src/solve.jl:317-361)."""
    res = [member_exact(sys_type, A[k], B[k], Xi[k], Xt[k], x, T, variant, objective) for k in range(len(A))]
    foms = np.array([r[0] for r in res])
    grads = np.array([r[1] for r in res])
    wts = np.asarray(wts, dtype=np.float64)
    F = 0.0
    G = np.zeros_like(grads[0])
    for k in range(len(A)):
        F += foms[k] * wts[k]
        G += grads[k] * wts[k]
    return (F, G, foms, grads) if per_member else (F, G)


def member_eval_rect(A, B, Xi, Xt, x, T, variant=0, trajectory=False, st=0):
    """n x m states (m < n), UnitaryGate dispatch only: (F, G) [+ props (N,n,n), states, costates (N+1,n,m)]."""
    if st != 0:
        raise ValueError("n x m states need UnitaryGate (the sandwich X P' is undefined for m != n)")
    A, B, Xi, Xt = pack_cm(A), pack_cm(B), pack_cm(Xi), pack_cm(Xt)
    n, K, m = A.shape[0], B.shape[0], Xi.shape[0]
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[1]
    xf = np.ascontiguousarray(x.T)
    grad = np.empty((N, K))
    fom = C.c_double()
    props = np.empty((N, n, n), np.complex128) if trajectory else None
    sts = np.empty((N + 1, m, n), np.complex128) if trajectory else None
    cos = np.empty((N + 1, m, n), np.complex128) if trajectory else None
    rc = lib().oracle_member_eval_rect(variant, n, m, K, N, float(T), _p(A), _p(B), _p(Xi), _p(Xt), _p(xf),
                                       C.byref(fom), _p(grad), _p(props) if trajectory else None,
                                       _p(sts) if trajectory else None, _p(cos) if trajectory else None)
    if rc:
        raise RuntimeError(f"oracle_member_eval_rect failed rc={rc}")
    G = np.ascontiguousarray(grad.T)
    if trajectory:
        sw = lambda a: np.ascontiguousarray(np.swapaxes(a, -1, -2))
        return fom.value, G, sw(props), sw(sts), sw(cos)
    return fom.value, G


def ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, T, variant=0, n_threads=1, per_member=False):
    """Ensemble closure (src/solve.jl:164-196): returns (F, G[K,N]) [+ foms[E], grads[E,K,N]]."""
    st = SYS_TYPES[sys_type] if isinstance(sys_type, str) else int(sys_type)
    if np.asarray(Xi).shape[-1] != np.asarray(A).shape[-1]:         # n x m states: member loop in k order, then the weighted sums
        res = [member_eval(sys_type, A[k], B[k], Xi[k], Xt[k], x, T, variant) for k in range(len(A))]
        foms = np.array([r[0] for r in res])
        grads = np.array([r[1] for r in res])
        wts = np.asarray(wts, dtype=np.float64)
        F = 0.0
        for k in range(len(A)):
            F += foms[k] * wts[k]
        G = np.zeros_like(grads[0])
        for k in range(len(A)):
            G += grads[k] * wts[k]
        return (F, G, foms, grads) if per_member else (F, G)
    A = pack_cm(A)
    E, n = A.shape[0], A.shape[1]
    B = pack_cm(B)
    K = B.shape[1]
    Xi, Xt = pack_cm(Xi), pack_cm(Xt)
    wts = np.ascontiguousarray(wts, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[1]
    xf = np.ascontiguousarray(x.T)
    G = np.empty((N, K))
    F = C.c_double()
    foms = np.empty(E) if per_member else None
    grads = np.empty((E, N, K)) if per_member else None
    rc = lib().oracle_ensemble_eval(
        st, variant, n, K, N, E, float(T), _p(A), _p(B), _p(Xi), _p(Xt), _p(wts), _p(xf),
        C.byref(F), _p(G), _p(foms) if per_member else None, _p(grads) if per_member else None,
        int(n_threads))
    if rc:
        raise RuntimeError(f"oracle_ensemble_eval failed rc={rc}")
    Gt = np.ascontiguousarray(G.T)
    if per_member:
        return F.value, Gt, foms, np.ascontiguousarray(np.swapaxes(grads, 1, 2))
    return F.value, Gt
