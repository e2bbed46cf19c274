"""NumPy/SciPy restatement of the GRAPE hot path -- an independent second oracle.

TEST INFRASTRUCTURE ONLY (see oracle/grape_oracle.c).  It differs from the C oracle in
the one place the reference itself leaves open: the matrix exponential here is
scipy.linalg.expm (Al-Mohy & Higham 2009), a different algorithm from Julia's
LinearAlgebra.exp! that the C oracle restates, so agreement between the two checks that
the 1e-10 parity budget does not depend on the expm flavour.

Follows (relative to /root/reference): src/GRAPE.jl:25-96 and :103-166 (driver),
:216-251/:178-209 (sweeps), :261-303 (gradient), src/timeevolution.jl:98-110 / :45-57
(propagators), src/cost_functions.jl:13-17, :99-111 (figure of merit),
src/solve.jl:164-196 (ensemble closure).
"""
import numpy as np
from scipy.linalg import expm as _scipy_expm

UG, ST, CT = "UnitaryGate", "StateTransfer", "CoherenceTransfer"


def C1(KT, KN):
    """src/cost_functions.jl:13-17"""
    D = KT.shape[0]
    return 1.0 - abs(np.trace(KT.conj().T @ KN) / D) ** 2


def commutator(A, B):
    """src/tools.jl:17-19"""
    return A @ B - B @ A


def propagators(A, B, x, dt, variant=0, expm=_scipy_expm):
    K, N = x.shape
    out = []
    for i in range(N):
        if variant == 0:                      # src/timeevolution.jl:101-108
            H = np.zeros_like(A)
            for j in range(K):
                H = H + B[j] * x[j, i]
            G = (-1.0j * dt) * (H + A)
        else:                                 # src/timeevolution.jl:49-53
            H = A.copy()
            for j in range(K):
                H = H + B[j] * x[j, i]
            G = (-1.0j * dt) * H
        out.append(expm(G))
    return out


def member_eval(sys_type, A, B, Xi, Xt, x, T, variant=0, expm=_scipy_expm, trajectory=False):
    A = np.asarray(A, complex)
    B = np.asarray(B, complex)
    Xi = np.asarray(Xi, complex)
    Xt = np.asarray(Xt, complex)
    x = np.asarray(x, float)
    K, N = x.shape
    dt = T / N
    P = propagators(A, B, x, dt, variant, expm)
    X = [None] * (N + 1)
    L = [None] * (N + 1)
    X[0] = Xi
    L[N] = Xt
    sandwich = sys_type != UG
    for t in range(N):
        X[t + 1] = P[t] @ (X[t] @ P[t].conj().T) if sandwich else P[t] @ X[t]
    for t in range(N - 1, -1, -1):
        L[t] = P[t].conj().T @ (L[t + 1] @ P[t]) if sandwich else P[t].conj().T @ L[t + 1]
    g = np.zeros((K, N))
    for c in range(K):
        for t in range(N):
            if sandwich:
                g[c, t] = np.real(np.trace((1.0j * dt) * (L[t].conj().T @ commutator(B[c], X[t]))))
            else:
                sgn = 1.0j if variant == 0 else -1.0j
                g[c, t] = 2.0 * np.real((sgn * dt) * np.trace(L[t].conj().T @ B[c] @ X[t])
                                        * np.trace(X[t].conj().T @ L[t]))
    t = N - 1
    if sandwich:
        F = float(np.real(C1(L[t], X[t])))
    else:
        z = np.trace(X[t].conj().T @ L[t])
        F = float(np.real(z * z))
    if trajectory:
        return F, g, np.array(P), np.array(X), np.array(L)
    return F, g


def ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, T, variant=0, expm=_scipy_expm):
    E = len(A)
    F = 0.0
    G = None
    for k in range(E):
        f, g = member_eval(sys_type, A[k], B[k], Xi[k], Xt[k], x, T, variant, expm)
        F += f * wts[k]
        G = g * wts[k] if G is None else G + g * wts[k]
    return F, G
