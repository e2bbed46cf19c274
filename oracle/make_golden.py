#!/usr/bin/env python3
"""Generate tests/golden/*.json with a 50-digit mpmath restatement of the GRAPE hot path.

TEST INFRASTRUCTURE (see oracle/grape_oracle.c).  The reference (Julia) cannot run here and
holds no golden vectors of its own, so these fixtures are the project's pinned numbers: the
math of SURVEY.md Appendix A evaluated at 50 significant digits (own Taylor/scaling-squaring
expm, no Pade, no float64 rounding anywhere), then rounded once to float64.  Both float64
oracles (C and NumPy) and the HIP path are checked against them.

    python oracle/make_golden.py            # rewrites tests/golden/*.json

Follows /root/reference: src/GRAPE.jl:25-96 (driver), :216-251 (sweeps), :261-303 (gradient,
both sign variants), src/timeevolution.jl:98-110 / :45-57, src/cost_functions.jl:13-17,
:99-111, src/solve.jl:164-196 (weights).
"""
import json
import os
import sys

import numpy as np
from mpmath import mp, mpc, mpf, matrix, eye, zeros

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402  (host-side workload generator only)

mp.dps = 50


def expm_mp(A):
    n = A.rows
    nrm = max(sum(abs(A[i, j]) for i in range(n)) for j in range(n))
    s = 0
    while nrm > mpf("0.25"):
        nrm /= 2
        s += 1
    As = A / (mpf(2) ** s)
    term = eye(n)
    out = eye(n)
    k = 1
    while True:
        term = term * As / k
        out = out + term
        if max(abs(term[i, j]) for i in range(n) for j in range(n)) < mpf(10) ** (-(mp.dps + 5)):
            break
        k += 1
    for _ in range(s):
        out = out * out
    return out


def to_mp(M):
    M = np.asarray(M)
    return matrix([[mpc(float(M[i, j].real), float(M[i, j].imag)) for j in range(M.shape[1])]
                   for i in range(M.shape[0])])


def tr(M):
    return sum(M[i, i] for i in range(M.rows))


def member(sys_type, variant, A, B, Xi, Xt, x, T):
    K, N = x.shape
    dt = mpf(T) / N
    A, Xi, Xt = to_mp(A), to_mp(Xi), to_mp(Xt)
    B = [to_mp(b) for b in B]
    n = A.rows
    P = []
    for i in range(N):
        H = zeros(n) if variant == 0 else A.copy()
        for j in range(K):
            H = H + B[j] * mpf(float(x[j, i]))
        if variant == 0:
            H = H + A
        P.append(expm_mp(H * mpc(0, -1) * dt))
    X = [None] * (N + 1)
    L = [None] * (N + 1)
    X[0], L[N] = Xi, Xt
    sand = sys_type != "UnitaryGate"
    for t in range(N):
        X[t + 1] = P[t] * X[t] * P[t].H if sand else P[t] * X[t]
    for t in range(N - 1, -1, -1):
        L[t] = P[t].H * L[t + 1] * P[t] if sand else P[t].H * L[t + 1]
    g = [[None] * N for _ in range(K)]
    for c in range(K):
        for t in range(N):
            if sand:
                g[c][t] = (mpc(0, 1) * dt * tr(L[t].H * (B[c] * X[t] - X[t] * B[c]))).real
            else:
                sgn = mpc(0, 1) if variant == 0 else mpc(0, -1)
                g[c][t] = 2 * (sgn * dt * tr(L[t].H * B[c] * X[t]) * tr(X[t].H * L[t])).real
    t = N - 1
    if sand:
        z = tr(L[t].H * X[t]) / n
        F = 1 - (z.real ** 2 + z.imag ** 2)
    else:
        z = tr(X[t].H * L[t])
        F = (z * z).real
    return F, g, P, X, L


def member_exact(sys_type, variant, objective, A, B, Xi, Xt, x, T):
    """Exact gradient at 50 digits (the ADGRAPE functional path): dP from the block-triangular exponential
    exp([[G, E], [0, G]]) = [[e^G, dexp_G(E)], [0, e^G]]."""
    K, N = x.shape
    dt = mpf(T) / N
    A, Xi, Xt = to_mp(A), to_mp(Xi), to_mp(Xt)
    B = [to_mp(b) for b in B]
    n = A.rows
    sand = sys_type != "UnitaryGate"
    Gs, P = [], []
    for i in range(N):
        H = zeros(n) if variant == 0 else A.copy()
        for j in range(K):
            H = H + B[j] * mpf(float(x[j, i]))
        if variant == 0:
            H = H + A
        Gs.append(H * mpc(0, -1) * dt)
        P.append(expm_mp(Gs[-1]))
    X = [None] * (N + 1)
    L = [None] * (N + 1)
    X[0], L[N] = Xi, Xt
    for t in range(N):
        X[t + 1] = P[t] * X[t] * P[t].H if sand else P[t] * X[t]
    for t in range(N - 1, -1, -1):
        L[t] = P[t].H * L[t + 1] * P[t] if sand else P[t].H * L[t + 1]
    Phi = tr(Xt.H * X[N])
    c1 = sand or objective == 1
    F = 1 - (Phi.real ** 2 + Phi.imag ** 2) / (n * n) if c1 else (Phi.conjugate() ** 2).real
    g = [[None] * N for _ in range(K)]
    for t in range(N):
        for c in range(K):
            big = zeros(2 * n)
            E = B[c] * mpc(0, -1) * dt
            for i in range(n):
                for j in range(n):
                    big[i, j] = Gs[t][i, j]
                    big[i + n, j + n] = Gs[t][i, j]
                    big[i, j + n] = E[i, j]
            ex = expm_mp(big)
            dP = matrix([[ex[i, j + n] for j in range(n)] for i in range(n)])
            if sand:
                dPhi = tr(L[t + 1].H * (dP * X[t] * P[t].H + P[t] * X[t] * dP.H))
            else:
                dPhi = tr(L[t + 1].H * dP * X[t])
            g[c][t] = (-2 * (Phi.conjugate() * dPhi).real / (n * n)) if c1 else (2 * (Phi * dPhi).real)
    return F, g


def make_exact_case(name, w, variant):
    out = {"name": name, "sys_type": w.sys_type, "variant": variant, "n": w.n, "m": w.n, "K": w.K, "N": w.N, "E": w.E,
           "T": w.T, "digits": mp.dps, "layout": "matrices column-major [re, im]; x, G, g as [c][t]",
           "inputs": {"A": [np_cm(a) for a in w.A], "B": [[np_cm(b) for b in bk] for bk in w.B],
                      "Xi": [np_cm(a) for a in w.Xi], "Xt": [np_cm(a) for a in w.Xt],
                      "wts": [float(v) for v in w.wts], "x": [[float(v) for v in row] for row in w.x]},
           "exact": {}}
    for objective in (0, 1):
        res = [member_exact(w.sys_type, variant, objective, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T) for k in range(w.E)]
        Ftot = sum(res[k][0] * mpf(float(w.wts[k])) for k in range(w.E))
        Gtot = [[sum(res[k][1][c][t] * mpf(float(w.wts[k])) for k in range(w.E)) for t in range(w.N)] for c in range(w.K)]
        out["exact"][f"objective{objective}"] = {
            "F": float(Ftot), "G": [[float(v) for v in row] for row in Gtot],
            "member_F": [float(r[0]) for r in res],
            "member_g": [[[float(v) for v in row] for row in r[1]] for r in res]}
    return out


def cm_list(M):
    """mp matrix -> [[re, im], ...] column-major float64"""
    return [[float(M[i, j].real), float(M[i, j].imag)] for j in range(M.cols) for i in range(M.rows)]


def np_cm(M):
    M = np.asarray(M)
    return [[float(M[i, j].real), float(M[i, j].imag)] for j in range(M.shape[1]) for i in range(M.shape[0])]


def make_case(name, w, variant):
    Fs, gs = [], []
    traj0 = None
    for k in range(w.E):
        F, g, P, X, L = member(w.sys_type, variant, w.A[k], w.B[k], w.Xi[k], w.Xt[k], w.x, w.T)
        Fs.append(F)
        gs.append(g)
        if k == 0:
            traj0 = (P, X, L)
    Ftot = sum(Fs[k] * mpf(float(w.wts[k])) for k in range(w.E))
    Gtot = [[sum(gs[k][c][t] * mpf(float(w.wts[k])) for k in range(w.E)) for t in range(w.N)]
            for c in range(w.K)]
    return {
        "name": name, "sys_type": w.sys_type, "variant": variant, "n": w.n, "m": int(np.asarray(w.Xi).shape[-1]),
        "K": w.K, "N": w.N, "E": w.E,
        "T": w.T, "digits": mp.dps,
        "layout": "matrices column-major [re, im]; x, G, g as [c][t]",
        "inputs": {"A": [np_cm(a) for a in w.A], "B": [[np_cm(b) for b in bk] for bk in w.B],
                   "Xi": [np_cm(a) for a in w.Xi], "Xt": [np_cm(a) for a in w.Xt],
                   "wts": [float(v) for v in w.wts], "x": [[float(v) for v in row] for row in w.x]},
        "expected": {"F": float(Ftot), "G": [[float(v) for v in row] for row in Gtot],
                     "member_F": [float(v) for v in Fs],
                     "member_g": [[[float(v) for v in row] for row in g] for g in gs],
                     "member0_props": [cm_list(m) for m in traj0[0]],
                     "member0_states": [cm_list(m) for m in traj0[1]],
                     "member0_costates": [cm_list(m) for m in traj0[2]]},
    }


def main(only=None):
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    wl = qoc.workloads
    cases = [
        ("st_2x2_single", wl.config("C1", N=10)),                       # README / test shape
        ("st_2x2_ens", wl.reference_ensemble("StateTransfer", 3, 8, 5.0)),
        ("ug_2x2_ens", wl.reference_ensemble("UnitaryGate", 3, 8, 5.0)),
        ("ug_4x4_ens", wl.config("C3", E=3, N=12)),
        ("st_4x4_ens", None),
        ("vec_4x1_liou", wl.liouville_vec(1, 3, 9, 1.0)),                 # n x 1 states, Hermitian superoperators
        ("vec_16x1_diss", wl.liouville_vec(2, 2, 4, 1.0, dissipative=True)),   # n x 1, non-Hermitian (tile kernels)
        ("vec_32x1_5q", "vec32"),                     # n x 1 states at n = 32 (two tiles per side): five-qubit operators of C5
        ("ug_4x4_bignorm", "bignorm"),                # dt |H| ~ 10-20: expm scaling + squaring path
        ("st_8x8_pairs", "rand8"),                    # tile kernels, two members per 16x16 tile
        ("ct_16x16_nonherm", "rand16"),               # tile kernels, non-Hermitian generator
    ]
    wv = wl.liouville_vec(2, 2, 3, 1.0, dissipative=True)      # 16 x 1 vec(rho), dissipative: stored zero-padded to 16 x 16
    pad = lambda X: np.concatenate([X, np.zeros((X.shape[0], 16, 15), complex)], axis=2)
    wv.Xi, wv.Xt = pad(wv.Xi), pad(wv.Xt)
    exact_cases = [("exact_ug_4x4", wl.config("C3", E=2, N=6), 1), ("exact_st_2x2", wl.reference_ensemble("StateTransfer", 3, 7, 5.0), 1),
                   ("exact_ug_2x2", wl.reference_ensemble("UnitaryGate", 2, 6, 5.0), 0), ("exact_vec_16x1_diss", wv, 1)]
    for name, w, variant in exact_cases:
        if only and name not in only:
            continue
        case = make_exact_case(name, w, variant)
        path = os.path.join(out, "exact", f"{name}.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(case, fh)
        print(path, os.path.getsize(path), "bytes  F0 =", case["exact"]["objective0"]["F"], " F1 =", case["exact"]["objective1"]["F"])
    for name, w in cases:
        if only and name not in only:
            continue
        if w == "bignorm":                             # C3 operators, T = 40 over 10 slices
            w = wl.config("C3", E=2, N=10)
            w.T = 40.0
            variants = (0,)
        elif w == "vec32":                             # C5's drift and controls acting on one state vector per member
            w = wl.config("C5", E=2, N=4)
            w.T = 0.8
            rng = np.random.default_rng(32)

            def unit():
                v = rng.standard_normal((32, 1)) + 1j * rng.standard_normal((32, 1))
                return v / np.linalg.norm(v)
            w.Xi = np.array([unit(), unit()])
            w.Xt = np.array([unit(), unit()])
            w.wts = np.array([0.4, 0.6])
            variants = (0,)
        elif isinstance(w, str):                       # seeded random problems for the MFMA tile kernels
            n = 8 if w == "rand8" else 16
            rng = np.random.default_rng(n)
            E, K, N = (3, 2, 5) if n == 8 else (2, 2, 4)

            def rnd(herm):
                M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
                return (M + M.conj().T) / 2 if herm else M
            herm = n == 8
            A = np.array([rnd(herm) for _ in range(E)]) * 0.6
            B = np.array([[rnd(herm) for _ in range(K)] for _ in range(E)]) * 0.4

            def rho():
                v = rng.standard_normal(n) + 1j * rng.standard_normal(n)
                v /= np.linalg.norm(v)
                return np.outer(v, v.conj())
            w = wl.Workload(name, "StateTransfer" if n == 8 else "CoherenceTransfer", n, K, N, E, 1.0, A, B,
                            np.array([rho() for _ in range(E)]), np.array([rho() for _ in range(E)]),
                            rng.uniform(0.3, 1.2, E), rng.uniform(-1, 1, (K, N)))
            variants = (0,)
        elif w is None:                                # 4x4 StateTransfer: C3 operators, rho targets
            w = wl.config("C3", E=2, N=10)
            rho0 = np.zeros((4, 4), complex); rho0[0, 0] = 1
            psi = np.array([1, 1j, -1, 0.5]) / np.linalg.norm([1, 1j, -1, 0.5])
            rhoT = np.outer(psi, psi.conj())
            w.sys_type = "StateTransfer"
            w.Xi = np.array([rho0, rho0]); w.Xt = np.array([rhoT, rhoT])
            variants = (0, 1)
        elif name == "vec_16x1_diss":
            variants = (0,)
        else:
            variants = (0, 1)
        for variant in variants:
            case = make_case(f"{name}_v{variant}", w, variant)
            path = os.path.join(out, f"{name}_v{variant}.json")
            with open(path, "w") as fh:
                json.dump(case, fh)
            print(path, os.path.getsize(path), "bytes  F =", case["expected"]["F"])


if __name__ == "__main__":
    main(set(sys.argv[1:]))              # optional: names of the cases to (re)generate
