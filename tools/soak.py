#!/usr/bin/env python3
"""Randomised differential soak: many random problems of every kernel family (n = 2..64, all system types, both
variants, Hermitian / non-Hermitian generators, pure / mixed / rectangular states, dense / sparse controls, the data-flow
flags; round 5: the n = 33..64 grid family and a workspace budget that forces member-chunked evaluation as further random
dimensions) through the C ABI against the CPU oracle at the 1e-10 parity bar.  usage: tools/soak.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from conftest import assert_parity  # noqa: E402
from oracle import grape_oracle as orc  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
F = qoc.engine
fails = 0
n_action = 0
n_chunked = 0
n_grid = 0
n_member_chunked = 0
n_scaled = 0
t0 = time.time()
for i in range(cases):
    n = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 40, 48, 57, 64, 66, 80, 97],      # (> 64: round 6, matrix-core products)
                       p=[.1, .08, .13, .06, .05, .07, .07, .07, .1, .05, .05, .07, .02, .02, .02, .02, .014, .002, .002, .002]))
    K = int(rng.integers(1, 9))
    N = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 100, 257])) if n <= 16 else int(rng.choice([1, 2, 5, 9, 20] if n <= 32 else [1, 2, 5, 9, 17, 24]))      # (17, 24: round 6, chunked time axis beyond 32)
    E = int(rng.choice([1, 2, 3, 5, 9, 17])) if n <= 16 else int(rng.choice([1, 2, 3]))
    # a workspace budget of a few members' arrays: the evaluation walks the ensemble in blocks (bitwise the unchunked result,
    # tests/test_gpu_chunked.py; here against the oracle like every other case)
    chunk_budget = bool(E >= 3 and rng.random() < 0.2)
    sys_type = str(rng.choice(["UnitaryGate", "StateTransfer", "CoherenceTransfer"]))
    variant = int(rng.integers(0, 2))
    herm = bool(rng.integers(0, 2))
    sparse = bool(rng.integers(0, 2))
    states = str(rng.choice(["pure", "mixed", "rect", "vec"]))      # vec: n x 1 states (UnitaryGate), pure states otherwise
    flag = int(rng.choice([0, 0, F.FLAG_FORCE_GENERAL, F.FLAG_KEEP_COSTATES]))
    shared_ctrl = bool(rng.random() < 0.4)             # the SAME control operators for every member: the hoisted control sum

    def mat(h):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if h else M
    # generator norms as in the BASELINE configs up to a few units of dt |H|_1 (each squaring of a non-normal propagator
    # doubles its rounding error: the sweep's Taylor-8 scales further down than the oracle's Pade-13)
    gscale = min(1.0, 4.0 / n)
    A = np.array([mat(herm) for _ in range(E)]) * 0.6 * gscale
    if sparse:
        B = np.zeros((E, K, n, n), complex)
        for k in range(E):
            for c in range(K):
                # (now and then operators with 65 .. 256 non-zeros, n >= 12: control lists longer than a wavefront)
                for _ in range(int(rng.integers(1, 14)) if n < 12 or rng.random() < 0.8 else int(rng.integers(36, 110))):
                    a, b = rng.integers(0, n, 2)
                    v = (rng.standard_normal() + 1j * rng.standard_normal()) * (1.0 if gscale == 1.0 else 0.5)
                    if a == b:
                        B[k, c, a, a] = v.real
                    else:
                        B[k, c, a, b] = v
                        B[k, c, b, a] = np.conj(v) if herm else 0.3 * v
    else:
        B = np.array([[mat(herm) for _ in range(K)] for _ in range(E)]) * 0.4 * gscale

    if shared_ctrl:
        B = np.broadcast_to(B[0], B.shape).copy()
        if rng.random() < 0.35:                        # round 6: B_k = s_k B_0 (EnsembleProblem.B_g with amplitude inhomogeneity)
            B = B * (1.0 + 0.2 * rng.uniform(-1, 1, E))[:, None, None, None]
            n_scaled += 1
        os.environ["GRAPE_HOIST"] = "1"                # (forced for the small ensembles of a soak run; n <= 4 has no such path)
        # rank-one states, n = 9..32: the vector flow of action_thin.hip (forced likewise), or the flows it replaces
        os.environ["GRAPE_ACTION"] = "1" if rng.random() < 0.6 else "0"
        os.environ["GRAPE_THIN_DPP"] = "1" if rng.random() < 0.6 else "0"      # (9..16, not the Taylor flow: chain_prop_kernel / sweep_thin.hip)
        os.environ["GRAPE_ACT_WHOLE"] = "1" if rng.random() < 0.5 else "0"     # (round 4: one or two members per wave of action_parts_kernel)
    else:
        os.environ.pop("GRAPE_HOIST", None)
        os.environ.pop("GRAPE_ACTION", None)
        os.environ.pop("GRAPE_THIN_DPP", None)
        os.environ.pop("GRAPE_ACT_WHOLE", None)
    # rank-one states, 9..16, at least 64 slices: the chunked propagator chain (forced for dense controls too), or the
    # library's own choice
    if rng.random() < 0.5:
        os.environ["GRAPE_DPP_CHUNKS"] = "1"
    else:
        os.environ.pop("GRAPE_DPP_CHUNKS", None)

    def vec(m=1):
        v = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
        return v / np.linalg.norm(v)
    if sys_type == "UnitaryGate":
        if states in ("rect", "vec"):
            m = int(rng.integers(1, n)) if states == "rect" else 1
            Xi = np.array([vec(m) for _ in range(E)])
            Xt = np.array([vec(m) for _ in range(E)])
        else:
            Xi = np.array([np.eye(n, dtype=complex)] * E)
            Xt = np.array([np.linalg.qr(mat(False))[0] for _ in range(E)])
    else:
        def rho():
            if states == "mixed":
                return sum(p * (lambda v: v @ v.conj().T)(vec()) for p in (0.5, 0.3, 0.2))
            v = vec()
            return v @ v.conj().T
        Xi = np.array([rho() for _ in range(E)])
        Xt = np.array([rho() for _ in range(E)])
    wts = rng.uniform(0.2, 1.0, E)
    x = rng.uniform(-1, 1, (K, N))
    T = float(rng.uniform(0.3, 2.0))
    what = (f"case {i}: n={n} K={K} N={N} E={E} {sys_type} v{variant} herm={herm} sparse={sparse} states={states} flag={flag} "
            f"shared_ctrl={shared_ctrl} action={os.environ.get('GRAPE_ACTION', '-')} whole={os.environ.get('GRAPE_ACT_WHOLE', '-')} dpp={os.environ.get('GRAPE_THIN_DPP', '-')} dppc={os.environ.get('GRAPE_DPP_CHUNKS', '-')}")
    exact = rng.random() < 0.15 and N <= 33 and n <= 64 and states not in ("rect", "vec")      # (the C oracle has no exact gradient for n x m states)
    if chunk_budget:
        nt = (n + 15) // 16
        # one member's share of one workspace array: n <= 4 S x chunks <= N + 512 slices' worth of n x n matrices; tiles: N dumps
        unit = (N + 512) * 16 * n * n if n <= 4 else N * nt * nt * 256 * 16
        os.environ["GRAPE_MAX_WORKSPACE_BYTES"] = str(int(unit * (max(8, E // 2) if n <= 4 else 3 * max(2, E // 2))))
    else:
        os.environ.pop("GRAPE_MAX_WORKSPACE_BYTES", None)
    if exact:                                             # exact gradient of the figure of merit / of the C1 functional
        objective = int(rng.integers(0, 2))
        try:
            F_ref, G_ref = orc.ensemble_exact(sys_type, A, B, Xi, Xt, wts, x, T, variant=variant, objective=objective)
            with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, T, N, variant=variant, gradient="exact",
                                 objective="c1" if objective else "fom") as eng:
                Fv, G = eng.eval(x)
            gs = max(np.abs(G_ref).max(), 1e-6)
            assert abs(Fv - F_ref) <= 1e-10 * max(abs(F_ref), 1e-3 * n * n) and np.abs(G - G_ref).max() <= 1e-10 * gs, \
                (what + f" exact objective={objective}", Fv, F_ref, np.abs(G - G_ref).max() / gs)
        except Exception as exc:                          # noqa: BLE001
            fails += 1
            print("FAIL", what, f"exact objective={objective}", "->", repr(exc)[:300], flush=True)
        continue
    try:
        F_ref, G_ref, foms_ref, grads_ref = orc.ensemble_eval(sys_type, A, B, Xi, Xt, wts, x, T, variant=variant, per_member=True)
        with qoc.GrapeEngine(sys_type, A, B, Xi, Xt, wts, T, N, variant=variant, flags=flag, member_results=True) as eng:
            Fv, G = eng.eval(x)
            foms, grads = eng.member_results()
            info = eng.info
        n_action += int(info.get("expm_action", 0))
        n_chunked += int(info.get("prop_chain", 0) and info.get("time_chunks", 0) >= 2)
        n_grid += int(n > 32)
        n_member_chunked += int(info.get("member_chunk", E) < E)
        for k in range(E):
            # a member's whole gradient can be a near-zero (K = 1, N = 1: one entry passing through zero): the norm-wise
            # bar then has no scale left, so an absolute floor of a few ulp of the traces applies -- O(1) for unitary
            # propagators, O(|F_k|) for the non-normal ones whose overlaps exceed 1 (seed 32, case 1233: one slice with
            # |dt H| = 6.6, cond(P) = 104, |tr(Xt' X_N)| = 17, F_k = -11: |G - G_ref| = 3.2e-13 on terms of size 300)
            if (np.abs(grads[k] - grads_ref[k]).max() <= 5e-14 * n * max(1.0, abs(foms_ref[k])) and
                    abs(foms[k] - foms_ref[k]) <= 1e-10 * max(abs(foms_ref[k]), 1e-3 * n * n)):
                continue
            assert_parity(foms[k], grads[k], foms_ref[k], grads_ref[k], n, what=what + f" member {k}")
        # ensemble: members' figures of merit can cancel in the weighted sum (UnitaryGate F_k = Re(z^2) has either sign),
        # so the bar for F is taken relative to sum w_k |F_k|, the scale the members' own 1e-10 errors add up on
        scale = float(np.abs(foms_ref) @ wts)
        assert abs(Fv - F_ref) <= 1e-10 * max(scale, 1e-3 * n * n), (what, Fv, F_ref, scale)
        if np.abs(G - G_ref).max() > 5e-14 * n * max(1.0, scale):
            assert_parity(F_ref, G, F_ref, G_ref, n, what=what)
    except Exception as exc:                          # noqa: BLE001
        fails += 1
        print("FAIL", what, "->", repr(exc)[:300], flush=True)
print(f"soak: {cases} cases, {fails} failures, {n_action} of them through the vector flow, {n_chunked} through the chunked propagator chain, "
      f"{n_grid} with n > 32 (grid family; n > 64: size-generic kernel on the matrix cores), {n_member_chunked} member-chunked, "
      f"{n_scaled} with scaled per-member controls, {time.time() - t0:.1f} s (seed {seed})")
sys.exit(1 if fails else 0)
