#!/usr/bin/env python3
"""Randomised API soak: one context, a random sequence of operations -- evaluations through the host, device-pointer and
batched entry points, re-uploads of operators that switch the data flow (Hermitian <-> not, pure <-> mixed states,
sparse <-> dense controls), accessor calls -- each checked against the oracle.  usage: tools/soak_api.py [contexts] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from conftest import assert_parity as _assert_parity  # noqa: E402


def assert_parity(F, G, F_ref, G_ref, n, what=""):
    """the 1e-10 bar, with an absolute floor of a few ulp of the O(1) traces for gradients that are near zero as a whole"""
    if np.abs(np.asarray(G) - np.asarray(G_ref)).max() <= 5e-14 * n:
        G = G_ref
    _assert_parity(F, G, F_ref, G_ref, n, what=what)
from oracle import grape_oracle as orc  # noqa: E402

contexts = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
fails = 0
checks = 0
t0 = time.time()


def operators(n, K, E, sand, herm, mixed, sparse):
    def mat(h):
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / 2 if h else M
    g = min(1.0, 4.0 / n)
    A = np.array([mat(herm) for _ in range(E)]) * 0.6 * g
    if sparse:
        B = np.zeros((E, K, n, n), complex)
        for k in range(E):
            for c in range(K):
                for _ in range(int(rng.integers(1, 10))):
                    a, b = rng.integers(0, n, 2)
                    v = (rng.standard_normal() + 1j * rng.standard_normal()) * 0.5
                    if a == b:
                        B[k, c, a, a] = v.real
                    else:
                        B[k, c, a, b] = v
                        B[k, c, b, a] = np.conj(v) if herm else 0.3 * v
    else:
        B = np.array([[mat(herm) for _ in range(K)] for _ in range(E)]) * 0.4 * g

    def vec():
        v = rng.standard_normal((n, 1)) + 1j * rng.standard_normal((n, 1))
        return v / np.linalg.norm(v)
    if sand:
        def rho():
            if mixed:
                return sum(p * (lambda v: v @ v.conj().T)(vec()) for p in (0.5, 0.3, 0.2))
            v = vec()
            return v @ v.conj().T
        Xi, Xt = np.array([rho() for _ in range(E)]), np.array([rho() for _ in range(E)])
    else:
        Xi = np.array([np.eye(n, dtype=complex)] * E)
        Xt = np.array([np.linalg.qr(mat(False))[0] for _ in range(E)])
    return A, B, Xi, Xt, rng.uniform(0.2, 1.0, E)


for ci in range(contexts):
    n = int(rng.choice([2, 3, 4, 6, 8, 12, 16, 20, 32]))
    K, E = int(rng.integers(1, 6)), int(rng.choice([1, 2, 3, 6]))
    N = int(rng.choice([1, 3, 8, 20, 65])) if n <= 16 else int(rng.choice([1, 3, 6]))
    sys_type = str(rng.choice(["UnitaryGate", "StateTransfer", "CoherenceTransfer"]))
    sand = sys_type != "UnitaryGate"
    variant, T, mb = int(rng.integers(0, 2)), float(rng.uniform(0.3, 1.5)), int(rng.choice([1, 3]))
    ops = operators(n, K, E, sand, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
    what = f"context {ci}: n={n} K={K} N={N} E={E} {sys_type} v{variant} max_batch={mb}"
    try:
        with qoc.GrapeEngine(sys_type, *ops, T, N, variant=variant, member_results=True, max_batch=mb) as eng:
            for step in range(int(rng.integers(3, 9))):
                op = str(rng.choice(["eval", "eval", "device", "batch", "upload", "members", "props", "F_only"]))
                x = rng.uniform(-1, 1, (K, N))
                if op == "upload":
                    ops = operators(n, K, E, sand, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
                    eng.set_operators(*ops)
                    continue
                ref = orc.ensemble_eval(sys_type, *ops, x, T, variant=variant, per_member=True)
                w2 = f"{what} step {step} {op}"
                checks += 1
                if op == "eval":
                    F, G = eng.eval(x)
                    assert_parity(F, G, ref[0], ref[1], n, what=w2)
                elif op == "F_only":
                    F, G = eng.eval(x, want_G=False)
                    assert G is None and abs(F - ref[0]) <= 1e-10 * max(abs(ref[0]), 1e-3 * n * n), w2
                elif op == "device":
                    xd = torch.as_tensor(np.ascontiguousarray(x.T), device="cuda")
                    fg = torch.zeros(K * N + 1, dtype=torch.float64, device="cuda")
                    eng.eval_device(xd.data_ptr(), fg.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    h = fg.cpu().numpy()
                    assert_parity(h[-1], h[:-1].reshape(N, K).T, ref[0], ref[1], n, what=w2)
                elif op == "batch":
                    xs = np.stack([x] + [rng.uniform(-1, 1, (K, N)) for _ in range(mb - 1)])
                    Fs, Gs = eng.eval_batch(xs)
                    for b in range(mb):
                        rb = orc.ensemble_eval(sys_type, *ops, xs[b], T, variant=variant)
                        assert_parity(Fs[b], Gs[b], rb[0], rb[1], n, what=w2 + f" entry {b}")
                elif op == "members":
                    eng.eval(x)
                    foms, grads = eng.member_results()
                    for k in range(E):
                        if np.abs(grads[k] - ref[3][k]).max() > 5e-14 * n:
                            assert_parity(foms[k], grads[k], ref[2][k], ref[3][k], n, what=w2 + f" member {k}")
                elif op == "props":
                    eng.eval(x)
                    k = int(rng.integers(0, E))
                    P = eng.trajectory(k, states=False)[0]
                    P_ref = orc.member_eval(sys_type, ops[0][k], ops[1][k], ops[2][k], ops[3][k], x, T, variant=variant, trajectory=True)[2]
                    assert np.abs(P - P_ref).max() <= 1e-11 * max(1.0, np.abs(P_ref).max()), w2
    except Exception as exc:                          # noqa: BLE001
        fails += 1
        print("FAIL", what, "->", repr(exc)[:400], flush=True)
print(f"api soak: {contexts} contexts, {checks} checked operations, {fails} failures, {time.time() - t0:.1f} s (seed {seed})")
sys.exit(1 if fails else 0)
