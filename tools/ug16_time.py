#!/usr/bin/env python3
"""Four-qubit gate synthesis with a robustness ensemble (16 x 16 UnitaryGate, Ising drift with detuned members, single-qubit
X / Y controls, N = 1000): evaluation time and its expm / chain split by ensemble size.  GRAPE_TP_SLOTS=4 restores the
chunking rule of rounds 1-2 (time chunks only below 2 x CUs units).  usage: tools/ug16_time.py [E ...]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import quoptimalcontrol_jl_amd as qoc
rng = np.random.default_rng(1)
X = np.array([[0, 1], [1, 0]], complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0, -1.0]).astype(complex); I2 = np.eye(2)
def on(op, q, nq):
    M = np.array([[1.0 + 0j]])
    for i in range(nq):
        M = np.kron(M, op if i == q else I2)
    return M
def problem(nq, K, N, E):
    n = 2 ** nq
    H0 = sum(on(Z, q, nq) @ on(Z, (q + 1) % nq, nq) for q in range(nq)) * 0.5
    A = np.array([H0 + 0.05 * rng.standard_normal() * sum(on(Z, q, nq) for q in range(nq)) for _ in range(E)])
    B0 = ([on(X, q, nq) for q in range(nq)] + [on(Y, q, nq) for q in range(nq)])[:K]
    B = np.array([B0] * E)
    Xi = np.array([np.eye(n, dtype=complex)] * E)
    U = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
    return A, B, Xi, np.array([U] * E), rng.uniform(-1, 1, (K, N))
for E in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    A, B, Xi, Xt, x = problem(4, 4, 1000, E)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, np.full(E, 1.0 / E), 5.0, 1000, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
        for _ in range(3): eng.eval(x)
        eng.kernel_time(reset=True)
        t0 = time.perf_counter(); steps = 10
        for _ in range(steps): F, G = eng.eval(x)
        dt = (time.perf_counter() - t0) / steps
        tot, first = eng.kernel_samples()
        info = eng.info
    print(f"UnitaryGate 16x16 K=4 N=1000 E={E:5d} env TP={os.environ.get('GRAPE_TP_CHUNKS','-')}: {dt*1e3:8.3f} ms/eval {dt/E*1e6:7.3f} us/member  expm {np.mean(first):6.3f} ms chain {np.mean(tot)-np.mean(first):6.3f} ms chunks={info['time_chunks']}", flush=True)
