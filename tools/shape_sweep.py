#!/usr/bin/env python3
"""Throughput over problem shapes away from the BASELINE configs (random Hermitian UnitaryGate ensembles, shared controls):
ms per evaluation and member-slices per microsecond for n x E x N grids -- to spot decomposition cliffs (round 6: the lane
kernels' slices-per-lane rule came out of such a sweep).   usage: tools/shape_sweep.py [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

rng = np.random.default_rng(7)


def prob(n, E, N, K=3):
    def herm():
        M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        return (M + M.conj().T) / (2 * np.sqrt(n))
    A = np.array([herm() for _ in range(E)])
    B = np.broadcast_to(np.array([herm() for _ in range(K)]), (E, K, n, n)).copy()
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    Xi = np.broadcast_to(np.eye(n, dtype=complex), (E, n, n)).copy()
    Xt = np.broadcast_to(q, (E, n, n)).copy()
    return A, B, Xi, Xt, np.full(E, 1.0 / E), rng.uniform(-1, 1, (K, N))


ns = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 48, 64]
for n in ns:
    for E in (16, 64, 256, 1024, 4096):
        for N in (50, 500, 2000):
            if E * N * n * n * 16 > 6e9:
                continue
            A, B, Xi, Xt, wts, x = prob(n, E, N)
            with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, wts, 2.0, N) as eng:
                xf = np.ascontiguousarray(x.T)
                call = eng.bind_eval(xf, np.empty_like(xf))
                call()
                reps = max(2, min(50, int(0.3 / max(1e-4, 2e-9 * E * N * n ** 3 / 1e3))))
                t0 = time.perf_counter()
                for _ in range(reps):
                    call()
                ms = (time.perf_counter() - t0) / reps * 1e3
                names = eng.kernel_names()
            print(f"n={n:3d} E={E:5d} N={N:5d}: {ms:10.4f} ms  {E * N / ms / 1e3:9.2f} member-slices/us  {n ** 3 * E * N * 8 * 7 / ms / 1e9:7.2f} TF(7 products)  | {';'.join(names)[:110]}")
