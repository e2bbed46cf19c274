#!/usr/bin/env python3
"""Single-problem (E = 1) and small-ensemble evaluation times in the tile family (n = 5..32): host -> host grape_eval.
usage: tools/single_time.py [E ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

Es = [int(a) for a in sys.argv[1:]] or [1]
rng = np.random.default_rng(3)


def herm(n):
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    return (M + M.conj().T) / 2


for n, K, N in [(8, 4, 500), (16, 4, 1000), (32, 6, 2000)]:
    for E in Es:
        A = np.array([herm(n) for _ in range(E)]) * 0.3
        B = np.array([[herm(n) for _ in range(K)] for _ in range(E)]) * 0.2
        Xi = np.array([np.eye(n, dtype=complex) for _ in range(E)])
        Q = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))[0]
        Xt = np.array([Q for _ in range(E)])
        x = rng.uniform(-1, 1, (K, N))
        with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, np.ones(E) / E, 2.0, N, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
            for _ in range(12):                                       # (large workspaces are still being paged in)
                eng.eval(x)
            eng.kernel_time(reset=True)
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                eng.eval(x)
            dt = (time.perf_counter() - t0) / reps
            ms, cnt = eng.kernel_time()
            info = eng.info
        print(f"n={n:2d} K={K} N={N:4d} E={E:4d}  call {dt * 1e3:8.3f} ms  kernels {ms / max(cnt, 1):8.3f} ms  unitary={info['unitary_flow']}", flush=True)
