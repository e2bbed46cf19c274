#!/usr/bin/env python3
"""One small-ensemble 32 x 32 problem, a few evaluations: for rocprofv3 --kernel-trace --stats (diagnostic)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, K, N = 32, 6, 2000
rng = np.random.default_rng(3)


def herm(n):
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    return (M + M.conj().T) / 2


A = np.array([herm(n) for _ in range(E)]) * 0.3
B = np.array([[herm(n) for _ in range(K)] for _ in range(E)]) * 0.2
Xi = np.array([np.eye(n, dtype=complex) for _ in range(E)])
Xt = Xi.copy()
x = rng.uniform(-1, 1, (K, N))
with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xt, np.ones(E) / E, 2.0, N) as eng:
    print(eng.info["time_chunks"], eng.info["slices_per_lane"])
    for _ in range(3):
        eng.eval(x)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.eval(x)
    print("call ms", (time.perf_counter() - t0) / 5 * 1e3)
