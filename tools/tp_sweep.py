#!/usr/bin/env python3
"""Chunk-count sweep of the time-parallel unitary chain (GRAPE_TP_CHUNKS) against the library's own choice.
usage: tools/tp_sweep.py n N E [chunk counts ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

n, N, E = (int(a) for a in sys.argv[1:4])
counts = [int(a) for a in sys.argv[4:]]
K = 4
rng = np.random.default_rng(3)


def herm(n):
    M = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    return (M + M.conj().T) / 2


A = np.array([herm(n) for _ in range(E)]) * 0.3
B = np.array([[herm(n) for _ in range(K)] for _ in range(E)]) * 0.2
Xi = np.array([np.eye(n, dtype=complex) for _ in range(E)])
x = rng.uniform(-1, 1, (K, N))
for cnt in [0] + counts:
    if cnt:
        os.environ["GRAPE_TP_CHUNKS"] = str(cnt)
    with qoc.GrapeEngine("UnitaryGate", A, B, Xi, Xi.copy(), np.ones(E) / E, 2.0, N) as eng:
        C = eng.info["time_chunks"]
        for _ in range(5):
            eng.eval(x)
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            eng.eval(x)
        dt = (time.perf_counter() - t0) / reps
    print(f"n={n} N={N} E={E} chunks={C:4d} ({'auto' if not cnt else 'forced'})  {dt * 1e3:8.3f} ms", flush=True)
