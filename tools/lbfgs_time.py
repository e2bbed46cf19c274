#!/usr/bin/env python3
"""grape_lbfgs on the bench's C3-shaped StateTransfer problem (4x4, K=4, N=500, E=1024): wall time per iteration and, under
rocprofv3 --kernel-trace --stats, the step kernel's duration.  usage: tools/lbfgs_time.py [iterations] [line_search]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else "hagerzhang"
w = qoc.workloads.config("C3")
rho0 = np.zeros((4, 4), complex); rho0[0, 0] = 1
psi = np.array([1, 1j, -1, 0.5]) / np.linalg.norm([1, 1j, -1, 0.5])
Xi = np.broadcast_to(rho0, (w.E, 4, 4)).copy()
Xt = np.broadcast_to(np.outer(psi, psi.conj()), (w.E, 4, 4)).copy()
with qoc.GrapeEngine("StateTransfer", w.A, w.B, Xi, Xt, w.wts, w.T, w.N) as eng:
    eng.lbfgs(w.x, iterations=3, line_search=mode)
    best = None
    for rep in range(5):
        t0 = time.perf_counter()
        x, info = eng.lbfgs(w.x, iterations=iters, line_search=mode)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, info)
    dt, info = best
    t0 = time.perf_counter()
    for _ in range(200):
        eng.eval(w.x)
    ev = (time.perf_counter() - t0) / 200
print(f"{mode}: {info['iterations']} iterations, {info['evaluations']} evaluations, {dt * 1e3:.3f} ms, minimum {info['minimum']!r}, "
      f"bare eval {ev * 1e6:.1f} us, per evaluation {dt / info['evaluations'] * 1e6:.1f} us, "
      f"per iteration beyond its evaluations {(dt - info['evaluations'] * ev) / max(1, info['iterations']) * 1e6:.1f} us")
