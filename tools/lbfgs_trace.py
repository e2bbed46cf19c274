#!/usr/bin/env python3
"""One device-resident L-BFGS run on the C3-shaped StateTransfer ensemble (bench.py's optimiser problem), for
`rocprofv3 --kernel-trace --stats -- python3 tools/lbfgs_trace.py`: where an iteration's time goes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

w = qoc.workloads.config("C3")
rho0 = np.zeros((4, 4), complex); rho0[0, 0] = 1
psi = np.array([1, 1j, -1, 0.5]) / np.linalg.norm([1, 1j, -1, 0.5])
Xi = np.broadcast_to(rho0, (w.E, 4, 4)).copy()
Xt = np.broadcast_to(np.outer(psi, psi.conj()), (w.E, 4, 4)).copy()
with qoc.GrapeEngine("StateTransfer", w.A, w.B, Xi, Xt, w.wts, w.T, w.N) as eng:
    for _ in range(5):
        x, info = eng.lbfgs(w.x, iterations=30)
    print(info)
