#!/usr/bin/env python3
"""n x 1 states at n = 4 (vec(rho) of ONE qubit under a Liouvillian, test/liou.jl:38-48): ms per grape_eval for E members,
N slices -- the zero-padded 4 x 4 run (GRAPE_VEC4=0) against the native chain.   usage: tools/vec4_time.py [E] [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
for diss in (True, False):
    w = qoc.workloads.liouville_vec(nq=1, E=E, N=N, T=5.0, dissipative=diss)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
        xf = np.ascontiguousarray(w.x.T)
        call = eng.bind_eval(xf, np.empty_like(xf))
        for _ in range(50):
            call()
        t0 = time.perf_counter()
        for _ in range(300):
            call()
        ms = (time.perf_counter() - t0) / 300 * 1e3
        print(f"vec4x1 E={E} N={N} dissipative={diss}: {ms:.4f} ms per evaluation; kernels {eng.kernel_names()}; "
              f"unitary_flow={eng.info['unitary_flow']} S={eng.info['slices_per_lane']} W={eng.info['waves_per_member']}")
