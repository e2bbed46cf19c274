#!/usr/bin/env python3
"""64 x 1 states (vec(rho) of THREE qubits under a Liouvillian, test/liou.jl:38-48 two sizes up; K = 6 Pauli-type control
superoperators): ms per grape_eval for E members, N slices -- the vector chain of sweep_grid.hip (grid_thin_kernel) against
the dense chain of the zero-padded states (GRAPE_NO_THIN=1), kernel by kernel.   usage: tools/vec64_time.py [E] [N]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
w = qoc.workloads.liouville_vec(nq=3, E=E, N=N, T=2.5, dissipative=True)
res = {}
for thin in (True, False):
    if thin:
        os.environ.pop("GRAPE_NO_THIN", None)
    else:
        os.environ["GRAPE_NO_THIN"] = "1"
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
        xf = np.ascontiguousarray(w.x.T)
        g = np.empty_like(xf)
        call = eng.bind_eval(xf, g)
        for _ in range(5):
            F = call()
        eng.kernel_time(reset=True)
        t0 = time.perf_counter()
        reps = 30
        for _ in range(reps):
            call()
        ms = (time.perf_counter() - t0) / reps * 1e3
        tot, first = (float(np.median(v)) for v in eng.kernel_samples())
        res[thin] = (F, g.copy())
        print(f"vec64x1 E={E} N={N} rank_one_chain={eng.info['rank_one_chain']}: {ms:.3f} ms per evaluation "
              f"(kernels {tot:.3f} ms, expm part {first:.3f} ms); {eng.kernel_names()}")
dF = abs(res[True][0] - res[False][0])
dG = np.abs(res[True][1] - res[False][1]).max() / max(1e-300, np.abs(res[False][1]).max())
print(f"vector chain vs dense chain: |dF| = {dF:.2e}, max |dG| / max |G| = {dG:.2e}")
