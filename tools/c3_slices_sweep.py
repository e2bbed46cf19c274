#!/usr/bin/env python3
"""C3's kernel over the number of slices N (the propagator workspace grows with it: 33 .. 524 MB for 1024 members): kernel time
per slice and the phase medians -- does the backward sweep's P_t stream get faster while the workspace fits the 256 MB
Infinity Cache, and what does a longer time axis per lane change?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

for N in (125, 250, 500, 1000, 2000, 4000):
    w = qoc.workloads.config("C3", N=N)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_PHASE_STAMPS) as eng:
        for _ in range(5):
            eng.eval(w.x)
        st = eng.phase_stamps().astype(np.int64)
        info = eng.info
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
        for _ in range(300):
            eng.eval(w.x)
        eng.kernel_time(reset=True)
        for _ in range(200):
            eng.eval(w.x)
        ms, cnt = eng.kernel_time()
    d = np.diff(st[:, :5], axis=1)
    tot = st[:, 4] - st[:, 0]
    S = info["slices_per_lane"]
    q = lambda a: "/".join(f"{v:.0f}" for v in np.percentile(a, [10, 50, 90]))
    print(f"N={N:5d} S={S:2d} W={info['waves_per_member']} waves {st.shape[0]} P_t {1024 * N * 256 / 1e6:6.1f} MB: kernels {1e3 * ms / cnt:7.1f} us "
          f"= {1e6 * ms / cnt / N:6.1f} ns per slice;  cycles (10/50/90 %) A {q(d[:, 0])}  B {q(d[:, 1])}  D {q(d[:, 3])}  wave {q(tot)}")
