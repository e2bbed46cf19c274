#!/usr/bin/env python3
"""Single problems (the `Problem` closure, src/solve.jl:63-143) at the sizes that have no chunked time axis: 64 x 64 (C6's shape,
sweep_grid.hip) and 128 x 128 (C7's shape, sweep_any.hip) -- ms per grape_eval for E = 1, 4, 16 members, next to the per-member
cost of the bench's ensembles (C6: 23.0 ms / 256, C7: ~100 ms / 64).  The chain of a member walks its N slices sequentially in
ONE workgroup at these sizes (n <= 32 cuts the time axis into parallel chunks: C5x1 0.16 ms).   usage: tools/single_big_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

for cfg in ("C6", "C7"):
    for E in (1, 4, 16):
        w = qoc.workloads.config(cfg, E=E)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_TIME_KERNELS) as eng:
            xf = np.ascontiguousarray(w.x.T)
            call = eng.bind_eval(xf, np.empty_like(xf))
            for _ in range(3):
                call()
            eng.kernel_time(reset=True)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            ms = (time.perf_counter() - t0) / reps * 1e3
            tot, first = (float(np.median(v)) if len(v) else 0.0 for v in eng.kernel_samples())
            print(f"{cfg} n={w.n} K={w.K} N={w.N} E={E:3d}: {ms:8.3f} ms per evaluation (first kernel part {first:.3f} ms, rest {tot - first:.3f} ms) "
                  f"= {ms / E:7.3f} ms per member | {';'.join(eng.kernel_names())}")
