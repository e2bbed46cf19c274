#!/usr/bin/env python3
"""Diagnostic: sweep-kernel time (HIP events) and phase shares (stamp build) of one config, one line.
usage: [GRAPE_HIP_LIB=...] tools/ktime.py [--config C3] [--ensemble E] [--waves-per-member W] [--slices-per-lane S]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--ensemble", type=int, default=0)
ap.add_argument("--slices-per-lane", type=int, default=0)
ap.add_argument("--waves-per-member", type=int, default=0)
ap.add_argument("--force-general", action="store_true")
ap.add_argument("--evals", type=int, default=60)
ap.add_argument("--tag", default="")
a = ap.parse_args()
w = qoc.workloads.config(a.config, E=a.ensemble or None)
fl = qoc.engine.FLAG_FORCE_GENERAL if a.force_general else 0
kw = dict(slices_per_lane=a.slices_per_lane, waves_per_member=a.waves_per_member)
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=fl | qoc.engine.FLAG_TIME_KERNELS, **kw) as eng:
    for _ in range(10):
        eng.eval(w.x)
    eng.kernel_time(reset=True)
    for _ in range(a.evals):
        eng.eval(w.x)
    ms, n = eng.kernel_time()
    info = eng.info
line = f"{a.tag or os.environ.get('GRAPE_HIP_LIB', 'default')[-24:]:>26s} kernel {1e3 * ms / n:7.2f} us  S={info['slices_per_lane']} W={info['waves_per_member']}"
if info["kernel_family"] == 0:
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=fl | qoc.engine.FLAG_PHASE_STAMPS, **kw) as eng:
        for _ in range(3):
            eng.eval(w.x)
        st = eng.phase_stamps().astype(np.int64)
    d = np.diff(st[:, :5], axis=1)
    tot = st[:, 4] - st[:, 0]
    line += ("  stamps: total %6.0f  A %6.0f [%6.0f..%6.0f]  B %6.0f [min %6.0f]  D %6.0f  span %5.1f us" %
             (np.median(tot), np.median(d[:, 0]), np.percentile(d[:, 0], 10), np.percentile(d[:, 0], 90),
              np.median(d[:, 1]), d[:, 1].min(), np.median(d[:, 3]), (st[:, 6].max() - st[:, 5].min()) * 0.01))
print(line)
