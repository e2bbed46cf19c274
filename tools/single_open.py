#!/usr/bin/env python3
"""Single / small-ensemble OPEN-system problems (16 x 16 Liouvillians, C4's operators): rank-one states (vector chain)
and full-rank states (dense general flow).  usage: tools/single_open.py [E ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

for E in [int(a) for a in sys.argv[1:]] or [1]:
    w = qoc.workloads.config("C4", E=E)
    for label, flags in (("rank-one chain", 0), ("dense general flow", qoc.engine.FLAG_FORCE_GENERAL)):
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=flags) as eng:
            for _ in range(8):
                eng.eval(w.x)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                eng.eval(w.x)
            dt = (time.perf_counter() - t0) / reps
            info = eng.info
        print(f"C4 shape E={E:4d} {label:20s} {dt * 1e3:8.3f} ms  rank_one={info['rank_one_chain']} chunks={info['time_chunks']}", flush=True)
