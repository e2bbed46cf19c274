#!/usr/bin/env python3
"""One 16 x 16 open-system problem (C4's operators, full-rank path forced): for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

w = qoc.workloads.config("C4", E=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_FORCE_GENERAL) as eng:
    for _ in range(200):
        eng.eval(w.x)
