#!/usr/bin/env python3
"""Per-member evaluation cost over operator sizes, system types and ensemble sizes (Hermitian generators, sparse Pauli-like
controls for n a power of two, dense otherwise; full-rank / pure states): a table to spot sizes that fall off their
neighbours (a launch with one wave per SIMD, a flow threshold in the wrong place).  usage: tools/shape_sweep.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from test_gpu_tile import _random_problem  # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [4, 8, 12, 16, 24, 32]
for n in ns:
    N = 500 if n <= 16 else 300
    for sys_type, mixed in (("UnitaryGate", True), ("StateTransfer", True), ("StateTransfer", False)):
        row = []
        for E in (1, 16, 128, 512, 1024, 2048, 4096):
            if n > 16 and E > 1024:
                continue
            w = _random_problem(qoc, n, 4, N, E, sys_type, seed=n + E, mixed=mixed)
            w.B[:] = w.B[0]                                 # shared controls
            with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
                for _ in range(3):
                    eng.eval(w.x)
                reps = 20 if E <= 128 else 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.eval(w.x)
                dt = (time.perf_counter() - t0) / reps
                info = eng.info
            flow = "A" if info["expm_action"] else ("P" if info.get("prop_chain") else ("T" if info["rank_one_chain"] else ("U" if info["unitary_flow"] else "G")))
            row.append(f"E={E}: {dt * 1e3:7.3f} ms {dt / E * 1e6:6.2f} us/mem {flow}{info['time_chunks']}")
        print(f"n={n:2d} {sys_type:13s} {'mixed' if mixed else 'pure '} | " + " | ".join(row), flush=True)
