#!/usr/bin/env python3
"""Lane-pair kernel (n = 4): ms per evaluation over the waves per member (0 = the library's own choice) for C3-shaped
ensembles larger / longer than the headline -- the measurement behind the "no more than 16 slices per pair" rule of
grape_create (round 6).   usage: tools/w_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import quoptimalcontrol_jl_amd as qoc
for E, N in ((2048, 1000), (4096, 1000), (1024, 2000), (4096, 2000), (1024, 1500), (3000, 700)):
    w = qoc.workloads.config("C3", E=E, N=N)
    for W in (0, 2, 4, 8):
        try:
            with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, waves_per_member=W) as eng:
                xf = np.ascontiguousarray(w.x.T)
                call = eng.bind_eval(xf, np.empty_like(xf))
                for _ in range(10): call()
                t0 = time.perf_counter()
                for _ in range(60): call()
                ms = (time.perf_counter() - t0) / 60 * 1e3
                print(f"C3 E={E} N={N} W_req={W}: {ms:.4f} ms; S={eng.info['slices_per_lane']} W={eng.info['waves_per_member']} uni={eng.info['unitary_flow']}")
        except Exception as e:
            print("W", W, "failed", repr(e)[:100])
