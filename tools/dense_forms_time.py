#!/usr/bin/env python3
"""Rank-one 16 x 16 problems (C4's drifts and states, N = 1000) with DENSE shared control operators: the forms kernel on
the matrix cores (default) against the vector-ALU one (GRAPE_FORMS_VALU=1), and for small ensembles the chunked propagator
chain (GRAPE_DPP_CHUNKS=1) against the flows the library picks for dense controls.
usage: tools/dense_forms_time.py [E ...]   (run once per GRAPE_FORMS_VALU setting: it is read once per process)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

rng = np.random.default_rng(4)
for E in [int(a) for a in sys.argv[1:]] or [1, 8, 128, 1024]:
    w = qoc.workloads.config("C4", E=E)
    n, K = w.n, w.K
    B0 = np.array([(lambda M: 0.1 * (M + M.conj().T))(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) for _ in range(K)])
    B = np.array([B0] * E)
    for mode in (["1", "0"] if E < 41 else [None]):
        if mode is None:
            os.environ.pop("GRAPE_DPP_CHUNKS", None)
        else:
            os.environ["GRAPE_DPP_CHUNKS"] = mode
        with qoc.GrapeEngine(w.sys_type, w.A, B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
            for _ in range(30):
                eng.eval(w.x)
            reps = 300 if E < 200 else 50
            t0 = time.perf_counter()
            for _ in range(reps):
                F, G = eng.eval(w.x)
            dt = (time.perf_counter() - t0) / reps
            info = eng.info
        print(f"dense controls E={E:4d} DPP_CHUNKS={mode} FORMS_VALU={os.environ.get('GRAPE_FORMS_VALU', '-')} {dt * 1e3:8.4f} ms "
              f"prop_chain={info['prop_chain']} action={info['expm_action']} chunks={info['time_chunks']} F={F:.12f}", flush=True)
