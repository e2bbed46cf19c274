#!/usr/bin/env python3
"""Small ensembles of rank-one 16 x 16 problems (C4's operators, N = 1000): the chunked propagator chain of action_thin.hip
(GRAPE_DPP_CHUNKS unset) against the flows it replaced (=0).  usage: tools/dpp_chunks_time.py [E ...] [chunks=C]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("chunks=")]
chunk_opts = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("chunks=")] or [""]
for E in [int(a) for a in args] or [1, 4, 16, 32, 64, 79]:
    w = qoc.workloads.config("C4", E=E)
    for mode in ("1", "0"):
        for ch in (chunk_opts if mode == "1" else [""]):
            os.environ["GRAPE_DPP_CHUNKS"] = mode
            os.environ.pop("GRAPE_TP_CHUNKS", None)
            if ch:
                os.environ["GRAPE_TP_CHUNKS"] = ch
            with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N) as eng:
                for _ in range(50):
                    eng.eval(w.x)
                t0 = time.perf_counter()
                reps = 300
                for _ in range(reps):
                    F, G = eng.eval(w.x)
                dt = (time.perf_counter() - t0) / reps
                info = eng.info
            print(f"C4 E={E:3d} DPP_CHUNKS={mode} {dt * 1e3:8.4f} ms prop_chain={info['prop_chain']} chunks={info['time_chunks']}"
                  f" F={F:.12f}", flush=True)
