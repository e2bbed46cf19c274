#!/usr/bin/env python3
"""n x 1 states at n = 32 (the five-qubit operators of C5 acting on one state vector per member): the vector flow of
csrc/action_thin.hip against the zero-padded dense chains (GRAPE_ACTION=0).  usage: tools/vec32_bench.py [E] [N]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from quoptimalcontrol_jl_amd import workloads  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = workloads.config("C5", E=E, N=N)
rng = np.random.default_rng(0)
v0 = rng.standard_normal((32, 1)) + 1j * rng.standard_normal((32, 1))
v0 /= np.linalg.norm(v0)
Xi = np.repeat(v0[None], E, axis=0)
Xt = w.Xt[:, :, :1] @ np.ones((1, 1))            # first column of the target unitary applied to e_0 ... any unit vector
out = {"E": E, "N": N}
for mode in ("1", "0"):
    os.environ["GRAPE_ACTION"] = mode
    with qoc.GrapeEngine("UnitaryGate", w.A, w.B, Xi, Xt, w.wts, w.T, N) as eng:
        info = eng.info
        F, G = eng.eval(w.x)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 0.5 or reps < 3:
            eng.eval(w.x)
            reps += 1
        t0 = time.perf_counter()
        for _ in range(reps):
            F, G = eng.eval(w.x)
        ms = (time.perf_counter() - t0) / reps * 1e3
    out["vector flow" if mode == "1" else "padded dense chains"] = {"ms_per_evaluation": ms, "expm_action": info["expm_action"],
                                                                     "F": F, "G_norm": float(np.linalg.norm(G))}
out["speedup"] = out["padded dense chains"]["ms_per_evaluation"] / out["vector flow"]["ms_per_evaluation"]
print(json.dumps(out))
