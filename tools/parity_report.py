#!/usr/bin/env python3
"""Parity evidence: HIP path vs the CPU oracle on the BASELINE configs (at sizes the oracle finishes
in seconds) and vs the mpmath golden fixtures; prints one JSON document (profiles/r01_parity.json)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402
from oracle import grape_oracle  # noqa: E402
from test_oracle_golden import load_case  # noqa: E402

rows = []


def compare(label, w, variant=0, **ekw):
    F_ref, G_ref, foms_ref, grads_ref = grape_oracle.ensemble_eval(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.x, w.T,
                                                                  variant=variant, per_member=True)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, variant=variant, member_results=True,
                         **ekw) as eng:
        F, G = eng.eval(w.x)
        foms, grads = eng.member_results()
        info = eng.info
    gm = [float(np.abs(grads[k] - grads_ref[k]).max() / np.abs(grads_ref[k]).max()) for k in range(w.E)]
    rows.append({"case": label, "n": w.n, "K": w.K, "N": w.N, "E": w.E, "sys_type": w.sys_type, "variant": variant,
                 "kernel_family": info["kernel_family"], "unitary_flow": info["unitary_flow"],
                 "rank_one_chain": info["rank_one_chain"], "time_chunks": info["time_chunks"],
                 "fused_forward": info["fused_forward"],
                 "abs_err_F": float(abs(F - F_ref)), "rel_err_G_inf": float(np.abs(G - G_ref).max() / np.abs(G_ref).max()),
                 "worst_member_rel_err_G": max(gm), "worst_member_abs_err_F": float(np.abs(foms - foms_ref).max()),
                 "bar": 1e-10})


wl = qoc.workloads
compare("C1 2x2 StateTransfer N=10", wl.config("C1"))
compare("C2 2x2 StateTransfer N=1000", wl.config("C2"))
compare("C3 4x4 UnitaryGate N=500, 64 of 1024 members", wl.config("C3", E=64))
compare("C3 same, general flow forced", wl.config("C3", E=64), flags=qoc.engine.FLAG_FORCE_GENERAL)
compare("C3 same, static variant", wl.config("C3", E=64), variant=1)
compare("C4 16x16 Liouvillian CoherenceTransfer N=1000, 8 of 1024 members", wl.config("C4", E=8))
compare("C4 same, dense MFMA chain forced", wl.config("C4", E=8), flags=qoc.engine.FLAG_FORCE_GENERAL)
compare("C5 32x32 UnitaryGate N=2000, 2 of 4096 members", wl.config("C5", E=2))
golden = []
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.json"))):
    c, A, B, Xi, Xt, wts, x, exp, traj = load_case(path)
    with qoc.GrapeEngine(c["sys_type"], A, B, Xi, Xt, wts, c["T"], c["N"], variant=c["variant"]) as eng:
        F, G = eng.eval(x)
    Gx = np.array(exp["G"])
    golden.append({"fixture": os.path.basename(path), "abs_err_F": float(abs(F - exp["F"])),
                   "rel_err_G_inf": float(np.abs(G - Gx).max() / np.abs(Gx).max())})
print(json.dumps({"oracle": "oracle/grape_oracle.c (float64 restatement of the reference)", "vs_oracle": rows,
                  "vs_mpmath_golden_50_digits": golden}, indent=1))
