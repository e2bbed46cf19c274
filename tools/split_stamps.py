#!/usr/bin/env python3
"""Diagnostic: where chain_tile_split_kernel's waves spend their cycles at C4dense.  Needs the stamp build
(tools/variant.sh sweep_tile.hip stamp -DGRAPE_SPLIT_PAD=1 -DGRAPE_SPLIT_STAMP=1; GRAPE_HIP_LIB=build/abl/libgrape_stamp.so):
every wave sums the shader-cycle counter's increments between points of its slice loops and leaves them in its member's
gradient rows (the results are wrong on purpose)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = qoc.workloads.config("C4", E=E)
os.environ["GRAPE_NO_THIN"] = "1"
names = ["p1 Gc_t arrives, expm, P_t store", "p1 propagate (2 products)", "barrier wait", "-", "p2 propagate (2 products)", "p2 state arrives",
         "p2 product X L", "p2 commutator", "p2 traces", "p1 state store (wave 1: + its products)"]
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True) as eng:
    for _ in range(3):
        eng.eval(w.x)
    foms, grads = eng.member_results()
    print(eng.kernel_names())
g = np.swapaxes(grads, 1, 2).reshape(E, -1)     # [member][t * K + c]
K, Nh = w.K, w.N // 2
for nm, P, n_sl in (("wave 0 (forward; emits slices [%d, %d))" % (Nh, w.N), g[:, K * Nh:K * Nh + 10], w.N - Nh),
                    ("wave 1 (backward; emits slices [0, %d))" % Nh, g[:, :10], Nh)):
    med = np.median(P, axis=0)
    tot = med.sum()
    print(f"{nm}: {tot:.0f} cycles in the stamped regions, {tot / n_sl:.0f} per slice of either phase")
    for i, nme in enumerate(names):
        if nme != "-":
            print(f"   {nme:30s} {med[i]:12.0f}  {100 * med[i] / tot:5.1f} %   {med[i] / n_sl:8.0f} / slice")
