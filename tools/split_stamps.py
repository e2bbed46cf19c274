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
names = ["p1 P_t arrives + conversion", "p1 products", "p1 barrier wait", "exchange", "p2 pull-back", "p2 X_t arrives / rebuild", "p2 product XL",
         "p2 commutator", "p2 traces", "p1 store + load issue"]
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, member_results=True) as eng:
    for _ in range(3):
        eng.eval(w.x)
    foms, grads = eng.member_results()
    print(eng.kernel_names())
g = np.swapaxes(grads, 1, 2).reshape(E, -1)     # [member][t * K + c]
p0 = g[:, :10]
# part 1's rows start at K * lo: find lo as the first row whose entries look like cycle counts at position K * lo
K = w.K
cand = [t for t in range(3, w.N) if np.all(g[:, K * t + 1] > 1e3) and np.all(g[:, K * t:K * t + 2] == np.round(g[:, K * t:K * t + 2]))]
lo = cand[0]
p1 = g[:, K * lo:K * lo + 10]
for nm, P, n_sl in (("wave 0 (slices [0, %d))" % lo, p0, lo), ("wave 1 (slices [%d, %d))" % (lo, w.N), p1, w.N - lo)):
    med = np.median(P, axis=0)
    tot = med.sum()
    print(f"{nm}: {tot:.0f} cycles in the stamped regions, {tot / n_sl:.0f} per slice")
    for i, nme in enumerate(names):
        print(f"   {nme:30s} {med[i]:12.0f}  {100 * med[i] / tot:5.1f} %   {med[i] / n_sl:8.0f} / slice")
