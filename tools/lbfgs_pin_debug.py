import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import quoptimalcontrol_jl_amd as qoc
from oracle import optim_lbfgs
from test_gpu_lbfgs import _ensemble_case
for case in ("ug_static", "ug_inplace"):
    args, x0, variant = _ensemble_case(qoc, case)
    with qoc.GrapeEngine(*args, variant=variant) as eng:
        ref = optim_lbfgs.lbfgs(lambda x: eng.eval(x), x0, iterations=8)
        x, info = eng.lbfgs(x0, iterations=8, line_search="optim")
        al, ev = eng.lbfgs_trace()
    print(case, "ref alphas", [float(f"{t['alpha']:.6g}") for t in ref["trace"]])
    print(case, "dev alphas", [float(f"{a:.6g}") for a in al])
    print(case, "ref evals ", [t["evaluations"] for t in ref["trace"]])
    print(case, "dev evals ", list(ev))
    print(case, "ref f", [float(f"{t['f']:.10g}") for t in ref["trace"]], "dev min", info["minimum"], info["status"])
