import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import quoptimalcontrol_jl_amd as qoc
for name, kw in (("C3", {}), ("C4", {"E": 256}), ("C5", {"E": 64})):
    w = qoc.workloads.config(name, **kw)
    for grad in ("reference", "exact"):
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, gradient=grad, objective="c1" if grad == "exact" else "fom") as eng:
            for _ in range(2): eng.eval(w.x)
            n = 5
            t0 = time.perf_counter()
            for _ in range(n): eng.eval(w.x)
            dt = (time.perf_counter() - t0) / n
        print(name, w.E, grad, f"{dt*1e3:.3f} ms", flush=True)
