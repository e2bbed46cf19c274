import os, sys
sys.path.insert(0, ".")
import torch
import quoptimalcontrol_jl_amd as qoc
w = qoc.workloads.config("C3", E=64, N=100)
wt = qoc.workloads.config("C4", E=4, N=20)
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
def cyc(ww, flags=0, ev=True, **kw):
    with qoc.GrapeEngine(ww.sys_type, ww.A, ww.B, ww.Xi, ww.Xt, ww.wts, ww.T, ww.N, flags=flags, **kw) as eng:
        if ev: eng.eval(ww.x)
cyc(w); cyc(wt)
F = qoc.engine
full = F.FLAG_TIME_KERNELS | F.FLAG_MEMBER_RESULTS | F.FLAG_KEEP_COSTATES
for name, fn in [("c4-timed", lambda: cyc(wt, F.FLAG_TIME_KERNELS)), ("c4-rows", lambda: cyc(wt, F.FLAG_MEMBER_RESULTS)),
                 ("c4-keep", lambda: cyc(wt, F.FLAG_KEEP_COSTATES)), ("c4-full", lambda: cyc(wt, full)), ("c3-full", lambda: cyc(w, full)),
                 ("c3+c4", lambda: (cyc(w), cyc(wt))), ("c3+c4 full", lambda: (cyc(w, full), cyc(wt, full))),
                 ("all", lambda: (cyc(w), cyc(wt), cyc(w, full), cyc(wt, full)))]:
    f0 = free()
    d = []
    for blk in range(4):
        for i in range(30): fn()
        f1 = free(); d.append((f0 - f1) / 1024); f0 = f1
    print(name, ["%.0f KiB" % v for v in d], flush=True)
