import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import quoptimalcontrol_jl_amd as qoc
from test_gpu_edges import _problem
Wopt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for n in (2, 3):
    for kern in ("default", "pair"):
        if n == 3 and kern == "pair": continue
        if kern != "default": os.environ["GRAPE_SMALL_KERNEL"] = kern
        else: os.environ.pop("GRAPE_SMALL_KERNEL", None)
        w = _problem(qoc, n, 2, 500, 1024, "StateTransfer", seed=1)
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_TIME_KERNELS, waves_per_member=Wopt) as eng:
            xf = np.ascontiguousarray(w.x.T); call = eng.bind_eval(xf, np.empty_like(xf))
            for _ in range(20): call()
            eng.kernel_time(reset=True)
            t0 = time.perf_counter()
            for _ in range(200): call()
            dt = (time.perf_counter() - t0) / 200
            ms, cnt = eng.kernel_time()
            info = eng.info
        print(f"n={n} {kern:8s} call {dt*1e6:7.1f} us  sweep {1e3*ms/max(cnt,1):7.1f} us  S={info['slices_per_lane']} W={info['waves_per_member']} pair={info['lane_pair']} unitary={info['unitary_flow']}", flush=True)
