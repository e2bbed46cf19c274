#!/usr/bin/env python3
"""Device-resident throughput of grape_eval_batch_device (multi-start extension):
gradient-evaluations per second when n_x control arrays are evaluated per launch."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--ensemble", type=int, default=0)
ap.add_argument("--batches", default="1,8,64,256,1024")
a = ap.parse_args()
w = qoc.workloads.config(a.config, E=a.ensemble or None)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for nb in [int(v) for v in a.batches.split(",")]:
    X = rng.uniform(0, 1, (nb, w.N, w.K))                       # each (K,N) column-major
    xd = torch.as_tensor(X, device=dev)
    fg = torch.zeros(nb * (w.K * w.N + 1), dtype=torch.float64, device=dev)
    with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, max_batch=nb) as eng:
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(5):
            eng.eval_batch_device(nb, xd.data_ptr(), fg.data_ptr(), st)
        torch.cuda.synchronize()
        t, n = time.perf_counter(), 50
        for _ in range(n):
            eng.eval_batch_device(nb, xd.data_ptr(), fg.data_ptr(), st)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t) / n
        inf = eng.info
    print(f"{a.config} E={w.E} n_x={nb}: {1e6 * el:9.1f} us per launch  {nb / el:12.0f} gradient-evals/s  "
          f"(W={inf['waves_per_member']}, S={inf['slices_per_lane']})")
