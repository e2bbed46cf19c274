#!/usr/bin/env python3
"""Fixed per-evaluation cost of a SHARDED evaluation, measured on one GPU: the C3 ensemble (or --ensemble) behind ONE context
as G = 1, 2, 4, 8 shards (device_ids = [0] * G, GRAPE_FLAG_GROUP_PEER_SUM: peer copies + one reduction instead of RCCL).
The shards' kernels share the GPU, so their sum is the single-context kernel time; what changes with G is the host side:
x fan-out, launch skew between the first and the last shard, issuing the sum, the wait.  Prints one JSON object."""
import argparse
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--ensemble", type=int, default=0)
    ap.add_argument("--steps", type=int, default=400)
    args = ap.parse_args()
    w = qoc.workloads.config(args.config, E=args.ensemble) if args.ensemble else qoc.workloads.config(args.config)
    out = {"workload": f"{args.config} E={w.E}", "rows": []}
    for G in (1, 2, 4, 8):
        kw = dict(devices=[0] * G, flags=qoc.engine.FLAG_GROUP_PEER_SUM) if G > 1 else {}
        with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, **kw) as eng:
            xf = np.ascontiguousarray(w.x.T)
            call = eng.bind_eval(xf, np.empty_like(xf))
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:
                call()
            if G > 1:
                eng.group_timing(reset=True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                call()
            el = time.perf_counter() - t0
            row = {"shards": G, "members_per_shard": -(-w.E // G), "us_per_eval": 1e6 * el / args.steps}
            if G > 1:
                row.update(eng.group_timing())
            out["rows"].append(row)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
