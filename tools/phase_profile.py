#!/usr/bin/env python3
"""Diagnostic: where does a wave of the sweep kernel spend its cycles?  Uses the phase-stamp
build (GRAPE_FLAG_PHASE_STAMPS); shares only -- never quote this build's run time."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--ensemble", type=int, default=0)
ap.add_argument("--slices-per-lane", type=int, default=0)
ap.add_argument("--waves-per-member", type=int, default=0)
ap.add_argument("--force-general", action="store_true")
a = ap.parse_args()
w = qoc.workloads.config(a.config, E=a.ensemble or None)
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_PHASE_STAMPS | (qoc.engine.FLAG_FORCE_GENERAL if a.force_general else 0),
                     slices_per_lane=a.slices_per_lane, waves_per_member=a.waves_per_member) as eng:
    for _ in range(3):
        eng.eval(w.x)
    eng_stamps_raw = eng.phase_stamps()
    st = eng_stamps_raw.astype(np.int64)
    info = eng.info
d = np.diff(st[:, :5], axis=1)                      # cycles per phase per wave
real = (st[:, 6] - st[:, 5]) * 10.0                 # ns (100 MHz counter)
tot = st[:, 4] - st[:, 0]
names = ["A propagators", "B scan", "C forward", "D backward+grad"]
out = {"config": a.config, "E": w.E, "S": info["slices_per_lane"], "W": info["waves_per_member"],
       "waves": int(len(st)), "wave_total_cycles_median": float(np.median(tot)),
       "wave_real_ns_median": float(np.median(real)),
       "clock_GHz_median": float(np.median(tot / np.maximum(real, 1))),
       "kernel_span_ns": float((st[:, 6].max() - st[:, 5].min()) * 10.0),
       "phases": {n: {"cycles_median": float(np.median(d[:, i])), "share": float(np.median(d[:, i]) / np.median(tot))}
                  for i, n in enumerate(names)}}
pc = lambda a: [float(np.percentile(a, q)) for q in (0, 10, 50, 90, 100)]
out["percentiles_0_10_50_90_100"] = {
    "wave_start_ns_after_first": pc((st[:, 5] - st[:, 5].min()) * 10.0),
    "wave_end_ns_after_first_start": pc((st[:, 6] - st[:, 5].min()) * 10.0),
    "wave_real_ns": pc(real), "wave_total_cycles": pc(tot),
    **{n: pc(d[:, i]) for i, n in enumerate(names)}}
hw = eng_stamps_raw[:, 7]
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int)
out["per_xcc_wave_total_cycles_median"] = {int(x): float(np.median(tot[xcc == x])) for x in sorted(set(xcc.tolist()))}
print(json.dumps(out, indent=1))
