#!/usr/bin/env python3
"""Diagnostic: per-wave phase stamps of the lane-pair sweep with the hardware placement of every wave (XCC, SE, CU, SIMD from
HW_ID) -> an .npz for offline analysis + a summary: which waves share a SIMD, how far apart their phases run.
usage: tools/wave_map.py out.npz [--ensemble E]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quoptimalcontrol_jl_amd as qoc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--config", default="C3")
ap.add_argument("--ensemble", type=int, default=0)
a = ap.parse_args()
w = qoc.workloads.config(a.config, E=a.ensemble or None)
with qoc.GrapeEngine(w.sys_type, w.A, w.B, w.Xi, w.Xt, w.wts, w.T, w.N, flags=qoc.engine.FLAG_PHASE_STAMPS) as eng:
    for _ in range(3):
        eng.eval(w.x)
    raw = eng.phase_stamps()
    info = eng.info
st = raw.astype(np.int64)
hw = raw[:, 7]
lo = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
wave_id, simd, cu, sh, se = lo & 0xF, (lo >> 4) & 3, (lo >> 8) & 0xF, (lo >> 12) & 1, (lo >> 13) & 7
W = info["waves_per_member"]
member = np.arange(len(st)) // W
wv = np.arange(len(st)) % W
np.savez(a.out, st=st, xcc=xcc, se=se, sh=sh, cu=cu, simd=simd, wave_id=wave_id, member=member, wv=wv)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
slot = key * 4 + simd
t0 = st[:, 0].min()
out = {"waves": int(len(st)), "W": int(W), "cus_used": int(len(set(key.tolist()))), "simds_used": int(len(set(slot.tolist())))}
# who shares a SIMD: difference of member index within the workgroup
pairs = {}
for s in np.unique(slot):
    idx = np.nonzero(slot == s)[0]
    if len(idx) == 2:
        d = (int(member[idx[0]]) % 4, int(wv[idx[0]]), int(member[idx[1]]) % 4, int(wv[idx[1]]))
        pairs[d] = pairs.get(d, 0) + 1
out["simd_sharing_(mb,wave,mb,wave)_counts"] = {str(k): v for k, v in sorted(pairs.items(), key=lambda kv: -kv[1])[:12]}
d = np.diff(st[:, :5], axis=1)
for i, n in enumerate(["A", "B", "C", "D"]):
    out[n] = [float(np.percentile(d[:, i], q)) for q in (0, 10, 50, 90, 100)]
out["end_D_cycles_after_first_start"] = [float(np.percentile(st[:, 4] - t0, q)) for q in (0, 10, 50, 90, 100)]
out["end_A_cycles_after_first_start"] = [float(np.percentile(st[:, 1] - t0, q)) for q in (0, 10, 50, 90, 100)]
out["span_ns"] = float((st[:, 6].max() - st[:, 5].min()) * 10.0)
print(json.dumps(out))
