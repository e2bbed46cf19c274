#!/usr/bin/env python3
"""Turn a tools/prof_pmc.sh output directory into profiles/<tag>_pmc.json: per-kernel counter means plus the derived
figures DESIGN.md quotes (HBM traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md, VALU / wait shares,
MFMA pipe utilisation).  usage: tools/profile_summary.py <pmc_dir> <tag> "<note>" [traffic_key]"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc_dir, tag, note = sys.argv[1], sys.argv[2], sys.argv[3]
raw = json.load(open(os.path.join(pmc_dir, "pmc_summary.json")))
out = {"note": note + "  rocprofv3 --pmc passes (tools/prof_pmc.sh: separate passes, kernel-trace only), per-launch means. "
               "HBM traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 tallies 128-B read requests at 64 B; WRITE_SIZE exact). "
               "SQ_* cycle counters are quad-cycles. MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).",
       "kernels": {}}
for k, cs in raw.items():
    c = {n: v["mean"] for n, v in cs.items()}
    d = {"counters": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_traffic_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        d["valu_active_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / wc
        d["wait_any_share"] = c.get("SQ_WAIT_ANY", 0) / wc
        d["wait_inst_any_share"] = c.get("SQ_WAIT_INST_ANY", 0) / wc
        d["valu_instructions_per_wave"] = c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)
        d["waves"] = c.get("SQ_WAVES")
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        d["mfma_instructions"] = c.get("SQ_INSTS_MFMA")
        d["mfma_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    if c.get("GRBM_GUI_ACTIVE"):
        d["kernel_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
    out["kernels"][k] = d
path = os.path.join(root, "profiles", f"{tag}_pmc.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
if len(sys.argv) > 4:
    # profiles/traffic.json: what bench.py quotes "from profile" -- HBM bytes of the timed kernels per evaluation and the
    # matrix-core pipe utilisation of every MFMA kernel of this configuration
    key = sys.argv[4]
    timed = [k for k in out["kernels"] if not k.startswith("reduce") and "hbm_traffic_bytes" in out["kernels"][k]]
    tpath = os.path.join(root, "profiles", "traffic.json")
    t = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if timed:
        t[key] = sum(out["kernels"][k]["hbm_traffic_bytes"] for k in timed)
    mf = {k: round(d["mfma_utilisation"], 4) for k, d in out["kernels"].items() if d.get("mfma_utilisation")}
    if mf:
        t.setdefault("mfma_utilisation", {})[key] = mf
    t.setdefault("_sources", {})[key] = f"profiles/{tag}_pmc.json ({' + '.join(timed)})"
    t.pop("_source", None)
    json.dump(t, open(tpath, "w"), indent=1)
    print("updated", tpath)
