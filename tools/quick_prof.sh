#!/bin/bash
# Quick look at one bench configuration on the GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) and two PMC
# passes (wave cycles / VALU; MFMA busy / LDS) -> gpurun_out/qp_<name>/.   usage: tools/quick_prof.sh <name> [bench args...]
set -u
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/qp_$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/qp_st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qp_st -- python3 "$ROOT/bench.py" --blocks 1 --no-cpu-baseline --no-extra "$@" > "$OUT/stats.log" 2>&1
f=$(find /tmp/qp_st -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf "$OUT/pass$i"
  timeout 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-extra "$@" > "$OUT/pass$i.log" 2>&1
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
try:
    rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
    for r in rows[:8]:
        print("%-70s calls %6s avg %10.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
except Exception as e:
    print("no kernel stats:", e)
d = json.load(open(out + "/pmc_summary.json"))
for k, c in d.items():
    g = lambda n: c.get(n, {}).get("mean", float("nan"))
    if g("SQ_INSTS_MFMA") > 0:
        util = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8)
        print("%-22s MFMA pipe %.3f | per wave: VALU %.0f MFMA %.0f LDS %.0f | wait_any %.2f wait_inst %.2f valu_active %.2f" % (
            k, util, g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_MFMA") / g("SQ_WAVES"), g("SQ_INSTS_LDS") / g("SQ_WAVES"),
            g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
PY
find "$OUT" -name 'pass*' -maxdepth 1 -type d -exec rm -rf {} + 2> /dev/null
