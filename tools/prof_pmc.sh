#!/bin/bash
# PMC passes for the sweep kernel (separate runs, kernel-trace only -- never with sys/hip traces).
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline --no-extra "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done
python3 "$ROOT/tools/summarize_pmc.py" "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
