#!/bin/bash
# Regenerates the evidence under profiles/ for the kernels of HEAD, on the GPU box (one gpurun call):
#   tools/refresh_profiles.sh <tag>      -> gpurun_out/prof_<tag>/...   (copy what is to be judged into profiles/)
# rocprofv3 runs the program itself after `--` (python3 bench.py ...), PMC passes are separate kernel-trace-only runs.
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, env, bench args...
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/st_$name
  env $envs true
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python3 "$ROOT/bench.py" --blocks 1 --no-cpu-baseline --no-extra "$@" > "$OUT/stats_$name.log" 2>&1 )
  f=$(find /tmp/st_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
}
stats C3_E1024 GRAPE_X=0 --steps 400 --warmup 50    # (a 36-launch run averages the cold first launches in: 78 us)
stats C4_E1024 GRAPE_X=0 --config C4 --steps 40 --warmup 5
stats C4dense_E1024 GRAPE_NO_THIN=1 --config C4 --steps 40 --warmup 5
stats C4expm_E1024 GRAPE_ACTION=0 --config C4 --steps 40 --warmup 5      # the MFMA expm + vector chain the vector flow replaces at C4
stats C4_E2048 GRAPE_X=0 --config C4 --ensemble 2048 --steps 30 --warmup 5     # two members per wave (action_parts_kernel<true>)
stats C4_E128 GRAPE_X=0 --config C4 --ensemble 128 --steps 100 --warmup 10      # the per-GPU shard of an 8-GPU run: expm + chain_prop_kernel
stats C5_E4096 GRAPE_X=0 --config C5 --steps 2 --warmup 1
stats C6_E256 GRAPE_X=0 --config C6 --steps 3 --warmup 1       # 64 x 64 (sweep_grid.hip)
stats C5x1 GRAPE_X=0 --config C5 --ensemble 1 --steps 200 --warmup 20     # single problems: the chunked time axis
stats C4x1 GRAPE_X=0 --config C4 --ensemble 1 --steps 200 --warmup 20
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C3" > "$OUT/pmc_C3.log" 2>&1
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C4" --config C4 > "$OUT/pmc_C4.log" 2>&1
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C4_E2048" --config C4 --ensemble 2048 > "$OUT/pmc_C4_E2048.log" 2>&1
GRAPE_NO_THIN=1 bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C4dense" --config C4 > "$OUT/pmc_C4dense.log" 2>&1
GRAPE_ACTION=0 bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C4expm" --config C4 > "$OUT/pmc_C4expm.log" 2>&1
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C5" --config C5 --steps 2 --warmup 1 > "$OUT/pmc_C5.log" 2>&1
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C6" --config C6 --steps 3 --warmup 1 > "$OUT/pmc_C6.log" 2>&1
cd "$ROOT"
python3 tools/phase_profile.py --config C3 > "$OUT/C3_phase_stamps.json" 2> /dev/null
python3 tools/group_overhead.py > "$OUT/group_overhead_C3.json" 2> /dev/null
python3 tools/group_overhead.py --ensemble 128 > "$OUT/group_overhead_C3_E128.json" 2> /dev/null
./tools/ubench/pipe_mix > "$OUT/pipe_mix.txt" 2>&1
./tools/ubench/dpp_fmac > "$OUT/dpp_fmac.txt" 2>&1
./tools/ubench/horner_step > "$OUT/horner_step.txt" 2>&1
./tools/ubench/wave_placement > "$OUT/wave_placement.txt" 2>&1
( python3 tools/lbfgs_time.py 30; python3 tools/lbfgs_time.py 30 optim; GRAPE_LBFGS_MB=0 python3 tools/lbfgs_time.py 30; GRAPE_LBFGS_MB=0 python3 tools/lbfgs_time.py 30 optim; GRAPE_LBFGS_MB=0 GRAPE_LBFGS_FUSED_PROBE=0 python3 tools/lbfgs_time.py 30 optim ) > "$OUT/lbfgs_time.txt" 2> /dev/null
for E in 1024 1536 2048 3072 4096; do for W in 0 1; do
  echo "E=$E whole=$W $(GRAPE_ACTION=1 GRAPE_ACT_WHOLE=$W python3 bench.py --config C4 --ensemble $E --steps 20 --warmup 5 --blocks 2 --no-extra --no-cpu-baseline --verbose 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "evals/s", round(d["ms_per_step"],3), "ms", [round(k["avg_us"],1) for k in d["roofline"]["kernels"]])')"
done; done > "$OUT/C4_whole.txt"
( python3 tools/vec32_bench.py 1024 2000; python3 tools/vec32_bench.py 4096 2000 ) > "$OUT/vec32_bench.json" 2> /dev/null
python3 tools/exact_time.py > "$OUT/exact_time.txt" 2> /dev/null
python3 tools/dpp_chunks_time.py 1 2 8 24 40 48 2> /dev/null > "$OUT/dpp_chunks_time.txt"      # small rank-one ensembles: chunked propagator chain vs the flows it replaced
( python3 tools/dense_forms_time.py; GRAPE_FORMS_VALU=1 python3 tools/dense_forms_time.py ) 2> /dev/null > "$OUT/dense_forms_time.txt"
for E in 128 192 224 256 288 320 512 1024 2048 4096; do for m in 1 0; do echo "E=$E GRAPE_ACTION=$m $(GRAPE_ACTION=$m python3 bench.py --config C4 --ensemble $E --steps 20 --warmup 5 --blocks 2 --no-extra --no-cpu-baseline 2> /dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "evals/s", round(d["ms_per_step"],3), "ms")')"; done; done > "$OUT/C4_flow_crossover.txt"
for seed in 51 52 53 54 55 56 57 58; do python3 tools/soak.py 1500 $seed 2>&1 | tail -1; done > "$OUT/soak.txt"
python3 -m pytest tests/test_gpu_perf_gate.py -q -s 2>&1 | grep -E "perf gate|passed|failed" > "$OUT/perf_gate.txt"
python3 tools/soak_api.py 600 5 2>&1 | tail -1 >> "$OUT/soak.txt"
python3 tools/parity_report.py > "$OUT/parity.json" 2> "$OUT/parity.log"
python3 bench.py --details "$OUT/bench_C3_1gpu_details.json" 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu.json"
python3 bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu_driver_args.json"
python3 bench.py --force-general --no-extra 2> /dev/null | tail -1 > "$OUT/bench_C3_general_flow_1gpu.json"
python3 bench.py --force-dist --no-extra 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu_forced_1rank_collective.json"
for E in 128 256 512; do
  python3 bench.py --ensemble $E --no-extra --no-cpu-baseline 2> /dev/null | tail -1 > "$OUT/bench_C3_shard_E$E.json"
done
# keep the raw PMC CSVs out of the merge (64 MiB cap): summaries only
find "$OUT" -name 'pass*' -maxdepth 2 -type d -exec rm -rf {} + 2> /dev/null
ls -la "$OUT"
