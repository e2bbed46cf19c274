#!/bin/bash
# Round 6's evidence under profiles/ (one gpurun call; a subset of tools/refresh_profiles.sh for the kernels this round touched):
#   tools/refresh_profiles_r06.sh      -> gpurun_out/prof_r06/...   then  tools/publish_profiles.sh r06 (+ the copies listed at its end)
set -u
TAG=r06
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, env, bench args...
  local name=$1 envs=$2; shift 2
  rm -rf /tmp/st_$name
  ( export $envs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python3 "$ROOT/bench.py" --blocks 1 --no-cpu-baseline --no-extra "$@" > "$OUT/stats_$name.log" 2>&1 )
  f=$(find /tmp/st_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
}
stats C3_E1024 GRAPE_X=0 --steps 400 --warmup 50
stats C4_E1024 GRAPE_X=0 --config C4 --steps 40 --warmup 5
stats C5_E4096 GRAPE_X=0 --config C5 --steps 2 --warmup 1
stats C6_E256 GRAPE_X=0 --config C6 --steps 3 --warmup 1
stats C7_E64 GRAPE_X=0 --config C7 --steps 5 --warmup 1
stats C6x1 GRAPE_X=0 --config C6 --ensemble 1 --steps 50 --warmup 5
stats C7x1 GRAPE_X=0 --config C7 --ensemble 1 --steps 10 --warmup 2
stats L1d GRAPE_X=0 --config L1d --steps 200 --warmup 20
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C3" > "$OUT/pmc_C3.log" 2>&1
bash "$ROOT/tools/prof_pmc.sh" "$OUT/pmc_C7" --config C7 --steps 3 --warmup 1 > "$OUT/pmc_C7.log" 2>&1
cd "$ROOT"
python3 tools/phase_profile.py --config C3 > "$OUT/C3_phase_stamps.json" 2> /dev/null
python3 -m pytest tests/test_gpu_perf_gate.py -q -s 2>&1 | grep -E "perf gate|passed|failed" > "$OUT/perf_gate.txt"
python3 tools/anysize_time.py > "$OUT/anysize_time.txt" 2>&1
python3 tools/single_big_time.py > "$OUT/single_big_time.txt" 2>&1
for seed in 61 62 63 64; do python3 tools/soak.py 1500 $seed 2>&1 | tail -1; done > "$OUT/soak.txt"
python3 tools/soak_api.py 400 6 2>&1 | tail -1 >> "$OUT/soak.txt"
python3 tools/parity_report.py > "$OUT/parity.json" 2> "$OUT/parity.log"
python3 bench.py --details "$OUT/bench_C3_1gpu_details.json" 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu.json"
python3 bench.py --steps 20 --warmup 5 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu_driver_args.json"
python3 bench.py --force-dist --no-extra 2> /dev/null | tail -1 > "$OUT/bench_C3_1gpu_forced_1rank_collective.json"
ls -la "$OUT" | tail -30
