#!/bin/bash
# Copies what tools/refresh_profiles.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/ (the tracked, judged copies):
#   tools/publish_profiles.sh <tag>
set -eu
TAG=${1:?tag}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/prof_$TAG
P=$ROOT/profiles
for f in C3_E1024 C4_E1024 C4_E2048 C4_E128 C4dense_E1024 C4expm_E1024 C4x1 C5_E4096 C5x1 C6_E256; do
  [ -f "$O/${f}_kernel_stats.csv" ] && cp "$O/${f}_kernel_stats.csv" "$P/${TAG}_${f}_kernel_stats.csv"
done
for f in C3_phase_stamps.json C4_flow_crossover.txt dpp_fmac.txt exact_time.txt group_overhead_C3.json group_overhead_C3_E128.json \
         parity.json pipe_mix.txt horner_step.txt wave_placement.txt lbfgs_time.txt C4_whole.txt bench_C3_1gpu_details.json vec32_bench.json dpp_chunks_time.txt dense_forms_time.txt soak.txt perf_gate.txt \
         bench_C3_1gpu.json bench_C3_1gpu_driver_args.json bench_C3_1gpu_forced_1rank_collective.json \
         bench_C3_general_flow_1gpu.json bench_C3_shard_E128.json bench_C3_shard_E256.json bench_C3_shard_E512.json; do
  [ -s "$O/$f" ] && cp "$O/$f" "$P/${TAG}_$f"
done
cd "$ROOT"
python3 tools/profile_summary.py "$O/pmc_C3" ${TAG}_C3_E1024 "C3 (4x4, K=4, N=500, E=1024), bench.py default config." C3_E1024 | tail -1
python3 tools/profile_summary.py "$O/pmc_C4" ${TAG}_C4_E1024 "C4 (16x16 Liouvillian, N=1000, E=1024): the vector flow of action_thin.hip (default; action_parts_kernel<false>)." C4_E1024 | tail -1
python3 tools/profile_summary.py "$O/pmc_C4_E2048" ${TAG}_C4_E2048 "C4 with 2048 members: two members per wave (action_parts_kernel<true>)." C4_E2048 | tail -1
python3 tools/profile_summary.py "$O/pmc_C4dense" ${TAG}_C4dense_E1024 "C4 with GRAPE_NO_THIN=1: dense MFMA chain." C4dense_E1024 | tail -1
python3 tools/profile_summary.py "$O/pmc_C4expm" ${TAG}_C4expm_E1024 "C4 with GRAPE_ACTION=0: MFMA expm on the hoisted control sum + backward vector chain." C4expm_E1024 | tail -1
python3 tools/profile_summary.py "$O/pmc_C5" ${TAG}_C5_E4096 "C5 (32x32, K=6, N=2000, E=4096): ctrl_sum_kernel + grid_prop_kernel (sweep_grid.hip's expm) + chain_tile_unitary_kernel." C5_E4096 | tail -1
python3 tools/profile_summary.py "$O/pmc_C6" ${TAG}_C6_E256 "C6 (64x64 six-qubit UnitaryGate, K=6, N=500, E=256): sweep_grid.hip." C6_E256 | tail -1
