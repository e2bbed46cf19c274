#!/bin/bash
# Diagnostic: build variants of libgrape_hip.so with parts of sweep_pair.hip cut out (GRAPE_ABL bitmask:
# 1 no operator LDS reads in the H build, 2 no P store, 4 no chunk product, 8 no P load in the backward
# sweep, 16 no gradient traces, 32 no expm, 64 backward sweep = loads only) into build/abl/.  Results are WRONG on purpose; only the
# kernel time of `python bench.py --no-extra --no-cpu-baseline` with GRAPE_HIP_LIB=<variant> is read.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/quoptimalcontrol.jl_amd/csrc
mkdir -p $ROOT/build/abl
for m in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=on -DGRAPE_ABL=$m -c $C/sweep_pair.hip -o $ROOT/build/abl/sweep_pair_$m.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/abl/libgrape_abl_$m.so $C/grape_api.o $C/sweep_small.o $ROOT/build/abl/sweep_pair_$m.o $C/sweep_tile.o $C/sweep_thin.o $C/prop_hoist.o $C/reduce.o $C/lbfgs.o $C/exact_grad.o $C/exact_tile.o -ldl ) &
done
wait
