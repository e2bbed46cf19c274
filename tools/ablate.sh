#!/bin/bash
# Diagnostic: build variants of libgrape_hip.so with parts of sweep_pair.hip cut out (GRAPE_ABL bitmask:
# 1 no operator LDS reads in the H build, 2 no P store, 4 no chunk product, 8 no P load in the backward
# sweep, 16 no gradient traces, 32 no expm, 64 backward sweep = loads only) into build/abl/.  Results are WRONG on purpose; only the
# kernel time of `python bench.py --no-extra --no-cpu-baseline` with GRAPE_HIP_LIB=<variant> is read.
# Flags, target id and the object list come from the product Makefile, so a variant differs from the product in the one file only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/quoptimalcontrol.jl_amd/csrc
make -C $C -j4 > /dev/null
FLAGS=$(make -s -C $C print-hipflags)
OFFLOAD=$(make -s -C $C print-offload)
OBJS=$(make -s -C $C print-objs)
mkdir -p $ROOT/build/abl
for m in "$@"; do
  ( objs=""
    for o in $OBJS; do
      if [ "$o" = sweep_pair.o ]; then objs="$objs $ROOT/build/abl/sweep_pair_$m.o"; else objs="$objs $C/$o"; fi
    done
    /opt/rocm/bin/hipcc $FLAGS -DGRAPE_ABL=$m -c $C/sweep_pair.hip -o $ROOT/build/abl/sweep_pair_$m.o &&
    /opt/rocm/bin/hipcc $OFFLOAD -shared -fPIC -o $ROOT/build/abl/libgrape_abl_$m.so $objs -ldl ) &
done
wait
