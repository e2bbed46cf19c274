#!/bin/bash
# Diagnostic: build a variant of libgrape_hip.so that differs from the product in ONE kernel file compiled with extra
# defines -> build/abl/libgrape_<tag>.so (use with GRAPE_HIP_LIB=).   usage: tools/variant.sh <file.hip> <tag> [-DX=1 ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/quoptimalcontrol.jl_amd/csrc
F=$1; TAG=$2; shift 2
make -C $C -j4 > /dev/null
FLAGS=$(make -s -C $C print-hipflags)
OFFLOAD=$(make -s -C $C print-offload)
OBJS=$(make -s -C $C print-objs)
mkdir -p $ROOT/build/abl
objs=""
for o in $OBJS; do
  if [ "$o" = "${F%.hip}.o" ]; then objs="$objs $ROOT/build/abl/${F%.hip}_$TAG.o"; else objs="$objs $C/$o"; fi
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$F -o $ROOT/build/abl/${F%.hip}_$TAG.o
/opt/rocm/bin/hipcc $OFFLOAD -shared -fPIC -o $ROOT/build/abl/libgrape_$TAG.so $objs -ldl
echo build/abl/libgrape_$TAG.so
