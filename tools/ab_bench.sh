#!/bin/bash
# A/B of two builds of the headline run on ONE box, alternating (VERDICT r4 #1b): the tree this script lives in ("head")
# against another checkout with its own library and bench.py ("base", e.g. `git worktree add build/r3tree 54dd453` +
# `make -C build/r3tree/quoptimalcontrol.jl_amd/csrc`).  Each leg is the driver's invocation without the extras.
#   tools/ab_bench.sh build/r3tree 10 > profiles/r05_C3_ab.txt
BASE=${1:-build/r3tree}
ROUNDS=${2:-10}
ARGS="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
ku=r.get("kernel_us") or {"avg":r.get("kernel_avg_us"),"min":r.get("kernel_min_us"),"median":r.get("kernel_median_us")}
b=d.get("box",{})
print("%-5s value %8.1f  ms_per_step %.5f  kernel_us avg %.2f min %.2f median %.2f  step-kernel %.2f us  clock %s" % (sys.argv[1], d["value"], d["ms_per_step"], ku["avg"], ku["min"], ku["median"], 1e3*d["ms_per_step"]-ku["avg"], b.get("clock_ghz")))'
echo "# A/B on one box, alternating, python bench.py $ARGS;  base = $BASE ($(git -C "$ROOT/$BASE" rev-parse --short HEAD 2>/dev/null)), head = $(git -C "$ROOT" rev-parse --short HEAD 2>/dev/null)"
for i in $(seq 1 "$ROUNDS"); do
    (cd "$ROOT/$BASE" && python3 bench.py $ARGS 2>/dev/null | python3 -c "$pick" base)
    (cd "$ROOT" && python3 bench.py $ARGS 2>/dev/null | python3 -c "$pick" head)
    if [ -n "$AB_THIRD_ENV" ]; then       # e.g. AB_THIRD_ENV="GRAPE_MF_PUBLISH=0": head again with one switch flipped
        (cd "$ROOT" && env $AB_THIRD_ENV python3 bench.py $ARGS 2>/dev/null | python3 -c "$pick" "head*")
    fi
done
