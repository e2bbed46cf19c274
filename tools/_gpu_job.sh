#!/bin/bash
cd $GRAFT_REPO_ROOT
for E in 128 192 256 320 384; do
for mode in 1 0; do
echo "== C4 E=$E GRAPE_ACTION=$mode"; GRAPE_ACTION=$mode timeout 300 python bench.py --config C4 --ensemble $E --steps 20 --warmup 5 --blocks 2 --no-extra --no-cpu-baseline 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], [(k['kernel'][:30], round(k['avg_us'])) for k in d['roofline'].get('kernels',[])])
    elif 'rror' in l: print(l.strip()[:300])
"
done
done
