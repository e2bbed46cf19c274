#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter CSVs of tools/prof_pmc.sh per kernel and counter."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if "grape::" not in k:
                continue
            k = k.split("grape::")[1].split("(")[0].split("<")[0]
            k = {"sweep_small_kernel": "sweep", "sweep_pair_kernel": "sweep", "chain_tile_split_kernel": "chain_tile_split", "chain_tile_unitary_kernel": "chain_tile_unitary", "chain_thin_kernel": "chain_thin", "prop_hoist1_kernel": "prop_hoist1", "prop_hoist2_kernel": "prop_hoist2", "grid_prop_kernel": "grid_prop", "grid_chain_kernel": "grid_chain", "reduce_stage1": "reduce1", "reduce_stage2": "reduce2"}.get(k, k)
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
