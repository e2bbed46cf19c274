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
            k = "sweep" if "sweep" in k else ("reduce1" if "stage1" in k else ("reduce2" if "stage2" in k else None))
            if k is None:
                continue
            out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in out.items()}
print(json.dumps(res, indent=1))
