#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes: python tools/pmc_dirs.py <pass dir> [<pass dir> ...]  ->  JSON on stdout."""
import collections
import csv
import glob
import json
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if "bg::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "").replace("bg::", "")
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[short]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    out[k] = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
    out[k]["launches_seen"] = max(len(v) for v in c.values())
print(json.dumps(out, indent=1, sort_keys=True))
