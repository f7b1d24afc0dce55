#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc*/pmc_counter_collection.csv) into profiles/<round>/ and
profiles/pmc_traffic.json (bytes per launch of each brepgen kernel; read back by bench.py's roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc*", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "bg::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("bg::", "")
        short = short.split("<")[0]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[short]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out, traffic = {}, {}
for k, c in acc.items():
    out[k] = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
    out[k]["samples"] = {n: len(v) for n, v in c.items()}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch, write = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        traffic[k] = {"bytes_per_launch": round((2.0 * fetch + write) * 1024), "fetch_kib_raw": round(fetch, 1),
                      "write_kib": round(write, 1), "note": "2x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, "
                      "averaged over all launches of the kernel in bench.py --steps 3"}
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "rocprofv3_pmc_per_kernel_avg_final.json"), "w"), indent=1, sort_keys=True)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
