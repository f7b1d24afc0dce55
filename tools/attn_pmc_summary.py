#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc pass over tools/attn_bench.py (gpurun_out/pmc_attn/**/*counter_collection.csv) for the attention kernels:
per kernel instantiation and launch shape (grid size) the average of every counter, the launch duration, and the derived figures
DESIGN.md quotes -- shader clock = SQ_BUSY_CYCLES / 32 shader engines / duration, matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs
/ (duration x clock).      python tools/attn_pmc_summary.py <round>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_attn", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "attn16" not in name:
            continue
        key = name.split("(")[0].replace("void ", "").replace("bg::", "") + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, c in acc.items():
    row = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
    row["launches"] = len(c["duration_ns"]) // max(1, len([n for n in c if n != "duration_ns"]))
    dur = row["duration_ns"]
    if "SQ_BUSY_CYCLES" in row and dur:
        ghz = row["SQ_BUSY_CYCLES"] / 32.0 / dur
        row["shader_clock_GHz"] = round(ghz, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in row:
            row["matrix_pipe_busy_at_that_clock"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (dur * ghz), 3)
            row["matrix_pipe_busy_vs_2.4GHz"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (dur * 2.4), 3)
    out[k] = row
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "attn_pmc_per_launch_shape.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items()):
    print(k, {n: v[n] for n in ("duration_ns", "shader_clock_GHz", "matrix_pipe_busy_at_that_clock", "matrix_pipe_busy_vs_2.4GHz") if n in v})
