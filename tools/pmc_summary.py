#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc*/pmc_counter_collection.csv) of
    bench.py --steps K --warmup W --no-cpu-baseline --no-roofline --no-extra --split 1
into profiles/<round>/ (per-kernel and per-INSTANTIATION averages) and profiles/pmc_traffic.json (fabric bytes per launch of each
brepgen kernel together with the launches per step of the run it was measured on; read back by bench.py's roofline.traffic, which
refuses a measurement taken on another step mix).

    python tools/pmc_summary.py <round> <K> <W>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W = max(int(sys.argv[3]) if len(sys.argv) > 3 else 3, 3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
inst = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc[0-9]", "pmc_counter_collection.csv"))):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"].startswith("SQ_"):                   # the SQ pass runs on a shorter, differently mixed run: tools/pmc_dirs.py
            continue
        name = r["Kernel_Name"]
        if "bg::" not in name:
            continue
        full = name.split("(")[0].replace("void ", "").replace("bg::", "")
        short = full.split("<")[0]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        # bench.py's profiler books the 256 x 256 kernel's split-residual launches (MODE 3) apart from its plain / fold launches
        row = short
        if short == "gemm16_p256_kernel":
            mode = full.split("<")[1].split(",")[1].strip() if "<" in full else ""
            row = "gemm16_p256_kernel(256x256, split-residual launches)" if mode == "3" else "gemm16_p256_kernel(256x256)"
        for table, key in ((acc, row), (inst, full)):
            table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            table[key]["duration_ns"].append(dur)


def summarise(table):
    out, traffic = {}, {}
    for k, c in table.items():
        out[k] = {n: round(sum(v) / len(v), 1) for n, v in c.items()}
        out[k]["samples"] = {n: len(v) for n, v in c.items()}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch, write = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            traffic[k] = {"bytes_per_launch": round((2.0 * fetch + write) * 1024), "fetch_kib_raw": round(fetch, 1),
                          "write_kib": round(write, 1), "launches_per_step": round(len(c["FETCH_SIZE"]) / (K + W), 3),
                          "measured_on": f"bench.py --steps {K} --warmup {W} --split 1 under rocprofv3 --pmc (all {K + W} steps' launches; "
                                         "2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE)"}
    return out, traffic


out, traffic = summarise(acc)
out_i, traffic_i = summarise(inst)
os.makedirs(os.path.join(ROOT, "profiles", rnd), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", rnd, "rocprofv3_pmc_per_kernel_avg.json"), "w"), indent=1, sort_keys=True)
json.dump({"counters": out_i, "traffic": traffic_i}, open(os.path.join(ROOT, "profiles", rnd, "rocprofv3_pmc_per_instantiation_avg.json"), "w"),
          indent=1, sort_keys=True)
json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in traffic.items() if k.startswith("gemm16")}, indent=1))
