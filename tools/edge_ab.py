#!/usr/bin/env python
"""In-process A/B of bg_tune settings on one eps-evaluation of the edge nets at the BASELINE configs[2..4] shapes (bench.py:
edge_net_extra):   python tools/edge_ab.py "10=2" "10=0" ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from brepgen_amd import _lib

SETTINGS = sys.argv[1:] or ["10=2", "10=0"]
KEYS = sorted({int(kv.split("=")[0]) for s in SETTINGS for kv in s.split(",") if kv})
lib = _lib.load()
for s in SETTINGS:
    for k in KEYS:
        lib.bg_tune_set(k, 0)
    for kv in filter(None, s.split(",")):
        k, v = kv.split("=")
        lib.bg_tune_set(int(k), int(v))
    rows = bench.edge_net_extra(torch.device("cuda"), evals=3)
    for r in rows:
        ker = {k.split("(")[0][-14:]: (v["total_ms"], v["tflops"]) for k, v in r["kernels_varlen"].items()}
        print(f"{s:10s} {r['workload'][:28]:28s} varlen {r['varlen']['ms_per_eval']:8.2f} ms {r['varlen']['executed_tflops']:6.1f} TF | dense {r['dense']['ms_per_eval']:8.2f} ms {r['dense']['executed_tflops']:6.1f} TF | {ker}")
