#!/usr/bin/env python
"""In-process A/B of bg_tune settings on the three face-LDM loops of bench.py (one box, one process, interleaved rounds):
    python tools/face_ldm_ab.py "10=2" "10=0" ...        each argument = one setting (key=value,...); BG_SPLITS=1,2,4: the n_split values
A key that is not a number names an attribute of the two drop-in modules instead (e.g. "fuse_output=0,time_table_steps=0")."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from brepgen_amd import _lib

SETTINGS = sys.argv[1:] or ["10=2", "10=0"]
KEYS = sorted({int(kv.split("=")[0]) for s in SETTINGS for kv in s.split(",") if kv and kv.split("=")[0].isdigit()})
ATTRS = sorted({kv.split("=")[0] for s in SETTINGS for kv in s.split(",") if kv and not kv.split("=")[0].isdigit()})
lib = _lib.load()
dev = torch.device("cuda")
ldm = bench.FaceLDM(dev, 0)
STEPS, ROUNDS = 20, 3
DEFAULTS = {a: getattr(ldm.pos_net, a) for a in ATTRS}


def apply(setting):
    for k in KEYS:
        lib.bg_tune_set(k, 0)
    for a, d in DEFAULTS.items():
        for net in (ldm.pos_net, ldm.z_net):
            setattr(net, a, d)
    for kv in filter(None, setting.split(",")):
        k, v = kv.split("=")
        if k.isdigit():
            lib.bg_tune_set(int(k), int(v))
        else:
            for net in (ldm.pos_net, ldm.z_net):
                setattr(net, k, type(DEFAULTS[k])(int(v)))


def clock(ks):
    ldm.run(*[1 if k else 0 for k in ks])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ldm.run(*ks)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / sum(ks) * 1e3


for split in [int(v) for v in os.environ.get("BG_SPLITS", "1,2").split(",")]:
    ldm.set_split(split)
    res = {(s, leg): [] for s in SETTINGS for leg in "ABC"}
    for r in range(ROUNDS):
        for s in SETTINGS:
            apply(s)
            for leg, ks in zip("ABC", ((STEPS, 0, 0), (0, STEPS, 0), (0, 0, STEPS))):
                res[(s, leg)].append(clock(ks))
    print(f"n_split = {split}   (ms per step, median of {ROUNDS} rounds x {STEPS} steps; composite = (158 A + 250 B + 209 C) / 617)")
    for s in SETTINGS:
        m = {leg: statistics.median(res[(s, leg)]) for leg in "ABC"}
        comp = (158 * m["A"] + 250 * m["B"] + 209 * m["C"]) / 617
        print(f"  {s:24s} A {m['A']:.3f}  B {m['B']:.3f}  C {m['C']:.3f}  composite {comp:.3f}")
apply("")
