#!/usr/bin/env python
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity_cases as pc
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
m, _ = pc.build_net("SurfZNet", 1, False, torch.bfloat16)
m.cache_conditioning = False
args = [t.cuda() if torch.is_tensor(t) else t for t in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
with torch.no_grad():
    print("B=512 one call:", timed(lambda: m(*args)), "us")
    for nb in (2, 4, 8):
        sz = 512 // nb
        parts = [[a[i*sz:(i+1)*sz].contiguous() if torch.is_tensor(a) and a.dim() > 1 else a for a in args] for i in range(nb)]
        print(f"{nb} micro-batches of {sz}:", timed(lambda: [m(*p) for p in parts]), "us")
