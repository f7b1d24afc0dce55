#!/usr/bin/env python
"""What do the workgroups of attn16_long_kernel that find no query block cost?  The grid is sized for the longest sample the caller
allows (N): a compacted batch whose samples are ~ 0.3 N long launches ~ 2 empty workgroups per working one (they read offsets[b], return).
Same compacted batch, same rows, timed with N = the padded length (1800) and with N = the longest sample present."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from brepgen_amd import _lib

lib = _lib.load()
g = torch.Generator().manual_seed(0)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


for B, N, lo, hi in [(256, 1800, 300, 800), (256, 1800, 542, 542), (512, 4000, 1000, 3400), (512, 2400, 300, 1200)]:
    nvalid = torch.randint(lo, hi + 1, (B,), generator=g)
    rows = int(nvalid.sum())
    qkv = (torch.randn(rows, 2304, generator=g) * 0.7).to(torch.bfloat16).cuda()
    out = torch.empty(rows, 768, dtype=torch.bfloat16, device="cuda")
    offs = torch.zeros(B + 1, dtype=torch.int32)
    offs[1:] = torch.cumsum(nvalid, 0)
    offs = offs.cuda()
    pairs = float((nvalid.double() ** 2).sum())
    row = {"B": B, "valid": [lo, hi], "rows": rows}
    for rnd in range(2):
        for name, n_arg in (("grid_for_padded_N", N), ("grid_for_longest_sample", int(nvalid.max()))):
            us = timed(lambda: _lib.check(lib.bg_attn_varlen_fwd(qkv.data_ptr(), None, out.data_ptr(), B, n_arg, _lib.BG_BF16, offs.data_ptr(),
                                                                  _lib.stream()), "attn"))
            row.setdefault(name, {"N": n_arg, "us": []})["us"].append(round(us, 1))
    for k in ("grid_for_padded_N", "grid_for_longest_sample"):
        row[k]["tflops"] = round(4.0 * 12 * 64 * pairs / min(row[k]["us"]) / 1e6, 1)
    print(json.dumps(row), flush=True)
