#!/usr/bin/env python
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from brepgen_amd import ops
g = torch.Generator().manual_seed(0)
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters
for B, N in [(512, 60), (512, 30), (8, 1800), (2, 4000), (16, 2400)]:
    qkv = torch.randn(B * N, 2304, generator=g).cuda().to(torch.bfloat16)
    mask = torch.zeros(B, N, dtype=torch.bool).cuda()
    us = timed(lambda: ops.attention(qkv, mask, B, N), iters=10)
    print(f"attn bf16 B={B} N={N}: {us:.1f} us  {4.0 * B * 12 * N * N * 64 / us / 1e6:.1f} TFLOP/s", flush=True)
