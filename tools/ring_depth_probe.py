#!/usr/bin/env python
"""EXPERIMENT (round 5): does a deeper LDS-DMA ring lift the K loop of the 128-row tiles?  The lock-step generic kernel
(gemm_16bit.hip: gemm16_kernel<BM, BN, WM, WN, STAGES>) in five shapes (bg_tune key 0), plain 16-bit output and the split-residual
epilogue, at the face-LDM size and at a long K where the K loop is all there is."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import hip_ops as ops
from brepgen_amd import _lib

lib = _lib.load()
dt = torch.bfloat16
NAMES = {0: "shipped kernels", 5: "128x128 2 slots (generic)", 1: "128x128 3 slots", 2: "128x128 4 slots", 4: "256x128 2 slots", 3: "256x128 3 slots"}


def timed(fn, n=30):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in ((30720, 768, 768), (30720, 768, 1024), (30720, 768, 4096), (30720, 2304, 768), (138752, 768, 768)):
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda()
    w = (torch.randn(N, K, generator=g) * 0.04).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    x = torch.randn(M, N, generator=g) * 2
    hi = x.to(dt).cuda()
    lo = (x - x.to(dt).float()).to(dt).cuda()
    refs = {}
    for v in (0, 5, 1, 2, 4, 3):
        lib.bg_tune_set(0, v)
        plain = ops.linear(a, w, b, out_dtype=dt)
        sp = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True) if N == 768 else None
        torch.cuda.synchronize()
        if v == 0:
            refs = {"plain": plain.clone(), "hi": sp["out"].clone() if sp else None, "lo": sp["lo"].clone() if sp else None}
        ok = torch.equal(plain, refs["plain"]) and (sp is None or (torch.equal(sp["out"], refs["hi"]) and torch.equal(sp["lo"], refs["lo"])))
        tp = statistics.median(timed(lambda: ops.linear(a, w, b, out_dtype=dt)) for _ in range(3))
        ts = statistics.median(timed(lambda: ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)) for _ in range(3)) if N == 768 else float("nan")
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:4d} K={K:4d}  {NAMES[v]:28s} plain {tp:7.1f} us {fl / tp / 1e6:6.0f} TF   split {ts:7.1f} us {fl / ts / 1e6:6.0f} TF   {'bit-identical' if ok else 'MISMATCH'}", flush=True)
lib.bg_tune_set(0, 0)
