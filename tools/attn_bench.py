#!/usr/bin/env python
"""Attention micro-benchmark: round-1 kernel (bg_tune key 6 = 1) vs the long-sequence kernel, interleaved in one process
(dense / ragged key mask / compacted variable-length batch), at the edge-net sizes of BASELINE configs[2..4]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops

lib = _lib.load()
g = torch.Generator().manual_seed(0)


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


rows = []
for B, N, dt in [(256, 1800, torch.bfloat16), (64, 4000, torch.bfloat16), (128, 2400, torch.float16), (512, 60, torch.bfloat16), (512, 128, torch.bfloat16)]:
    qkv = (torch.randn(B * N, 2304, generator=g) * 0.7).to(dt).cuda()
    out = torch.empty(B * N, 768, dtype=dt, device="cuda")
    nvalid = torch.randint(max(1, N // 8), N + 1, (B,), generator=g)
    mask = (torch.arange(N)[None] >= nvalid[:, None]).cuda()
    offs = torch.zeros(B + 1, dtype=torch.int32)
    offs[1:] = torch.cumsum(nvalid, 0)
    offs = offs.cuda()
    code = {torch.bfloat16: _lib.BG_BF16, torch.float16: _lib.BG_F16}[dt]
    call = lambda kp, of: _lib.check(lib.bg_attn_varlen_fwd(qkv.data_ptr(), kp, out.data_ptr(), B, N, code, of, _lib.stream()), "attn")
    cases = {"dense": (None, None, float(B) * N * N), "ragged-mask": (mask.view(torch.uint8).data_ptr(), None, float(B) * N * N),
             "varlen": (None, offs.data_ptr(), float((nvalid.double() ** 2).sum()))}
    for name, (kp, of, pairs) in cases.items():
        res = {}
        for rnd in range(2):
            # long kernel: (sample, head) units round-robin over the XCDs (the library's walk) vs a contiguous eighth per XCD (rounds 2-5;
            # bg_tune key 20), alternated in this process
            for name_, key in (("round_robin", 0), ("eighths", 1)) if N > 64 else (("new", 0),):
                lib.bg_tune_set(20, key)
                us = timed(lambda: call(kp, of), 5 if N > 64 else 20)
                res.setdefault(name_, []).append(us)
            lib.bg_tune_set(20, 0)
        row = {"B": B, "N": N, "dtype": str(dt)[6:], "case": name,
               **{k: {"us": round(min(v), 1), "tflops_executed": round(4.0 * 12 * 64 * pairs / min(v) / 1e6, 1)} for k, v in res.items()}}
        rows.append(row)
        print(json.dumps(row), flush=True)
