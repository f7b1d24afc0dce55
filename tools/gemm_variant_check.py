#!/usr/bin/env python
"""A/B of a 16-bit GEMM kernel variant (bg_tune key 0) against the shipped persistent kernel: bit-equality of every
epilogue mode on a ragged row count, then timings on the per-layer shapes.   python tools/gemm_variant_check.py VARIANT [M ...]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops

VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 6
MS = [int(v) for v in sys.argv[2:]] or [17280, 138752]
lib = _lib.load()
dt, dev = torch.bfloat16, "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)


def build(M):
    x = rn(M, 768) * 2
    hi = x.to(dt).to(dev)
    lo = (x - x.to(dt).float()).to(dt).to(dev)
    xf = x.to(dev)
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().to(dev)
    a1024 = (rn(M, 1024) * 0.5).to(dt).to(dev)
    cases = {}
    for name, N, K, a in (("qkv", 2304, 768, hi), ("ffn1", 1024, 768, hi)):
        w, b = (rn(N, K) * 0.04).to(dt).to(dev), rn(N).to(dev)
        cs = w.float().sum(1).contiguous()
        act = 1 if name == "ffn1" else 0
        cases[name + " plain"] = (lambda a=a, w=w, b=b, act=act: (ops.linear(a, w, b, out_dtype=dt, act=act),), N, K)
        cases[name + " fold"] = (lambda a=a, w=w, b=b, act=act, cs=cs: (ops.linear_ex(a, w, b, act=act, stats_in=stats, colsum=cs)["out"],), N, K)
    for name, N, K, a in (("outproj", 768, 768, hi), ("ffn2", 768, 1024, a1024)):
        w, b = (rn(N, K) * 0.04).to(dt).to(dev), rn(N).to(dev)

        def split(a=a, w=w, b=b):
            r = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)
            return r["out"], r["lo"], r["stats"]
        cases[name + " split"] = (split, N, K)
        cases[name + " fp32+add"] = (lambda a=a, w=w, b=b: (ops.linear(a, w, b, out_dtype=torch.float32, add=xf),), N, K)
    return cases


def timed(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# ---- bit-equality on a ragged row count ----
cases = build(1000 + 37)
for k, (fn, N, K) in cases.items():
    lib.bg_tune_set(0, 0)
    ref = [t.clone() for t in fn()]
    lib.bg_tune_set(0, VAR)
    got = fn()
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(ref, got))
    print(f"bit-equal {k:18s} {ok}")
# ---- timings ----
for M in MS:
    cases = build(M)
    res = {(k, v): [] for k in cases for v in (0, VAR)}
    for r in range(5):
        for v in (0, VAR):
            lib.bg_tune_set(0, v)
            for k, (fn, N, K) in cases.items():
                res[(k, v)].append(timed(fn))
    print(f"M = {M}")
    for k, (fn, N, K) in cases.items():
        a, b = statistics.median(res[(k, 0)]), statistics.median(res[(k, VAR)])
        print(f"  {k:18s} shipped {a:8.1f} us {2.0 * M * N * K / a / 1e6:6.0f} TF | variant {VAR} {b:8.1f} us {2.0 * M * N * K / b / 1e6:6.0f} TF  ({a / b:.2f}x)")
lib.bg_tune_set(0, 0)
# ---- K-loop quality away from the per-tile overhead: square problems (the guide's 8-phase template reaches ~1330 / ~1470 TF
# at 4096^3 / 8192^3 on uniform random operands) ----
if os.environ.get("BG_SQUARE", "1") == "1":
    for S in (4096, 8192):
        a = (rn(S, S) * 0.5).to(dt).to(dev)
        w = (rn(S, S) * 0.04).to(dt).to(dev)
        b = rn(S).to(dev)
        fn = lambda: ops.linear(a, w, b, out_dtype=dt)
        line = f"square {S}^3 plain:"
        for v in (0, 6, VAR):
            lib.bg_tune_set(0, v)
            us = statistics.median(timed(fn, 10) for _ in range(3))
            line += f"  variant {v}: {us:8.1f} us {2.0 * S * S * S / us / 1e6:6.0f} TF"
        print(line)
    lib.bg_tune_set(0, 0)
