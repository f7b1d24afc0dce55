"""Per-epilogue-mode timing of the persistent 16-bit GEMM on the four per-layer shapes of BASELINE configs[1]
(M = 30 720): plain 16-bit output vs LayerNorm fold (QKV, FFN1); fp32 read-modify-write residual vs split (hi, lo)
residual with row statistics (out-proj, FFN2).  Modes are interleaved inside one process; medians of R rounds.

    python tools/gemm_mode_bench.py [R] [M] [stagger,stagger,...]     (stagger: bg_tune key 8, see gemm_16bit.hip)
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 7
M = int(sys.argv[2]) if len(sys.argv) > 2 else 30720
STAGGERS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [None]
dt = torch.bfloat16
dev = "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)


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


x = rn(M, 768) * 2
hi = x.to(dt).to(dev)
lo = (x - x.to(dt).float()).to(dt).to(dev)
xf = x.to(dev)
grp = x.reshape(M, 12, 64)
stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().to(dev)
a768 = (rn(M, 768) * 0.5).to(dt).to(dev)
a1024 = (rn(M, 1024) * 0.5).to(dt).to(dev)
cases = {}
for name, N, K, a in (("qkv", 2304, 768, hi), ("ffn1", 1024, 768, hi)):
    w = (rn(N, K) * 0.04).to(dt).to(dev)
    b = rn(N).to(dev)
    cs = w.float().sum(1).contiguous()
    act = 1 if name == "ffn1" else 0
    cases[name + " plain"] = (lambda a=a, w=w, b=b, act=act: ops.linear(a, w, b, out_dtype=dt, act=act), N, K)
    cases[name + " fold "] = (lambda a=a, w=w, b=b, act=act, cs=cs: ops.linear_ex(a, w, b, act=act, stats_in=stats, colsum=cs), N, K)
for name, N, K, a in (("outproj", 768, 768, a768), ("ffn2", 768, 1024, a1024)):
    w = (rn(N, K) * 0.04).to(dt).to(dev)
    b = rn(N).to(dev)
    cases[name + " fp32-rmw"] = (lambda a=a, w=w, b=b: ops.linear(a, w, b, out_dtype=torch.float32, add=xf, out=xf), N, K)
    cases[name + " split   "] = (lambda a=a, w=w, b=b: ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True), N, K)
    cases[name + " split-ns"] = (lambda a=a, w=w, b=b: ops.linear_ex(a, w, b, split_out=True, res=(hi, lo)), N, K)

res = {(k, st): [] for k in cases for st in STAGGERS}
for r in range(R):
    for st in STAGGERS:
        if st is not None:
            _lib.load().bg_tune_set(8, st)
        for k, (fn, N, K) in cases.items():
            res[(k, st)].append(timed(fn))
print(f"M = {M}")
for k, (fn, N, K) in cases.items():
    for st in STAGGERS:
        us = statistics.median(res[(k, st)])
        tag = "" if st is None else f" stagger {st:4d}"
        print(f"{k:20s}{tag} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF   (min {min(res[(k, st)]):.1f})")
