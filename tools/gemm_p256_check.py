#!/usr/bin/env python
"""The 256 x 256 persistent 8-phase GEMM (csrc/gemm_p256.hip) against the 128 x 128 persistent kernel: bit-equality of the
plain / LayerNorm-fold epilogues on ragged row counts (bf16 and fp16), then interleaved timings on the per-layer shapes and on
square problems.   python tools/gemm_p256_check.py [M ...]      (bg_tune key 10: 0 = the 256 + 128 hybrid the library picks,
1 = force the 256 kernel, 2 = never)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops

MS = [int(v) for v in sys.argv[1:]] or [8640, 17280, 30720, 61440, 138752]
lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)


def setv(mode):
    lib.bg_tune_set(10, mode)


TIMING = False


def build(M, dt):
    x = rn(M, 768) * 2
    hi = x.to(dt).to(dev)
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().to(dev)
    cases = {}
    for name, N, K, a in (("qkv", 2304, 768, hi), ("ffn1", 1024, 768, hi)):
        w, b = (rn(N, K) * 0.04).to(dt).to(dev), rn(N).to(dev)
        cs = w.float().sum(1).contiguous()
        act = 1 if name == "ffn1" else 0
        cases[name + " plain"] = (lambda a=a, w=w, b=b, act=act: ops.linear(a, w, b, out_dtype=dt, act=act), N, K)
        cases[name + " nobias"] = (lambda a=a, w=w: ops.linear(a, w, None, out_dtype=dt), N, K)
        cases[name + " fold"] = (lambda a=a, w=w, b=b, act=act, cs=cs: ops.linear_ex(a, w, b, act=act, stats_in=stats, colsum=cs)["out"], N, K)
    lo = (x - x.to(dt).float()).to(dt).to(dev)
    a1024 = (rn(M, 1024) * 0.5).to(dt).to(dev)
    for name, N, K, a in (("outproj", 768, 768, hi), ("ffn2", 768, 1024, a1024)):
        w, b = (rn(N, K) * 0.04).to(dt).to(dev), rn(N).to(dev)

        def split(a=a, w=w, b=b, st=True):
            r = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=st)
            if TIMING:
                return r["out"]
            return torch.cat([r["out"].float().flatten(), r["lo"].float().flatten()] + ([r["stats"].flatten()] if st else []))
        cases[name + " split"] = (split, N, K)
        cases[name + " split-ns"] = (lambda f=split: f(st=False), N, K)
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


bad = 0
for dt in (torch.bfloat16, torch.float16):
    for M in (1037, 1038, 256 * 7, 4999, 17294, 30720 + 78):
        for k, (fn, N, K) in build(M, dt).items():
            lib.bg_tune_set(10, 2)
            ref = fn().clone()
            res = []
            for mode in (1, 0):
                setv(mode)
                for rep in range(3):                                  # repeated: a race would not necessarily show the first time
                    got = fn()
                    torch.cuda.synchronize()
                    res.append(torch.equal(ref, got))
            ok = all(res)
            bad += not ok
            if not ok or M == 1038:
                nd = (ref != got).sum().item()
                print(f"bit-equal {str(dt)[6:]:9s} M={M:5d} {k:12s} {ok} {res if not ok else ''} {'differing elements: %d' % nd if not ok else ''}")
print("BIT-EQUALITY", "OK" if bad == 0 else f"FAILED ({bad} cases)")
TIMING = True
VARS = [("128", 2), ("256", 1), ("hybrid", 0)]
for M in MS:
    cases = build(M, dt)
    res = {(k, v[0]): [] for k in cases for v in VARS}
    for r in range(5):
        for name, *kv in VARS:
            setv(*kv)
            for k, (fn, N, K) in cases.items():
                res[(k, name)].append(timed(fn))
    print(f"M = {M}")
    for k, (fn, N, K) in cases.items():
        line = f"  {k:12s}"
        for name, *kv in VARS:
            us = statistics.median(res[(k, name)])
            line += f" | {name} {us:6.1f} {2.0 * M * N * K / us / 1e6:4.0f}"
        print(line)
for S in (4096, 8192):
    a = (rn(S, S) * 0.5).to(dt).to(dev)
    w = (rn(S, S) * 0.04).to(dt).to(dev)
    b = rn(S).to(dev)
    fn = lambda: ops.linear(a, w, b, out_dtype=dt)
    line = f"square {S}^3 plain:"
    for name, *kv in VARS:
        setv(*kv)
        us = statistics.median(timed(fn, 10) for _ in range(3))
        line += f" | {name} {us:7.1f} {2.0 * S * S * S / us / 1e6:4.0f}"
    print(line)
setv(0)
