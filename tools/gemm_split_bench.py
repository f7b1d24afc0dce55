#!/usr/bin/env python
"""Residual-stream GEMMs (out-proj K = 768, FFN2 K = 1024; split residual in place + row statistics): the software-pipelined
kernel (csrc/gemm_split.hip, bg_tune key 12 = 0) against what ran them before -- the 256 + 128 hybrid (12 = 1) and the 128 x 128
kernel alone (12 = 1, 10 = 2).  Bit-equality first, then interleaved timings (medians of R rounds of 20 launches), with the
algorithmic bytes per launch (A + W + residual in (hi, lo) + (hi, lo) out + statistics) as GB/s next to the TFLOP/s.

    python tools/gemm_split_bench.py [R] [M ...]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MS = [int(v) for v in sys.argv[2:]] or [15360, 17280, 30720, 61440, 138752, 204800]
lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)
STAGS = [int(v) for v in os.environ.get("BG_STAGS", "12,18,24,30,36").split(",")]
# 128: the 128 x 128 persistent kernel alone; pipe: the pipelined 128 x 128 kernel alone; p256 ...: the 256 x 256 kernel alone, without /
# with the second phase group starting D x 1024 cycles late; hybrid: what the library picks (bg_common.h p256_rows)
VARS = [("128", {12: 1, 10: 2, 8: 0}), ("pipe", {12: 0, 10: 2, 8: 0}), ("p256 nostag", {12: 0, 10: 1, 8: -1})] + \
       [(f"p256 stag{d}", {12: 0, 10: 1, 8: d}) for d in STAGS] + [("hybrid", {12: 0, 10: 0, 8: 0})]


def setv(kv):
    for k, v in kv.items():
        lib.bg_tune_set(k, v)


def build(M, dt):
    x = rn(M, 768) * 2
    hi = x.to(dt).to(dev)
    lo = (x - x.to(dt).float()).to(dt).to(dev)
    cases = {}
    for name, K in (("outproj", 768), ("ffn2", 1024)):
        a = (rn(M, K) * 0.5).to(dt).to(dev)
        w, b = (rn(768, K) * 0.04).to(dt).to(dev), rn(768).to(dev)
        h, l = hi.clone(), lo.clone()                             # the in-place planes of the timed launches
        cases[name] = (a, w, b, h, l, K)
    return hi, lo, cases


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
    for M in (1409, 4999, 17294, 30720 + 78, 138752 + 5):
        hi, lo, cases = build(M, dt)
        for name, (a, w, b, _, _, K) in cases.items():
            outs = {}
            for vn, kv in VARS:
                setv(kv)
                res = []
                for rep in range(3):
                    h, l = hi.clone(), lo.clone()
                    r = ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)
                    torch.cuda.synchronize()
                    res.append(torch.cat([r["out"].float().flatten(), r["lo"].float().flatten(), r["stats"].flatten()]))
                outs[vn] = res
            ok = all(torch.equal(outs["128"][0], t) for v in outs.values() for t in v)
            bad += not ok
            if not ok or M == 4999:
                nd = {vn: [int((outs["128"][0] != t).sum()) for t in v] for vn, v in outs.items()}
                print(f"bit-equal {str(dt)[6:]:9s} M={M:6d} {name:8s} {ok} {nd if not ok else ''}")
print("BIT-EQUALITY", "OK" if bad == 0 else f"FAILED ({bad} cases)", flush=True)

dt = torch.bfloat16
for M in MS:
    hi, lo, cases = build(M, dt)
    res = {(k, v[0]): [] for k in cases for v in VARS}
    for r in range(R):
        for vn, kv in VARS:
            setv(kv)
            for k, (a, w, b, h, l, K) in cases.items():
                fn = lambda a=a, w=w, b=b, h=h, l=l: ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)
                res[(k, vn)].append(timed(fn))
    print(f"M = {M}")
    for k, (a, w, b, h, l, K) in cases.items():      # the same product WITHOUT the residual: 16-bit output, K loop + a light epilogue
        for vn, kv in (("128", {10: 2}), ("256+128", {10: 0})):
            setv(kv)
            us = statistics.median(timed(lambda: ops.linear(a, w, b, out_dtype=dt)) for _ in range(R))
            print(f"  {k:8s} plain 16-bit output, {vn:8s} {us:7.1f} us {2.0 * M * 768 * K / us / 1e6:5.0f} TF")
    for k, (a, w, b, h, l, K) in cases.items():
        by = 2.0 * M * K + 2.0 * 768 * K + 8.0 * M * 768 + 8.0 * M * 12 + 4 * 768
        print(f"  {k:8s} ({by / 1e6:6.1f} MB algorithmic)")
        for vn, _ in VARS:
            us = statistics.median(res[(k, vn)])
            print(f"      {vn:14s} {us:7.1f} us {2.0 * M * 768 * K / us / 1e6:5.0f} TF {by / us / 1e3:5.0f} GB/s", flush=True)
setv({12: 0, 10: 0, 8: 0})
