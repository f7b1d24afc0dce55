#!/usr/bin/env python
"""The 256 x 128 / three-slot residual-stream GEMM (csrc/gemm_split3.hip, bg_tune key 12 = 3) against the pipelined 128 x 128
kernel (12 = 0, with the 256 kernel off: 10 = 2) and the library's default choice: bit-equality (out of place, in place, repeated), then
interleaved timings.      python tools/split3_check.py [R] [M ...]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import hip_ops as ops
from brepgen_amd import _lib

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MS = [int(v) for v in sys.argv[2:]] or [15360, 17280, 18432, 30720, 61440, 138752]
lib = _lib.load()


def setv(kv):
    for k in (8, 10, 12, 15):
        lib.bg_tune_set(k, 0)
    for k, v in kv.items():
        lib.bg_tune_set(k, v)


def case(M, K, dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, 768) * 2
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    return (rn(M, K) * 0.5).to(dt).cuda(), (rn(768, K) * 0.04).to(dt).cuda(), rn(768).cuda(), hi.cuda(), lo.cuda()


def run(a, w, b, hi, lo, inplace):
    if inplace:
        h, l = hi.clone(), lo.clone()
        r = ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)
    else:
        r = ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)
    torch.cuda.synchronize()
    return r["out"].clone(), r["lo"].clone(), r["stats"].clone()


bad = 0
for dt in (torch.bfloat16, torch.float16):
    for K in (768, 1024):
        for M in (1409, 256 * 5, 4999, 17294, 30720 + 78, 30720):
            a, w, b, hi, lo = case(M, K, dt, seed=M + K)
            setv({12: 0, 10: 2, 15: -1})
            ref = run(a, w, b, hi, lo, False)
            setv({12: 3, 10: 2, 15: -1})
            for rep in range(3):
                for inplace in (False, True):
                    got = run(a, w, b, hi, lo, inplace)
                    ok = [torch.equal(x, y) for x, y in zip(ref, got)]
                    if not all(ok):
                        bad += 1
                        d = (ref[0].float() != got[0].float())
                        rows = d.any(1).nonzero().flatten()
                        print(f"MISMATCH {str(dt)[6:]} K={K} M={M} inplace={inplace} rep={rep}: hi {int(d.sum())} lo {int((ref[1].float() != got[1].float()).sum())} "
                              f"stats {int((ref[2] != got[2]).sum())}; rows {rows[:12].tolist()} ... {rows[-4:].tolist()} ({rows.numel()} rows); "
                              f"cols {d.any(0).nonzero().flatten()[:12].tolist()}; nan {int(torch.isnan(got[0].float()).sum())}", flush=True)
                        break
                else:
                    continue
                break
print("BIT-EQUALITY", "OK" if bad == 0 else f"FAILED ({bad} cases)", flush=True)


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


VARS = [("pipe 128x128", {12: 0, 10: 2}), ("library default", {}), ("split3 256x128", {12: 3, 10: 2}), ("split3, NO lo plane", {12: 4, 10: 2})]
dt = torch.bfloat16
for M in MS:
    for name, K in (("outproj", 768), ("ffn2", 1024)):
        a, w, b, hi, lo = case(M, K, dt)
        h, l = hi.clone(), lo.clone()
        res = {v: [] for v, _ in VARS}
        for r in range(R):
            for v, kv in VARS:
                setv(kv)
                res[v].append(timed(lambda: ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True)))
        fl = 2.0 * M * 768 * K
        by = M * (2.0 * K + 8.0 * 768 + 96) + 2.0 * 768 * K
        print(f"M={M:6d} {name:8s} " + "  ".join(f"{v}: {statistics.median(t):6.1f} us {fl / statistics.median(t) / 1e6:5.0f} TF {by / statistics.median(t) / 1e3:5.0f} GB/s" for v, t in res.items()), flush=True)
# the same launches on zero operands (no bit toggles: the part runs 2.4 GHz -- what the SCHEDULES are worth, without the power cap)
for M in (30720,):
    for name, K in (("outproj", 768), ("ffn2", 1024)):
        a, w, b, hi, lo = case(M, K, dt)
        a, w, b, h, l = torch.zeros_like(a), torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(hi), torch.zeros_like(lo)
        res = {v: [] for v, _ in VARS}
        for r in range(R):
            for v, kv in VARS:
                setv(kv)
                res[v].append(timed(lambda: ops.linear_ex(a, w, b, split_out=True, res=(h, l), want_stats=True, inplace=True), n=60))
        print(f"ZEROS M={M:6d} {name:8s} " + "  ".join(f"{v}: {statistics.median(t):6.1f} us" for v, t in res.items()), flush=True)
setv({})
