#!/usr/bin/env python
"""The fused QKV + attention kernel (csrc/qkv_attn.hip) against the two launches it replaces (LayerNorm-fold GEMM, attention):
bit-equality of the q|k|v image and of the output, with the mismatches broken down by region when there are any; then timings.

    python tools/qkv_attn_check.py [R]"""
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
lib = _lib.load()


def build(B, N, dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    M = B * N
    x = rn(M, 768) * 2
    a = x.to(dt).cuda()
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
    w = rn(2304, 768) * 0.04
    w[:768] *= 0.125
    w = w.to(dt).cuda()
    b = rn(2304).cuda()
    cs = w.float().sum(1).contiguous()
    return a, w, b, cs, stats


def two_launches(a, w, b, cs, stats, B, N):
    qkv = ops.linear_ex(a, w, b, stats_in=stats, colsum=cs)["out"]
    return ops.attention(qkv, None, B, N), qkv


bad = 0
for dt in (torch.bfloat16, torch.float16):
    for B, N in ((4, 60), (7, 60), (64, 60), (512, 60), (8, 30), (13, 30), (512, 30), (5, 64), (9, 32), (6, 2), (6, 34), (300, 48)):
        a, w, b, cs, stats = build(B, N, dt, seed=B * 100 + N)
        ref, qkv = two_launches(a, w, b, cs, stats, B, N)
        for rep in range(3):
            out, img = ops.qkv_attention(a, w, b, cs, stats, B, N, want_qkv=True)
            out2 = ops.qkv_attention(a, w, b, cs, stats, B, N)
            torch.cuda.synchronize()
            ok_img, ok_out, ok2 = torch.equal(img, qkv), torch.equal(out, ref), torch.equal(out2, ref)
            if not (ok_img and ok_out and ok2):
                bad += 1
                d = (img.float() != qkv.float()).reshape(B, N, 3, 12, 64)
                do = (out.float() != ref.float()).reshape(B, N, 12, 64)
                do2 = (out2.float() != ref.float()).reshape(B, N, 12, 64)
                print(f"MISMATCH {str(dt)[6:]} B={B} N={N} rep={rep}: image {int(d.sum())} of {d.numel()} (q {int(d[:, :, 0].sum())} k {int(d[:, :, 1].sum())} "
                      f"v {int(d[:, :, 2].sum())}; by head {d.sum((0, 1, 2, 4)).tolist()}; by token {d.sum((0, 2, 3, 4)).tolist()[:64]}; "
                      f"samples with errors {int(d.any(-1).any(-1).any(-1).any(-1).sum())}); out {int(do.sum())} of {do.numel()} "
                      f"(by head {do.sum((0, 1, 3)).tolist()}; by token {do.sum((0, 2, 3)).tolist()[:64]}; max |diff| {float((out.float() - ref.float()).abs().max()):.3e}); "
                      f"out (no image) {int(do2.sum())}, nan {int(torch.isnan(out.float()).sum())}", flush=True)
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


dt = torch.bfloat16
for B, N in ((512, 60), (256, 60), (512, 30), (256, 30), (1024, 60)):
    a, w, b, cs, stats = build(B, N, dt)
    M = B * N
    qkv = ops.linear_ex(a, w, b, stats_in=stats, colsum=cs)["out"]
    t_g = statistics.median(timed(lambda: ops.linear_ex(a, w, b, stats_in=stats, colsum=cs)) for _ in range(R))
    t_a = statistics.median(timed(lambda: ops.attention(qkv, None, B, N)) for _ in range(R))
    t_f = statistics.median(timed(lambda: ops.qkv_attention(a, w, b, cs, stats, B, N)) for _ in range(R))
    fl = 2.0 * M * 768 * 2304
    print(f"B={B:5d} N={N:3d} (M={M:6d}): GEMM {t_g:7.1f} us ({fl / t_g / 1e6:5.0f} TF) + attention {t_a:6.1f} us = {t_g + t_a:7.1f} us;  "
          f"fused {t_f:7.1f} us ({fl / t_f / 1e6:5.0f} TF on the GEMM FLOPs)  ratio {t_f / (t_g + t_a):.3f}", flush=True)


# ---- operand data and the clock: the same launches on random, constant and zero operands (a kernel that is bound by what the part
# may draw, not by its schedule, runs faster the fewer bits toggle) ----
B, N = 512, 60
a, w, b, cs, stats = build(B, N, dt)
M = B * N
for name, aa, ww in (("random operands", a, w), ("A = 1.0, W random", torch.ones_like(a), w), ("A = 0, W = 0", torch.zeros_like(a), torch.zeros_like(w))):
    t_f = statistics.median(timed(lambda: ops.qkv_attention(aa, ww, b, cs, stats, B, N), n=40) for _ in range(R))
    t_g = statistics.median(timed(lambda: ops.linear_ex(aa, ww, b, stats_in=stats, colsum=cs), n=40) for _ in range(R))
    t_p = statistics.median(timed(lambda: ops.linear(aa, ww, b, out_dtype=dt), n=40) for _ in range(R))
    print(f"{name:20s}: fused {t_f:7.1f} us   256 x 256 GEMM, LayerNorm fold {t_g:7.1f} us, plain {t_p:7.1f} us", flush=True)
