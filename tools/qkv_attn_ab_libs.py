#!/usr/bin/env python
"""Two builds of libbrepgen_hip.so in ONE process (brepgen_amd/_ab_old.so = the previous commit's build, copied there by hand): the fused
QKV + attention launch of both, bit-equality and interleaved timings.     python tools/qkv_attn_ab_libs.py"""
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from brepgen_amd import _lib

new = _lib.load()
old = C.CDLL(os.path.join(ROOT, "brepgen_amd", "_ab_old.so"))
for lib in (old,):
    res, args = _lib._SIGNATURES["bg_qkv_attn_fwd"]
    lib.bg_qkv_attn_fwd.restype, lib.bg_qkv_attn_fwd.argtypes = res, args
    lib.bg_tune_set.restype, lib.bg_tune_set.argtypes = C.c_int, [C.c_int, C.c_int]
dt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream


def build(B, N, dt, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    M = B * N
    x = rn(M, 768) * 2
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
    w = rn(2304, 768) * 0.04
    w[:768] *= 0.125
    w = w.to(dt).cuda()
    return x.to(dt).cuda(), w, rn(2304).cuda(), w.float().sum(1).contiguous(), stats


def run(lib, a, w, b, cs, stats, out, B, N, code):
    rc = lib.bg_qkv_attn_fwd(a.data_ptr(), w.data_ptr(), b.data_ptr(), cs.data_ptr(), stats.data_ptr(), None, out.data_ptr(), None, B, N, code, 1e-5, st)
    assert rc == 0, rc


def timed(fn, n=40):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dtt, code in ((torch.bfloat16, _lib.BG_BF16), (torch.float16, _lib.BG_F16)):
    for B, N in ((512, 60), (512, 30), (300, 48), (37, 60), (9, 64), (6, 2)):
        a, w, b, cs, stats = build(B, N, dtt, seed=B + N)
        o1 = torch.zeros(B * N, 768, dtype=dtt, device="cuda")
        o2 = torch.zeros_like(o1)
        run(old, a, w, b, cs, stats, o1, B, N, code)
        run(new, a, w, b, cs, stats, o2, B, N, code)
        torch.cuda.synchronize()
        same = torch.equal(o1, o2)
        to = statistics.median(timed(lambda: run(old, a, w, b, cs, stats, o1, B, N, code)) for _ in range(5))
        tn = statistics.median(timed(lambda: run(new, a, w, b, cs, stats, o2, B, N, code)) for _ in range(5))
        to2 = statistics.median(timed(lambda: run(old, a, w, b, cs, stats, o1, B, N, code)) for _ in range(5))
        print(f"{str(dtt)[6:]:9s} B={B:4d} N={N:2d}: {'bit-identical' if same else 'MISMATCH ' + str(int((o1.float() != o2.float()).sum()))}   old {to:7.1f} / {to2:7.1f} us   new {tn:7.1f} us   ({tn / min(to, to2):.3f})", flush=True)
