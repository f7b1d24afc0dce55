#!/usr/bin/env python
"""Two builds of libbrepgen_hip.so in ONE process (brepgen_amd/_ab_old.so = the previous commit's build, copied there by hand): the
residual-stream GEMMs (bg_gemm_ex_fwd, split residual in place + statistics) on the 256 x 256 kernel alone (bg_tune key 10 = 1) and as the
library picks, bit-equality and interleaved timings.     python tools/gemm_split_ab_libs.py [M ...]"""
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
old.bg_gemm_ex_fwd.restype, old.bg_gemm_ex_fwd.argtypes = _lib._SIGNATURES["bg_gemm_ex_fwd"]
old.bg_tune_set.restype, old.bg_tune_set.argtypes = C.c_int, [C.c_int, C.c_int]
MS = [int(v) for v in sys.argv[1:]] or [18432, 30720, 61440, 138752]
dt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream


def case(M, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, 768) * 2
    hi = x.to(dt)
    lo = (x - hi.float()).to(dt)
    return (rn(M, K) * 0.5).to(dt).cuda(), (rn(768, K) * 0.04).to(dt).cuda(), rn(768).cuda(), hi.cuda(), lo.cuda()


def desc(a, w, b, hi, lo, stats, M, K):
    d = _lib.GemmDesc()
    d.a, d.lda, d.w, d.bias, d.out, d.ldc = a.data_ptr(), K, w.data_ptr(), b.data_ptr(), hi.data_ptr(), 768
    d.M, d.N, d.N_pad, d.K = M, 768, 768, K
    d.ab_dtype = d.out_dtype = _lib.BG_BF16
    d.act, d.add_div, d.add2_div = 0, 1, 1
    d.out_lo, d.res_hi, d.res_lo, d.ld_res = lo.data_ptr(), hi.data_ptr(), lo.data_ptr(), 768
    d.stats_out = stats.data_ptr()
    d.ln_eps = 1e-5
    return d


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


for mode in (1, 0):
    for lib in (old, new):
        lib.bg_tune_set(10, mode)
    print("256 x 256 kernel alone (key 10 = 1)" if mode else "library default", flush=True)
    for M in MS:
        for name, K in (("outproj", 768), ("ffn2", 1024)):
            a, w, b, hi, lo = case(M, K, seed=M + K)
            outs = []
            for lib in (old, new):
                h, l = hi.clone(), lo.clone()
                stats = torch.zeros(12, M, 2, device="cuda")
                d = desc(a, w, b, h, l, stats, M, K)
                assert lib.bg_gemm_ex_fwd(C.byref(d), st) == 0
                torch.cuda.synchronize()
                outs.append((h, l, stats))
            same = all(torch.equal(x, y) for x, y in zip(*outs))
            h, l = hi.clone(), lo.clone()
            stats = torch.zeros(12, M, 2, device="cuda")
            d = desc(a, w, b, h, l, stats, M, K)
            res = {"old": [], "new": []}
            for r in range(5):
                for nm, lib in (("old", old), ("new", new)):
                    res[nm].append(timed(lambda: lib.bg_gemm_ex_fwd(C.byref(d), st)))
            to, tn = statistics.median(res["old"]), statistics.median(res["new"])
            print(f"  M={M:6d} {name:8s} {'bit-identical' if same else 'MISMATCH'}   old {to:7.1f} us   new {tn:7.1f} us   ({tn / to:.3f})", flush=True)
for lib in (old, new):
    lib.bg_tune_set(10, 0)
