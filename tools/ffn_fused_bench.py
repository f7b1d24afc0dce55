"""Isolated timing of the fused FFN launch (csrc/ffn_fused.hip) against the two GEMM launches it replaces, same operands, same process.
    python tools/ffn_fused_bench.py [M ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hip_ops as ops  # noqa: E402
from brepgen_amd import _lib  # noqa: E402
from brepgen_amd._lib import check, ptr, stream  # noqa: E402
from brepgen_amd.network import ffn_fragment_order  # noqa: E402
from test_gpu_round6 import _ffn_operands  # noqa: E402


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.load()
    for dt in (torch.bfloat16,):
        for M in [int(a) for a in sys.argv[1:]] or [15360, 18432, 30720, 61440, 138752]:
            hi, lo, stats, w1, b1, colsum1, w2, b2 = _ffn_operands(M, dt, 1)
            w1f, w2f = ffn_fragment_order(w1, 4), ffn_fragment_order(w2, 3)
            code = ops.bg_dtype(dt)
            h = torch.empty(M, 1024, device="cuda", dtype=dt)

            def fused():
                check(lib.bg_ffn_fused_fwd(ptr(hi), ptr(lo), ptr(stats), ptr(w1f), ptr(b1), ptr(colsum1), ptr(w2f), ptr(b2), M, M, None, code,
                                           1e-5, stream()), "ffn")

            d1 = _lib.GemmDesc()
            d1.a, d1.lda, d1.w, d1.bias, d1.out, d1.ldc = ptr(hi), 768, ptr(w1), ptr(b1), ptr(h), 1024
            d1.M, d1.N, d1.N_pad, d1.K = M, 1024, 1024, 768
            d1.ab_dtype, d1.out_dtype, d1.act = code, code, 1
            d1.stats_in, d1.colsum, d1.ln_eps = ptr(stats), ptr(colsum1), 1e-5
            d2 = _lib.GemmDesc()
            d2.a, d2.lda, d2.w, d2.bias, d2.out, d2.ldc = ptr(h), 1024, ptr(w2), ptr(b2), ptr(hi), 768
            d2.M, d2.N, d2.N_pad, d2.K = M, 768, 768, 1024
            d2.ab_dtype, d2.out_dtype, d2.act = code, code, 0
            d2.out_lo, d2.res_hi, d2.res_lo, d2.ld_res, d2.stats_out, d2.ln_eps = ptr(lo), ptr(hi), ptr(lo), 768, ptr(stats), 1e-5

            def two():
                check(lib.bg_gemm_ex_fwd(d1, stream()), "ffn1")
                check(lib.bg_gemm_ex_fwd(d2, stream()), "ffn2")

            for zero in (False, True):
                if zero:
                    for t in (hi, lo, w1, w2, w1f, w2f, h):
                        t.zero_()
                tf, tt = timed(fused), timed(two)
                fl = 2.0 * M * 768 * 1024 * 2
                print(f"M={M:7d} {str(dt)[6:]:9s} {'zeros ' if zero else 'random'}  fused {tf:7.1f} us = {fl / tf / 1e6:6.0f} TF   "
                      f"two launches {tt:7.1f} us = {fl / tt / 1e6:6.0f} TF", flush=True)


def stamps(M=30720):
    """s_memtime of workgroup 0 at the phase boundaries of its panels (100 MHz ticks -> us)."""
    lib = _lib.load()
    hi, lo, stats, w1, b1, colsum1, w2, b2 = _ffn_operands(M, torch.bfloat16, 1)
    w1f, w2f = ffn_fragment_order(w1, 4), ffn_fragment_order(w2, 3)
    buf = torch.zeros(32, dtype=torch.int64, device="cuda")
    addr = buf.data_ptr()
    run = lambda: check(lib.bg_ffn_fused_fwd(ptr(hi), ptr(lo), ptr(stats), ptr(w1f), ptr(b1), ptr(colsum1), ptr(w2f), ptr(b2), M, M, None,
                                             ops.bg_dtype(torch.bfloat16), 1e-5, stream()), "ffn")
    for _ in range(3):
        run()
    lib.bg_tune_set(17, addr & 0xffffffff if (addr & 0xffffffff) < 2 ** 31 else (addr & 0xffffffff) - 2 ** 32)
    lib.bg_tune_set(18, addr >> 32)
    run()
    torch.cuda.synchronize()
    lib.bg_tune_set(17, 0)
    lib.bg_tune_set(18, 0)
    t = buf.tolist()
    names = ["x panel -> LDS + barrier", "phase 1", "barrier wait", "epilogue 1 + barrier", "phase 2", "epilogue 2", "barrier wait + merge", "(loop)"]
    n = 8
    for k in range(0, 16, n):
        seg = t[k:k + n + 1]
        print("panel", k // n, "cycles:", " | ".join(f"{names[i]} {seg[i + 1] - seg[i]}" for i in range(n) if seg[i + 1] and seg[i]), "| total", seg[n - 1] - seg[0])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stamps":
        stamps()
        sys.exit(0)
    main()
