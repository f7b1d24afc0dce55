#!/usr/bin/env python
"""One-shot GPU diagnostic: every parity measurement + per-kernel timings, nothing asserted.

    gpurun -- 'python tests/gpu_check.py > gpurun_out/check.log 2>&1'

Each section is wrapped so one failing kernel does not hide the others.
"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import parity_cases as pc  # noqa: E402
import hip_ops as ops  # noqa: E402

F32, BF16 = torch.float32, torch.bfloat16
QUICK = "--quick" in sys.argv


def run(name, fn, *a, **k):
    try:
        t0 = time.time()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        r = {kk: (round(v, 9) if isinstance(v, float) else v) for kk, v in r.items()}
        print(f"[case] {name}: {json.dumps(r)}  ({time.time() - t0:.2f}s)", flush=True)
    except Exception:
        print(f"[FAIL] {name}:\n{traceback.format_exc()}", flush=True)
        try:
            torch.cuda.synchronize()
        except Exception:
            print("[FATAL] device unusable after failure", flush=True)
            raise


def timeit(name, fn, flops=None, bytes_=None, iters=20):
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        extra = ""
        if flops:
            extra += f"  {flops / us / 1e6:.1f} TFLOP/s"
        if bytes_:
            extra += f"  {bytes_ / us / 1e3:.1f} GB/s"
        print(f"[time] {name}: {us:.1f} us{extra}", flush=True)
    except Exception:
        print(f"[FAIL] time {name}:\n{traceback.format_exc()}", flush=True)


def main():
    print("device:", torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    # ---- elementwise
    run("sincos", pc.sincos_case)
    for od in (F32, BF16):
        for silu in (False, True):
            run(f"layernorm M=61 {od} silu={silu}", pc.layernorm_case, 61, od, silu)
    for t in (249, 1, 0):
        run(f"ddpm t={t}", pc.ddpm_case, t)
    run("ddpm t=100 cfg", pc.ddpm_case, 100, guidance=0.6)
    run("ddpm odd size", pc.ddpm_case, 7, shape=(3, 7, 6))
    run("pndm 209 steps", pc.pndm_case)
    run("pndm 30 steps cfg", pc.pndm_case, n_steps=30, guidance=0.6)
    # ---- fp32 GEMM
    for (M, N, K) in [(1, 768, 768), (60, 768, 6), (130, 768, 48), (61, 768, 12), (77, 6, 768), (64, 2304, 768)]:
        run(f"gemm f32 {M}x{N}x{K}", pc.gemm_case, M, N, K, F32)
    run("gemm f32 relu+resid", pc.gemm_case, 100, 1024, 768, F32, act=1, add_mode="resid")
    run("gemm f32 bcast add", pc.gemm_case, 100, 768, 48, F32, add_mode=30)
    # ---- bf16 GEMM
    for (M, N, K) in [(60, 768, 768), (128, 2304, 768), (257, 1024, 768), (1000, 768, 1024), (1, 768, 768)]:
        run(f"gemm bf16 {M}x{N}x{K}", pc.gemm_case, M, N, K, BF16)
    run("gemm bf16 relu bf16out", pc.gemm_case, 300, 1024, 768, BF16, act=1, out_dtype=BF16)
    run("gemm bf16 resid", pc.gemm_case, 300, 768, 1024, BF16, add_mode="resid")
    run("gemm bf16 bcast", pc.gemm_case, 300, 768, 768, BF16, add_mode=60)
    run("gemm bf16 N=64 pad (6 valid)", pc.gemm_case, 300, 64, 768, BF16, n_valid=6)
    run("gemm bf16 N=64 pad (48 valid)", pc.gemm_case, 130, 64, 768, BF16, n_valid=48)
    run("gemm bf16 nobias", pc.gemm_case, 130, 768, 768, BF16, bias=False)
    # ---- attention
    for N in (17, 30, 60, 64, 100, 130, 257):
        run(f"attn bf16 N={N} ragged", pc.attn_case, 3, N, BF16, "ragged")
    run("attn bf16 N=60 nomask", pc.attn_case, 2, 60, BF16, None)
    run("attn bf16 N=300 random", pc.attn_case, 2, 300, BF16, "random")
    run("attn bf16 N=1800 random", pc.attn_case, 1, 1800, BF16, "random")
    for N in (17, 60, 130):
        run(f"attn f32 N={N} ragged", pc.attn_case, 2, N, F32, "ragged")
    run("attn f32 N=300 random", pc.attn_case, 2, 300, F32, "random")
    # ---- whole nets vs golden (reference classes) and vs oracle
    for name in sorted(pc.MANIFEST["cases"]):
        run(f"golden {name} f32", pc.golden_case, name, F32)
        run(f"golden {name} bf16", pc.golden_case, name, BF16)
    if not QUICK:
        run("oracle SurfZNet B=4 N=60 f32", pc.oracle_case, "SurfZNet", 4, 60, 1, F32)
        run("oracle SurfZNet B=4 N=60 bf16", pc.oracle_case, "SurfZNet", 4, 60, 1, BF16)
        run("oracle EdgeZNet B=1 S=10 E=20 bf16", pc.oracle_case, "EdgeZNet", 1, 10, 20, BF16)
        run("oracle EdgePosNet cf B=2 S=8 E=20 bf16", pc.oracle_case, "EdgePosNet", 2, 8, 20, BF16, True)
        run("chain cfg1 f32 (10 steps)", pc.ddpm_chain_case, F32, 10)
        run("chain cfg1 bf16 (10 steps)", pc.ddpm_chain_case, BF16, 10)

    # ---- timings at the headline shape (B=512, N=60 -> M=30720)
    M = 30720
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, 768, generator=g).cuda()
    w = torch.ones(768).cuda()
    b = torch.zeros(768).cuda()
    timeit("layernorm bf16 M=30720", lambda: ops.layernorm(x, w, b, out_dtype=BF16), bytes_=M * 768 * 6)
    a16 = torch.randn(M, 768, generator=g).cuda().to(BF16)
    a16b = torch.randn(M, 1024, generator=g).cuda().to(BF16)
    for (N, K, a) in [(2304, 768, a16), (768, 768, a16), (1024, 768, a16), (768, 1024, a16b)]:
        wt = (torch.randn(N, K, generator=g) / 28).cuda().to(BF16)
        bias = torch.zeros(N).cuda()
        out = torch.empty(M, N, device="cuda", dtype=BF16)
        timeit(f"gemm bf16 {M}x{N}x{K} (bf16 out)", lambda: ops.linear(a, wt, bias, out=out), flops=2.0 * M * N * K)
        outf = torch.zeros(M, N, device="cuda", dtype=F32)
        timeit(f"gemm bf16 {M}x{N}x{K} (f32 resid)", lambda: ops.linear(a, wt, bias, add=outf, out=outf),
               flops=2.0 * M * N * K)
        timeit(f"torch bf16 matmul {M}x{N}x{K}", lambda: torch.matmul(a, wt.t()), flops=2.0 * M * N * K)
    qkv = torch.randn(M, 2304, generator=g).cuda().to(BF16)
    mask = torch.zeros(512, 60, dtype=torch.bool).cuda()
    timeit("attn bf16 B=512 N=60", lambda: ops.attention(qkv, mask, 512, 60), flops=4.0 * 512 * 12 * 60 * 60 * 64)
    qkv2 = torch.randn(8 * 1800, 2304, generator=g).cuda().to(BF16)
    timeit("attn bf16 B=8 N=1800", lambda: ops.attention(qkv2, None, 8, 1800), flops=4.0 * 8 * 12 * 1800 * 1800 * 64,
           iters=5)
    af = torch.randn(M, 48, generator=g).cuda()
    wf = torch.randn(768, 48, generator=g).cuda()
    timeit("gemm f32 30720x768x48", lambda: ops.linear(af, wf, None), flops=2.0 * M * 768 * 48)

    # ---- whole-net step time
    for net, B, S, E in [("SurfZNet", 512, 60, 1), ("SurfPosNet", 512, 60, 1), ("EdgeZNet", 8, 60, 30)]:
        try:
            m, _ = pc.build_net(net, 1, False, BF16)
            args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs(net, B, S, E, False)]
            with torch.no_grad():
                timeit(f"{net} fwd bf16 B={B} S={S} E={E}", lambda: m(*args), iters=5)
        except Exception:
            print(f"[FAIL] net timing {net}:\n{traceback.format_exc()}", flush=True)


if __name__ == "__main__":
    main()
