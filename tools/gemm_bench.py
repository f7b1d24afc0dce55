#!/usr/bin/env python
"""A/B the bf16 GEMM variants (bg_tune_set key 0) on the layer shapes of the headline config, interleaved rounds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from brepgen_amd import _lib
import hip_ops as ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4").split(",")]
M = int(os.environ.get("M", 30720))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / iters


def main():
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    shapes = [("qkv", 2304, 768, "bf16"), ("outproj", 768, 768, "resid"), ("ffn1", 1024, 768, "bf16"),
              ("ffn2", 768, 1024, "resid")]
    if os.environ.get("KBIG"):
        kb = int(os.environ["KBIG"])
        shapes = [("bigk_2304", 2304, kb, "bf16"), ("bigk_768", 768, kb, "bf16")]
    data = {}
    for name, N, K, mode in shapes:
        a = torch.randn(M, K, generator=g).cuda().to(BF16)
        w = (torch.randn(N, K, generator=g) / 28).cuda().to(BF16)
        b = torch.randn(N, generator=g).cuda()
        out = torch.zeros(M, N, device="cuda", dtype=BF16 if mode == "bf16" else F32)
        data[name] = (a, w, b, out, mode, N, K)
    # correctness of each variant against variant 0 first
    ref = {}
    for v in VARIANTS:
        lib.bg_tune_set(0, v)
        for name, (a, w, b, out, mode, N, K) in data.items():
            o = torch.zeros_like(out)
            ops.linear(a, w, b, out=o, act=1 if name == "ffn1" else 0)
            torch.cuda.synchronize()
            if 10 < v < 20 and v not in (14, 15):
                continue
            if v == VARIANTS[0]:
                ref[name] = o.float().clone()
            else:
                d = float((o.float() - ref[name]).abs().max())
                print(f"[check] variant {v} {name}: max|diff vs variant {VARIANTS[0]}| = {d:.3e}", flush=True)
    if os.environ.get("NG"):
        for name, (a, w, b, out, mode, N, K) in data.items():
            line = f"[n-groups] {name:8s}:"
            lib.bg_tune_set(0, 0)
            for ng in [int(x) for x in os.environ["NG"].split(",")]:
                lib.bg_tune_set(4, ng)
                if mode == "bf16":
                    us = timed(lambda: ops.linear(a, w, b, out=out, act=1 if name == "ffn1" else 0))
                else:
                    us = timed(lambda: ops.linear(a, w, b, add=out, out=out))
                line += f"  ng{ng} {us:6.1f}us"
            print(line, flush=True)
        lib.bg_tune_set(4, 0)
    if os.environ.get("DEPHASE"):
        for name, (a, w, b, out, mode, N, K) in data.items():
            line = f"[dephase] {name:8s}:"
            lib.bg_tune_set(0, 30)
            for d in [int(x) for x in os.environ["DEPHASE"].split(",")]:
                lib.bg_tune_set(3, d)
                if mode == "bf16":
                    us = timed(lambda: ops.linear(a, w, b, out=out, act=1 if name == "ffn1" else 0))
                else:
                    us = timed(lambda: ops.linear(a, w, b, add=out, out=out))
                line += f"  d{d} {us:6.1f}us"
            print(line, flush=True)
        lib.bg_tune_set(3, 0)
    for rnd in range(3):
        for name, (a, w, b, out, mode, N, K) in data.items():
            line = f"[round {rnd}] {name:8s} {M}x{N}x{K} {mode:5s}:"
            for v in VARIANTS:
                lib.bg_tune_set(0, v)
                if mode == "bf16":
                    us = timed(lambda: ops.linear(a, w, b, out=out, act=1 if name == "ffn1" else 0))
                else:
                    us = timed(lambda: ops.linear(a, w, b, add=out, out=out))
                line += f"  v{v} {us:6.1f}us {2.0 * M * N * K / us / 1e6:6.0f}TF"
            print(line, flush=True)
    if os.environ.get("NO_NET"):
        lib.bg_tune_set(0, 0)
        return
    # whole net (random-init module: timing only; bench.py is the measured headline)
    import brepgen_amd as bga
    torch.manual_seed(0)
    m = bga.SurfZNet(False).cuda().eval()
    m.compute_dtype = BF16
    g = torch.Generator().manual_seed(1234)
    nvalid = torch.randint(8, 61, (512,), generator=g)
    args = [torch.randn(512, 60, 48, generator=g).cuda(), torch.tensor([249]).cuda(),
            torch.randn(512, 60, 6, generator=g).clamp(-3, 3).cuda(), (torch.arange(60)[None] >= nvalid[:, None]).cuda(), None]
    with torch.no_grad():
        for v in VARIANTS:
            lib.bg_tune_set(0, v)
            us = timed(lambda: m(*args), iters=10)
            print(f"[net] SurfZNet B=512 variant {v}: {us:.0f} us/eval", flush=True)
    lib.bg_tune_set(0, 0)


if __name__ == "__main__":
    main()
