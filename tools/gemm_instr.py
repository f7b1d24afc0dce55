#!/usr/bin/env python
"""Per-phase cycle accounting of the persistent GEMM (variant 31: s_memtime stamps inside the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from brepgen_amd import _lib, ops
BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()
M = 30720
g = torch.Generator().manual_seed(0)
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
def s32(v):
    v &= 0xffffffff
    return v - (1 << 32) if v >= (1 << 31) else v
lib.bg_tune_set(1, s32(p))
lib.bg_tune_set(2, s32(p >> 32))
M = int(sys.argv[1]) if len(sys.argv) > 1 else M
for name, N, K, mode in [("qkv", 2304, 768, "bf16"), ("outproj", 768, 768, "resid"), ("ffn1", 1024, 768, "bf16"), ("ffn2", 768, 1024, "resid"),
                         ("outproj split", 768, 768, "split"), ("ffn2 split", 768, 1024, "split")]:
    a = torch.randn(M, K, generator=g).cuda().to(BF16)
    w = (torch.randn(N, K, generator=g) / 28).cuda().to(BF16)
    b = torch.randn(N, generator=g).cuda()
    out = torch.zeros(M, N, device="cuda", dtype=BF16 if mode == "bf16" else F32)
    hi, lo = torch.randn(M, N, device="cuda").to(BF16), (torch.randn(M, N, device="cuda") * 1e-3).to(BF16)

    def run():
        if mode == "bf16":
            ops.linear(a, w, b, out=out)
        elif mode == "resid":
            ops.linear(a, w, b, add=out, out=out)
        else:                                                         # split residual stream + row statistics (out-proj / FFN2 of the denoisers)
            ops.linear_ex(a, w, b, split_out=True, res=(hi, lo), want_stats=True)
    lib.bg_tune_set(0, 30)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.bg_tune_set(0, 31)
    dbg.zero_()
    run()                                                             # (allocations of linear_ex outside the timed launch)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    d = dbg.cpu().reshape(512, 4, 8).double()
    us = e0.elapsed_time(e1) * 1000
    tot = d[:, :, 3]
    span = (d[:, :, 5] + d[:, :, 3]).max() - d[:, :, 5][d[:, :, 5] > 0].min()
    print(f"{name}: kernel {us:.1f} us; per-wave cycles avg: total {tot.mean():.0f} (max {tot.max():.0f}) "
          f"wait+barrier {d[:,:,0].mean():.0f} compute {d[:,:,1].mean():.0f} epilogue {d[:,:,2].mean():.0f} "
          f"tiles/wg {d[:,:,4].mean():.2f}; span {span:.0f} cyc -> clock {span/us:.0f} MHz(memtime ticks/us)", flush=True)
lib.bg_tune_set(0, 0)
