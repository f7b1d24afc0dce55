#!/usr/bin/env python
"""Per-phase cycle accounting of the persistent GEMM (variant 31: s_memtime stamps inside the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from brepgen_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops
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
if len(sys.argv) > 2:
    lib.bg_tune_set(8, int(sys.argv[2]))          # start offset of the second workgroup of every CU, units of 64 cycles
    print(f"phase offset knob (bg_tune 8) = {sys.argv[2]}")
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
    ran = d[:, :, 5] > 0
    span = (d[:, :, 5] + d[:, :, 3])[ran].max() - d[:, :, 5][ran].min()
    # phase relation of the two workgroups that share a CU (blocks b and b + G/2: tools/cu_census.hip): the offset between the
    # starts of their first epilogues, as a fraction of a tile's duration -- 0 = in lock-step (the epilogue traffic of one
    # never runs under the K loop of the other), 0.5 = perfectly interleaved
    G = int((d[:, 0, 4] > 0).sum())
    if G >= 2 and G % 2 == 0:
        e0 = d[:G, 0, 6]
        tile = (tot[:G, 0] / d[:G, 0, 4].clamp(min=1)).mean()
        off = (e0[: G // 2] - e0[G // 2:]).abs() / tile
        print(f"  co-resident pairs: |first-epilogue offset| / tile time: median {off.median():.3f}, mean {off.mean():.3f}, "
              f"90th pct {off.kthvalue(max(1, int(0.9 * off.numel()))).values:.3f}  (tile = {tile:.0f} ticks)")
    print(f"{name}: kernel {us:.1f} us; per-wave cycles avg: total {tot.mean():.0f} (max {tot.max():.0f}) "
          f"wait+barrier {d[:,:,0].mean():.0f} compute {d[:,:,1].mean():.0f} epilogue {d[:,:,2].mean():.0f} "
          f"tiles/wg {d[:,:,4][ran].mean():.2f}; span {span:.0f} ticks -> {span/us:.0f} memtime ticks/us", flush=True)
lib.bg_tune_set(0, 0)
lib.bg_tune_set(8, 0)
