#!/usr/bin/env python
"""s_memtime stamps of one workgroup of the fused QKV + attention kernel (experiments): cycles per section and tile."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import hip_ops as ops
from brepgen_amd import _lib
from brepgen_amd._lib import check, ptr, stream

sys.argv = [sys.argv[0]]
import importlib.util
spec = importlib.util.spec_from_file_location("qc", os.path.join(ROOT, "tools", "qkv_attn_check.py"))

lib = _lib.load()
B, N, dt = 512, 60, torch.bfloat16
g = torch.Generator().manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g)
M = B * N
x = rn(M, 768) * 2
a = x.to(dt).cuda()
grp = x.reshape(M, 12, 64)
stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
w = (rn(2304, 768) * 0.04).to(dt).cuda()
b = rn(2304).cuda()
cs = w.float().sum(1).contiguous()
out = torch.empty(M, 768, device="cuda", dtype=dt)
names = ["tile top", "K loop", "-", "fold + images", "barrier", "attention + stores", "barrier"]
for rep in range(3):
    buf = torch.zeros(16 * 2 * 8, dtype=torch.int64, device="cuda")
    check(lib.bg_qkv_attn_fwd(ptr(a), ptr(w), ptr(b), ptr(cs), ptr(stats), ptr(out), ptr(buf), B, N, 1, -1e-5, stream()), "stamps")
    torch.cuda.synchronize()
    t = buf.cpu().reshape(16, 2, 8)
    if rep < 2:
        continue
    for grp_i, gname in enumerate(("waves 0-3", "waves 4-7")):
        print(gname)
        for tile in range(6):
            s = t[tile, grp_i]
            d = [int(s[k + 1] - s[k]) for k in range(7)]
            nxt = int(t[tile + 1, grp_i, 0] - s[7]) if tile < 5 else 0
            print(f"  tile {tile}: " + "  ".join(f"{n} {v}" for n, v in zip(names, d)) + f"  | total {int(s[7] - s[0]) if tile < 5 else int(s[6] - s[0])}")
