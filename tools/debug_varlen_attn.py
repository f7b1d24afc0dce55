#!/usr/bin/env python
"""Per-sample error of bg_attn_varlen_fwd (short kernel) for chosen lengths -- debugging aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from brepgen_amd import _lib
import parity_cases as pc
lib = _lib.load()
for nv in ([60, 1, 2, 31, 32, 33, 59, 17], [1], [60], [33], [32, 32], [5, 60]):
    for N in (60, 64, 40):
        nvalid = torch.tensor([min(n, N) for n in nv])
        B = len(nv)
        offs = torch.zeros(B + 1, dtype=torch.int32); offs[1:] = torch.cumsum(nvalid, 0)
        M = int(offs[-1])
        g = torch.Generator().manual_seed(1)
        qkv = torch.randn(M, 2304, generator=g); qkv[:, :768] *= 0.25
        qd = qkv.to(torch.bfloat16)
        qdev, odev = qd.cuda(), offs.cuda()
        out = torch.zeros(M, 768, dtype=torch.bfloat16, device="cuda")
        _lib.check(lib.bg_attn_varlen_fwd(qdev.data_ptr(), None, out.data_ptr(), B, N, _lib.BG_BF16, odev.data_ptr(), _lib.stream()), "x")
        torch.cuda.synchronize()
        errs = []
        for b in range(B):
            lo, hi = int(offs[b]), int(offs[b + 1])
            want = pc._attn_ref(qd[lo:hi], None, 1, hi - lo)
            errs.append(round(float((out[lo:hi].float().cpu().double() - want).abs().max()), 4))
        print(f"N={N} nvalid={nvalid.tolist()} errs={errs}", flush=True)
