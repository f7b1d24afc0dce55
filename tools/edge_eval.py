#!/usr/bin/env python
"""One edge-net eps-evaluation workload for rocprofv3: EdgeZNet at cfg3's shape (256 x 60 x 30, bf16), variable-length
(default) or dense (argv[1] == dense).  Masks as in bench.py's edge-net extra."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import brepgen_amd as bga
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
dev = "cuda"
B, S, E = 256, 60, 30
g = torch.Generator().manual_seed(99)
torch.manual_seed(1)
net = bga.EdgeZNet(False).to(dev).eval()
net.compute_dtype = torch.bfloat16
net.varlen = not dense
nf = torch.randint(8, S + 1, (B,), generator=g)
smask = torch.arange(S)[None] >= nf[:, None]
pos = torch.randn(B, S, 6, generator=g).clamp(-3, 3).to(dev)
sz = torch.randn(B, S, 48, generator=g).to(dev)
ne = torch.randint(3, E + 1, (B, S), generator=g)
emask = (torch.arange(E)[None, None] >= ne[:, :, None]) | smask[:, :, None]
args = (torch.randn(B, S, E, 18, generator=g).to(dev), torch.tensor([249], device=dev),
        torch.randn(B, S, E, 6, generator=g).clamp(-3, 3).to(dev), pos, sz, emask.to(dev), None)
with torch.no_grad():
    for _ in range(3):
        net(*args)
torch.cuda.synchronize()
print("valid tokens", int((~emask).sum()), "of", B * S * E)
