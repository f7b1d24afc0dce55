#!/usr/bin/env python
"""Attention-only workload for rocprofv3 passes: the long-sequence kernel on cfg3's shape (256 x 1800, bf16, dense)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hip_ops as ops
g = torch.Generator().manual_seed(0)
B, N = 256, 1800
qkv = (torch.randn(B * N, 2304, generator=g) * 0.7).to(torch.bfloat16).cuda()
for _ in range(4):
    ops.attention(qkv, None, B, N)
torch.cuda.synchronize()
