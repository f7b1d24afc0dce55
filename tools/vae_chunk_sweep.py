#!/usr/bin/env python
"""VAE decode of BASELINE configs[2]'s output (15 360 faces + 460 800 edges, bf16) against the workspace budget that sizes bg_vae_run's
chunks of samples (_HipVAE.WS_BUDGET): does a chunk whose activations stay in the 256 MB Infinity Cache beat few large chunks?"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import brepgen_amd as bga
from brepgen_amd.pipeline import EDGE_VAE_CFG, SURF_VAE_CFG

torch.manual_seed(0)
surf = bga.AutoencoderKLFastDecode(**SURF_VAE_CFG).cuda().eval()
edge = bga.AutoencoderKL1DFastDecode(**EDGE_VAE_CFG).cuda().eval()
surf.compute_dtype = edge.compute_dtype = torch.bfloat16
F_, G_ = (15360, 460800) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
zs = torch.randn(F_, 48, device="cuda")
ze = torch.randn(G_, 12, device="cuda")
budgets = [int(b) << 20 for b in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8 << 30, 2 << 30, 1 << 30, 512 << 20, 256 << 20, 128 << 20, 64 << 20]
rows = []
with torch.no_grad():
    for halves in (True, False):
        for bud in budgets:
            for m in (surf, edge):
                m.WS_BUDGET, m.two_streams = bud, halves
                m.release_workspace()
            t = {"surf": [], "edge": []}
            for rnd in range(3):
                for name, mod, z in (("surf", surf, zs), ("edge", edge, ze)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = mod.decode_tokens(z)
                    torch.cuda.synchronize()
                    t[name].append(time.perf_counter() - t0)
                    del out
            row = {"two_halves": halves, "ws_budget_MiB": bud >> 20, "surf_s": round(min(t["surf"]), 4), "edge_s": round(min(t["edge"]), 4),
                   "sum_s": round(min(t["surf"]) + min(t["edge"]), 4)}
            rows.append(row)
            print(json.dumps(row), flush=True)
