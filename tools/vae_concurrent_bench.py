#!/usr/bin/env python
"""VAE decode of BASELINE configs[2]'s output (15 360 faces + 460 800 edges, bf16): surface then edge pass on one stream against the two
passes on forked streams (sampling.decode_latents(concurrent=...)); interleaved rounds, one process."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import brepgen_amd as bga
from brepgen_amd.pipeline import EDGE_VAE_CFG, SURF_VAE_CFG
from brepgen_amd.sampling import decode_latents

torch.manual_seed(0)
surf = bga.AutoencoderKLFastDecode(**SURF_VAE_CFG).cuda().eval()
edge = bga.AutoencoderKL1DFastDecode(**EDGE_VAE_CFG).cuda().eval()
surf.compute_dtype = edge.compute_dtype = torch.bfloat16
B, S, E = (256, 60, 30) if len(sys.argv) < 4 else (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
lat = {"surfZ": torch.randn(B, S, 48, device="cuda"), "edgeZV": torch.randn(B, S, E, 18, device="cuda")}
res = {"sequential": [], "concurrent": [], "concurrent + two halves per pass": []}
with torch.no_grad():
    for rnd in range(4):
        for name, flag, halves in (("sequential", False, False), ("concurrent", True, False), ("concurrent + two halves per pass", True, True)):
            surf.two_streams = edge.two_streams = halves
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = decode_latents(surf, edge, lat, concurrent=flag)
            torch.cuda.synchronize()
            res[name].append(round(time.perf_counter() - t0, 4))
            del out
print(json.dumps({"faces": B * S, "edges": B * S * E, **res}))
