#!/usr/bin/env python
"""VAE decode of BASELINE configs[2]'s output (15 360 faces + 460 800 edges, bf16): convolutions as implicit GEMMs vs the
materialised im2col path, and the whole pass as one bg_vae_run call vs step by step from Python; interleaved.  Random-init weights (the kernels do not care)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import brepgen_amd as bga
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from vae_stepwise import stepwise  # noqa: E402
from brepgen_amd.pipeline import EDGE_VAE_CFG, SURF_VAE_CFG

torch.manual_seed(0)
surf = bga.AutoencoderKLFastDecode(**SURF_VAE_CFG).cuda().eval()
edge = bga.AutoencoderKL1DFastDecode(**EDGE_VAE_CFG).cuda().eval()
surf.compute_dtype = edge.compute_dtype = torch.bfloat16
F_, G_ = (15360, 460800) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
ONLY = sys.argv[3].split(",") if len(sys.argv) > 3 else None      # e.g. "one_call_program" for a rocprofv3 run
if len(sys.argv) > 4 and sys.argv[4] == "serial":                  # one stream, no concurrent halves: kernel durations a profiler can add up
    surf.two_streams = edge.two_streams = False
zs = torch.randn(F_, 48, device="cuda")
ze = torch.randn(G_, 12, device="cuda")
FLOP = F_ * 9.69e9 + G_ * 0.416e9
out = {}
with torch.no_grad():
    for rnd in range(2):
        for name, flag, ex in (("one_call_program", True, True), ("implicit_gemm", True, False), ("im2col", False, False)):
            if ONLY and name not in ONLY:
                continue
            # (product path = one bg_vae_run program per pass; the two step-by-step baselines live in tests/vae_stepwise.py)
            dec_s = (lambda z: surf.decode_tokens(z)) if ex else (lambda z: stepwise(surf, z.reshape(-1, 4, 4, 3).permute(0, 3, 1, 2), implicit_gemm=flag))
            dec_e = (lambda z: edge.decode_tokens(z)) if ex else (lambda z: stepwise(edge, z.reshape(-1, 4, 3).permute(0, 2, 1), implicit_gemm=flag))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a = dec_s(zs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            b = dec_e(ze)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out.setdefault(name, []).append({"surf_s": round(t1 - t0, 3), "edge_s": round(t2 - t1, 3), "total_s": round(t2 - t0, 3),
                                             "tflops": round(FLOP / (t2 - t0) / 1e12, 1)})
            del a, b
print(json.dumps({"faces": F_, "edges": G_, "algorithmic_tflop": round(FLOP / 1e12, 1), **out}, indent=1))
