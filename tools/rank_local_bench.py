#!/usr/bin/env python
"""BASELINE configs[3] / configs[4] as ONE rank executes them: the four loops of the cascade (sample.py:126-286) at the rank-local
batch -- ABC: 512 samples x (2 x 50 faces, 40 edges), bf16, no guidance (eval_config.yaml:12-13; 4096 samples over 8 GPUs);
furniture: 256 samples x (60 faces, 40 edges), fp16, classifier-free guidance (conditional + unconditional rows in one eps-eval,
eval_config.yaml:44-47; 1024 samples over 4 GPUs).  Random-init weights (no checkpoints offline): the bbox de-duplication then
keeps nearly every face and edge -- the cascade's WORST case for variable-length execution.

    python tools/rank_local_bench.py cfg4|cfg5 [K]      K: iterations per loop (default: the full 158 + 250 / 209 / 158 + 250 / 209)

With K the per-iteration time of every loop is measured on K iterations -- at the token counts such a short run leaves (more than
the full loops end with: an upper bound) -- and printed next to the one full run per round kept under profiles/."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import brepgen_amd as bga
from brepgen_amd.pipeline import SCHED_KW
from brepgen_amd.sampling import CascadeSampler

CFG = {
    # name: (workload, batch, faces before the late doubling, edges, autocast dtype, classifier-free guidance)
    "cfg4": ("BASELINE configs[3], one rank's share: ABC cascade, 512 samples x (2 x 50 faces, 40 edges), bf16", 512, 50, 40, torch.bfloat16, False),
    "cfg5": ("BASELINE configs[4], one rank's share: furniture cascade, 256 samples x (60 faces, 40 edges), fp16, classifier-free "
             "guidance (2 x 256 rows per eps-eval)", 256, 60, 40, torch.float16, True),
}
FULL = {"surfPos": 158 + 250, "surfZ": 209, "edgePos": 158 + 250, "edgeZV": 209}


def run(name, k=None):
    what, B, S, E, dt, cf = CFG[name]
    torch.manual_seed(0)
    dev = torch.device("cuda")
    nets = [cls(cf).to(dev).eval() for cls in (bga.SurfPosNet, bga.SurfZNet, bga.EdgePosNet, bga.EdgeZNet)]
    sampler = CascadeSampler(*nets, bga.PNDMScheduler(**SCHED_KW), bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **SCHED_KW),
                             use_cf=cf, class_id=6, guidance=0.6, bbox_threshold=0.08, autocast=dt)
    sampler.sample(2, S, E, generator=torch.Generator().manual_seed(0), pndm_pos_steps=2, ddpm_pos_steps=2, pndm_z_steps=2)   # warm-up
    kw = {} if k is None else dict(pndm_pos_steps=k, ddpm_pos_steps=k, pndm_z_steps=k)
    its = dict(FULL) if k is None else {"surfPos": 2 * k, "surfZ": k, "edgePos": 2 * k, "edgeZV": k}
    stages = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat = sampler.sample(B, S, E, generator=torch.Generator().manual_seed(1), timings=stages, **kw)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    per_it = {s: stages[s] / its[s] for s in its}
    out = {"workload": what + ", random-init weights, variable-length execution", "iterations_timed": its,
           "stage_s": {s: round(v, 3) for s, v in stages.items()}, "ms_per_iteration": {s: round(1e3 * v, 2) for s, v in per_it.items()},
           "loops_s": round(total, 3),
           "valid_faces_mean": round(float((~lat["surfMask"]).sum(1).float().mean()), 2),
           "valid_edges_mean_per_sample": round(float((~lat["edgeM"]).sum((1, 2)).float().mean()), 1),
           "finite": all(bool(torch.isfinite(v).all()) for v in lat.values() if v.is_floating_point())}
    if k is None:
        # executed FLOPs of the full loops on the masks this run produced (F(n) per sample and evaluation; with
        # guidance every evaluation runs conditional + unconditional rows): the unit's own roofline figure
        F = lambda n, c: n * (12 * 7_864_320 + c) + 36_864 * n * n
        rows = 2 if cf else 1
        nf = (~lat["surfMask"]).sum(1).double().cpu()
        ne = (~lat["edgeM"]).sum((1, 2)).double().cpu()
        S2 = S if cf else 2 * S                                       # (the late doubling of sample.py:140-142 happens without guidance only)
        fl = {"surfPos": rows * B * (158 * F(S, 2.38e6) + 250 * F(S2, 2.38e6)),
              "surfZ": rows * 209 * (float(sum(F(float(n), 3.70e6 - 2.44e6) for n in nf)) + B * S2 * 2.44e6),
              "edgePos": rows * 408 * (float(sum(F(float(n) * E, 2.38e6) for n in nf)) + B * S2 * 2.44e6),
              "edgeZV": rows * 209 * (float(sum(F(float(n), 4.78e6) for n in ne)) + B * S2 * 2.44e6)}
        out["roofline"] = {s: {"executed_tflop": round(v / 1e12, 1), "executed_tflops": round(v / 1e12 / stages[s], 1),
                               "frac_of_mfma_peak": round(v / 1e12 / stages[s] / 2500.0, 4)} for s, v in fl.items()}
        out["roofline"]["loops"] = {"executed_tflop": round(sum(fl.values()) / 1e12, 1), "executed_tflops": round(sum(fl.values()) / 1e12 / total, 1),
                                    "frac_of_mfma_peak": round(sum(fl.values()) / 1e12 / total / 2500.0, 4), "bound": "mfma",
                                    "peak": "2500 TFLOP/s dense " + ("fp16" if dt == torch.float16 else "bf16")}
    if k is not None:
        # A K-iteration run does NOT reproduce the full loops' token counts: after a few iterations the de-duplication between the
        # stages removes only the exact copies of the late doubling, and every edge slot stays valid, whereas the full loops (even
        # with random-init weights) end with about half of that (ABC: 1186 of 4000 edge tokens per sample).  The per-iteration
        # times below are therefore an UPPER bound on the full loops' -- they are the cost at the token counts printed beside them;
        # the full loops, run once per round, are in profiles/ (full_run).
        out["note"] = ("K-iteration run: per-iteration times at the validity such a short run leaves (valid_* above), an upper bound "
                       "on the full loops' per-iteration times")
    else:
        out["samples_per_s_per_rank"] = round(B / total, 2)
    del nets, sampler, lat
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    K = int(sys.argv[2]) if len(sys.argv) > 2 else None
    print(json.dumps(run(sys.argv[1], K), indent=1))
