#!/usr/bin/env python
"""BASELINE configs[2]: the whole DeepCAD cascade (sample.py:120-294: surface positions -> surface latents -> edge
positions -> edge latents + vertices, then the VAE decode of every face and edge), batch = 256, bf16, one MI355X.
Random-init weights (there are no checkpoints offline): the bbox de-duplication then keeps nearly every face and edge,
so variable-length execution has little to drop -- this is the cascade's WORST case; with trained weights about half
of the 2 x 30 faces are duplicates and most edge slots are padding.  Stage times from `sample(timings=...)`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import brepgen_amd as bga
from brepgen_amd.pipeline import EDGE_VAE_CFG, SCHED_KW, SURF_VAE_CFG
from brepgen_amd.sampling import CascadeSampler, decode_latents

S, E = 30, 30                                                        # eval_config.yaml: deepcad


def run(B=256, datalike=False, graphs=False):
    torch.manual_seed(0)
    dev = torch.device("cuda")
    nets = [cls(False).to(dev).eval() for cls in (bga.SurfPosNet, bga.SurfZNet, bga.EdgePosNet, bga.EdgeZNet)]
    if os.environ.get("BG_SURFZ_VARLEN") in ("0", "1"):             # experiments: force SurfZNet's variable-length execution on / off
        nets[1].varlen = os.environ["BG_SURFZ_VARLEN"] == "1"
    for kv in filter(None, os.environ.get("BG_TUNE", "").split(",")):   # experiments: bg_tune_set key=value[,key=value]
        from brepgen_amd import _lib
        _lib.load().bg_tune_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
    surf_vae = bga.AutoencoderKLFastDecode(**SURF_VAE_CFG).to(dev).eval()
    edge_vae = bga.AutoencoderKL1DFastDecode(**EDGE_VAE_CFG).to(dev).eval()
    surf_vae.compute_dtype = edge_vae.compute_dtype = torch.bfloat16
    sampler = CascadeSampler(*nets, bga.PNDMScheduler(**SCHED_KW), bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **SCHED_KW),
                             bbox_threshold=0.08, autocast=True, graphs=graphs)
    import brepgen_amd.sampling as smp
    real_s, real_e = smp.dedup_surfaces, smp.dedup_edges
    if datalike:
        # Emulate the validity statistics of a trained cascade on top of the real de-duplication: keep U{4..30} of the 60
        # face slots and U{2..8} of the 30 edge slots of every kept face (DeepCAD solids have a handful of faces with a few
        # edges each; the late doubling alone makes half of the 60 slots duplicates).  SYNTHETIC masks, for the cost model only.
        g = torch.Generator().manual_seed(5)

        def fake_surfaces(x, thr):
            pos, mask = real_s(x, thr)
            keep = torch.randint(4, 31, (x.shape[0], 1), generator=g).to(x.device)
            mask = mask | (torch.arange(x.shape[1], device=x.device)[None] >= keep)
            return pos.masked_fill(mask[..., None], 0.0), mask

        def fake_edges(ep, surf_mask, thr):
            m = real_e(ep, surf_mask, thr)
            keep = torch.randint(2, 9, (*ep.shape[:2], 1), generator=g).to(ep.device)
            return m | (torch.arange(ep.shape[2], device=ep.device)[None, None] >= keep)

        smp.dedup_surfaces, smp.dedup_edges = fake_surfaces, fake_edges
    try:
        sampler.sample(2, S, E, generator=torch.Generator().manual_seed(0), pndm_pos_steps=3, ddpm_pos_steps=3, pndm_z_steps=3)   # warm-up
        stages = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat = sampler.sample(B, S, E, generator=torch.Generator().manual_seed(1), timings=stages)
        torch.cuda.synchronize()
        t_cas = time.perf_counter() - t0
        t0 = time.perf_counter()
        dec = decode_latents(surf_vae, edge_vae, lat)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t0
    finally:
        smp.dedup_surfaces, smp.dedup_edges = real_s, real_e
    stages = {k: round(v, 3) for k, v in stages.items()}
    finite = all(bool(torch.isfinite(v).all()) for v in dec.values() if v.is_floating_point())
    # executed FLOPs of the four loops on the masks this run produced (F(n) per sample and evaluation; the
    # face-position net runs 158 PNDM steps on S faces and 250 DDPM steps on 2 S, no mask; the edge-position net on every edge slot of
    # a valid face; the per-face conditioning embeds of the edge nets: 2.44 MFLOP per face) -- the unit's own roofline figure
    F = lambda n, c: n * (12 * 7_864_320 + c) + 36_864 * n * n
    nf = (~lat["surfMask"]).sum(1).double().cpu()
    ne = (~lat["edgeM"]).sum((1, 2)).double().cpu()
    fl = {"surfPos": B * (158 * F(S, 2.38e6) + 250 * F(2 * S, 2.38e6)),
          "surfZ": 209 * float(sum(F(float(n), 3.70e6 - 2.44e6) for n in nf)) + 209 * B * 2 * S * 2.44e6,
          "edgePos": 408 * (float(sum(F(float(n) * E, 2.38e6) for n in nf)) + B * 2 * S * 2.44e6),
          "edgeZV": 209 * (float(sum(F(float(n), 4.78e6) for n in ne)) + B * 2 * S * 2.44e6)}
    roof = {k: {"executed_tflop": round(v / 1e12, 1), "executed_tflops": round(v / 1e12 / stages[k], 1),
                "frac_of_mfma_peak": round(v / 1e12 / stages[k] / 2500.0, 4)} for k, v in fl.items() if stages.get(k)}
    roof["cascade"] = {"executed_tflop": round(sum(fl.values()) / 1e12, 1), "executed_tflops": round(sum(fl.values()) / 1e12 / t_cas, 1),
                       "frac_of_mfma_peak": round(sum(fl.values()) / 1e12 / t_cas / 2500.0, 4), "bound": "mfma",
                       "peak": "2500 TFLOP/s dense bf16"}
    return {"workload": f"DeepCAD cascade, batch {B}, {S}x2 faces x {E} edges, bf16, random-init weights"
                        + (", SYNTHETIC data-like validity masks" if datalike else ""),
            "graphs": graphs, "stage_s": stages, "vae_decode_s": round(t_dec, 3), "cascade_s": round(t_cas, 3),
            "total_s": round(t_cas + t_dec, 3), "samples_per_s": round(B / (t_cas + t_dec), 2),
            "valid_faces_mean": round(float((~lat["surfMask"]).sum(1).float().mean()), 2),
            "valid_edges_mean_per_sample": round(float((~lat["edgeM"]).sum((1, 2)).float().mean()), 1),
            "roofline": roof, "finite": finite}


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    DATALIKE = len(sys.argv) > 2 and sys.argv[2] == "datalike"
    GRAPHS = {"graphs": True, "auto": "auto"}.get(sys.argv[3] if len(sys.argv) > 3 else "", False)
    print(json.dumps(run(B, DATALIKE, GRAPHS), indent=1))
