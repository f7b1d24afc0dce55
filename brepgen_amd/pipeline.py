"""Config-driven driver for the denoising path: the part of the reference's ``sample.py`` that this package owns.

Reads the reference's own ``eval_config.yaml`` (keys ``surfpos_weight … edgevae_weight, batch_size, bbox_threshold,
num_surfaces, num_edges, use_cf, class_label``; sample.py:35-49, 372-381), loads the published ``.pt`` state dicts
into the HIP modules (identical keys; the VAE files hold the full auto-encoder and are loaded ``strict=False`` exactly
as sample.py:83,98 does), runs stages 1-5 of sample.py:120-294 (cascade + VAE decode) sharded over the ranks of
``torch.distributed`` if it is initialised, and writes the tensors the OpenCascade post-process consumes
(sample.py:296-299) as one ``.npz`` per batch.  The B-rep construction itself stays in the reference.

    python -m brepgen_amd.pipeline --mode deepcad [--config eval_config.yaml] [--batches 1] [--seed 0]
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m brepgen_amd.pipeline --mode abc --batch-size 4096
"""
import argparse
import os

import numpy as np
import torch
import yaml

from . import _lib
from .network import EdgePosNet, EdgeZNet, SurfPosNet, SurfZNet
from .sampling import CascadeSampler, decode_latents
from .schedulers import DDPMScheduler, PNDMScheduler
from .vae import AutoencoderKL1DFastDecode, AutoencoderKLFastDecode

# sample.py:21-32
TEXT2INT = {"uncond": 0, "bathtub": 1, "bed": 2, "bench": 3, "bookshelf": 4, "cabinet": 5, "chair": 6, "couch": 7,
            "lamp": 8, "sofa": 9, "table": 10}

SURF_VAE_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                    up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512],
                    layers_per_block=2, act_fn="silu", latent_channels=3, norm_num_groups=32, sample_size=512)   # sample.py:72-82
EDGE_VAE_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownBlock1D"] * 3, up_block_types=["UpBlock1D"] * 3,
                    block_out_channels=[128, 256, 512], layers_per_block=2, act_fn="silu", latent_channels=3,
                    norm_num_groups=32, sample_size=512)                                                         # sample.py:86-97
SCHED_KW = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001,
                beta_end=0.02)                                                                                   # sample.py:101-117
REQUIRED_KEYS = ("surfpos_weight", "surfz_weight", "edgepos_weight", "edgez_weight", "surfvae_weight", "edgevae_weight",
                 "batch_size", "bbox_threshold", "num_surfaces", "num_edges", "use_cf")


def load_eval_args(config_path, mode):
    """eval_config.yaml section -> dict (sample.py:378-381), with the keys this path needs checked."""
    with open(config_path) as f:
        config = yaml.safe_load(f)
    if mode not in config:
        raise KeyError(f"{config_path} has no section {mode!r} (have {sorted(config)})")
    args = dict(config[mode])
    missing = [k for k in REQUIRED_KEYS if k not in args]
    if missing:
        raise KeyError(f"{config_path}[{mode}] lacks {missing}")
    return args


def class_id(eval_args):
    """sample.py:47-49: the text label of the furniture model -> its embedding row (0 = unconditional)."""
    if not eval_args["use_cf"]:
        return 0
    label = eval_args["class_label"]
    if label not in TEXT2INT:
        raise KeyError(f"class_label {label!r} is not one of {sorted(TEXT2INT)}")
    return TEXT2INT[label]


def _load(module, path, strict, device):
    sd = torch.load(path, map_location="cpu")
    result = module.load_state_dict(sd, strict=strict)
    if not strict and result.missing_keys:          # strict=False is for the EXTRA (encoder) keys of the full-VAE files only
        raise RuntimeError(f"{path}: state dict lacks {len(result.missing_keys)} tensors, e.g. {result.missing_keys[:3]}")
    return module.to(device).eval()


def build(eval_args, device="cuda", dist=None, autocast=torch.float16, weight_root="."):
    """-> (CascadeSampler, surf_vae, edge_vae).  autocast: operand dtype inside the sampling loop -- the reference runs
    ``torch.cuda.amp.autocast()`` = fp16 (sample.py:121); torch.bfloat16 and False (exact fp32) are accepted too."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.BrepgenHipError("brepgen_amd.pipeline runs on the MI355X only; there is no CPU fallback")
    p = lambda k: os.path.join(weight_root, eval_args[k])
    cf = bool(eval_args["use_cf"])
    nets = [_load(cls(cf), p(k), True, dev) for cls, k in ((SurfPosNet, "surfpos_weight"), (SurfZNet, "surfz_weight"),
                                                         (EdgePosNet, "edgepos_weight"), (EdgeZNet, "edgez_weight"))]
    surf_vae = _load(AutoencoderKLFastDecode(**SURF_VAE_CFG), p("surfvae_weight"), False, dev)
    edge_vae = _load(AutoencoderKL1DFastDecode(**EDGE_VAE_CFG), p("edgevae_weight"), False, dev)
    sampler = CascadeSampler(*nets, PNDMScheduler(**SCHED_KW), DDPMScheduler(clip_sample=True, clip_sample_range=3, **SCHED_KW),
                             use_cf=cf, class_id=class_id(eval_args), guidance=0.6,
                             bbox_threshold=eval_args["bbox_threshold"], dist=dist, autocast=autocast)
    if autocast:
        surf_vae.compute_dtype = edge_vae.compute_dtype = torch.bfloat16 if autocast is True else autocast
    return sampler, surf_vae, edge_vae


@torch.no_grad()
def sample_batch(sampler, surf_vae, edge_vae, eval_args, generator=None, batch_size=None, **schedule):
    """One pass of sample.py:120-299: latents + decoded point grids for the whole batch on every rank, as numpy arrays
    named like the locals of sample.py (bbox values already divided by 3 as at sample.py:297-299).

    Sharded run: each rank denoises AND VAE-decodes only its own samples (the decode is per face / per edge, SURVEY
    8(e)); the latents and the decoded grids then travel in the ONE all_gather of the path."""
    from .sampling import gather_latents
    B = batch_size or eval_args["batch_size"]
    lat = sampler.sample(B, eval_args["num_surfaces"], eval_args["num_edges"], generator=generator, gather=False,
                         **schedule)
    dec = decode_latents(surf_vae, edge_vae, lat)                    # this rank's rows only
    dec = gather_latents(dec, sampler.dist, batch_size=B)
    host = lambda t: t.detach().float().cpu().numpy()
    return {"surfPos": host(dec["surfPos"]) / 3.0, "surfMask": dec["surfMask"].cpu().numpy(), "surfZ": host(dec["surfZ"]),
            "edge_pos": host(dec["edgePos"]) / 3.0, "edge_mask": dec["edgeM"].cpu().numpy(),
            "edge_z": host(dec["edgeZV"][..., :12]), "edgeV": host(dec["edgeV"]),
            "surf_ncs": host(dec["surf_ncs"]), "edge_ncs": host(dec["edge_ncs"])}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--mode", choices=["abc", "deepcad", "furniture"], default="abc")          # sample.py:374-376
    ap.add_argument("--config", default="eval_config.yaml")
    ap.add_argument("--weights", default=".", help="directory the *_weight paths of the config are relative to")
    ap.add_argument("--batches", type=int, default=1, help="the reference loops forever (sample.py:383); here: N batches")
    ap.add_argument("--batch-size", type=int, default=None, help="override the config (whole job, sharded over ranks)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", choices=["fp16", "bf16", "fp32"], default="fp16")
    a = ap.parse_args(argv)
    eval_args = load_eval_args(a.config, a.mode)
    dist = None
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)                                                        # before the communicator exists
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))              # RCCL over xGMI
    autocast = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": False}[a.dtype]
    sampler, surf_vae, edge_vae = build(eval_args, "cuda", dist, autocast, a.weights)
    rank = dist.get_rank() if dist is not None else 0
    os.makedirs(eval_args.get("save_folder", "samples"), exist_ok=True)
    gen = torch.Generator().manual_seed(a.seed)
    for i in range(a.batches):
        out = sample_batch(sampler, surf_vae, edge_vae, eval_args, gen, a.batch_size)
        if rank == 0:
            path = os.path.join(eval_args.get("save_folder", "samples"), f"latents_{a.mode}_{a.seed}_{i:04d}.npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {out['surfPos'].shape[0]} samples")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
