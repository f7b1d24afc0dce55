"""Forward half of the reference's LDM trainers on the device: VAE encode -> latent tokens
-> ``add_noise`` -> eps-net -> masked MSE, i.e. everything ``trainer.py`` does between loading a batch and calling
``backward()``, and all of its ``test_val`` loops (trainer.py:374-408, 557-602, 752-797, 975-1030).  There is no backward
pass here -- the package implements the denoising (inference) path; this module lets the same kernels compute the
training / validation *losses* of a checkpoint.

Every function takes and returns device tensors and enqueues on the current stream; nothing synchronises.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream

VAL_STEPS = (10, 50, 100, 200, 500)          # timesteps the reference validates at (trainer.py:394, 587, 781, 1008)


def masked_mse(pred, target, mask=None, cols=None):
    """``nn.MSELoss()(pred[~mask], target[~mask])`` and the ``test_val`` reduction, from one deterministic kernel pair.

    pred / target: fp32 [..., C]; mask: bool [...] (True = padded) or None; cols: (start, stop) column slice or None
    (EdgeZTrainer also reports the latent 0:12 and vertex 12:18 parts, trainer.py:951-952).
    Returns {"mean": scalar tensor, "row_mean_sum": sum over valid rows of the per-row mean, "rows": valid rows}."""
    if not pred.is_cuda:
        raise _lib.BrepgenHipError(f"brepgen_amd runs on the MI355X only (tensor on {pred.device}); no CPU fallback")
    assert pred.shape == target.shape and pred.dtype == torch.float32 and target.dtype == torch.float32
    C = pred.shape[-1]
    c0, c1 = cols if cols is not None else (0, C)
    p, t = pred.contiguous(), target.contiguous()
    rows = p.numel() // C
    mk = None
    if mask is not None:
        assert mask.shape == pred.shape[:-1]
        mk = mask.contiguous()
        mk = mk.view(torch.uint8) if mk.dtype == torch.bool else mk.to(torch.uint8)
    scratch = torch.empty(1024, device=p.device, dtype=torch.float64)
    out = torch.empty(3, device=p.device, dtype=torch.float32)
    check(_lib.load().bg_masked_mse(ptr(p), ptr(t), ptr(mk), rows, C, c0, c1 - c0, ptr(scratch), ptr(out), stream()),
          "bg_masked_mse")
    return {"mean": out[0], "row_mean_sum": out[1], "rows": out[2]}


def augment(ddpm, conditions, generator=None, max_t=15):
    """Conditioning augmentation of the Z / edge trainers (trainer.py:507-514, 935-941): a little forward diffusion
    (t ~ U{0..14}) on every conditioning tensor.  Draws on the CPU generator like utils.randn_tensor."""
    out = []
    for x in conditions:
        t = torch.randint(0, max_t, (x.shape[0],), generator=generator)
        noise = torch.randn(x.shape, generator=generator).to(x.device)
        out.append(ddpm.add_noise(x, noise, t))
    return out


@torch.no_grad()
def ldm_loss(net, ddpm, x0, timesteps, noise, conditions=(), mask=None, class_label=None, is_train=False, cols=None):
    """One loss evaluation of any of the four LDM trainers: diffuse x0 to per-sample ``timesteps`` with ``noise``
    (trainer.py:346-348), predict it with ``net(x_t, timesteps, *conditions[, mask], class_label, is_train)``
    (trainer.py:351, 534, 729, 947) and compare on the un-padded rows (trainer.py:354, 537, 732, 950)."""
    x_t = ddpm.add_noise(x0, noise, timesteps)
    args = [x_t, timesteps, *conditions] + ([mask] if mask is not None else []) + [class_label]
    pred = net(*args, is_train) if is_train else net(*args)
    row_mask = mask
    if mask is not None and mask.dim() < pred.dim() - 1:           # EdgePosNet: [B,S] face mask over [B,S,E,6] edges
        row_mask = mask.unsqueeze(-1).expand(pred.shape[:-1])
    return masked_mse(pred, noise.to(pred.device), row_mask, cols)


@torch.no_grad()
def validation_losses(net, ddpm, x0, conditions=(), mask=None, class_label=None, generator=None, steps=VAL_STEPS):
    """The body of ``test_val`` for one batch: loss at t = step-1 for step in (10, 50, 100, 200, 500) with fresh noise
    each (trainer.py:394-401, 587-597).  Returns a list of dicts as from masked_mse, one per step."""
    B = x0.shape[0]
    out = []
    for step in steps:
        t = torch.full((B,), step - 1, dtype=torch.int64, device=x0.device)       # randint(step-1, step) has one value
        noise = torch.randn(x0.shape, generator=generator).to(x0.device)
        out.append(ldm_loss(net, ddpm, x0, t, noise, conditions, mask, class_label))
    return out


@torch.no_grad()
def surface_tokens(surf_vae_encoder, surfPnt, z_scaled=1.0):
    """trainer.py:519-526: point grids [B,S,32,32,3] -> VAE posterior mode -> rescaled latent tokens [B,S,48]."""
    return surf_vae_encoder.encode_tokens(surfPnt) * z_scaled


@torch.no_grad()
def edge_tokens(edge_vae_encoder, edgePnt, vertPos, z_scaled=1.0):
    """trainer.py:924-933: polylines [B,S,E,32,3] -> latent tokens [B,S,E,12], concatenated with the 6 vertex
    coordinates -> the 18-channel joint data EdgeZNet denoises."""
    return torch.cat([edge_vae_encoder.encode_tokens(edgePnt) * z_scaled, vertPos], -1)
