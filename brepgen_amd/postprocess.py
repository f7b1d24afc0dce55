"""Device part of the reference's B-rep post-process.

``joint_optimize`` (utils.py:672-772) first fits the decoded edges to their corner vertices in numpy (cheap, stays in
the reference) and then runs 200 AdamW iterations on the GPU that slide every decoded surface onto its boundary edges
under a one-directional Chamfer loss (``chamferdist``, the only third-party CUDA kernel of the inference pipeline).
``optimize_surface_offsets`` is that loop as ONE kernel launch (``bg_chamfer_offset_fit``, csrc/chamfer.hip).
"""
import torch

from . import _lib
from ._lib import check, ptr, stream

# utils.py:680-686 / 749
ADAMW = dict(lr=1e-3, beta1=0.95, beta2=0.999, weight_decay=1e-6, eps=1e-8)
ITERS = 200


@torch.no_grad()
def optimize_surface_offsets(surf_wcs_init, face_edges, iters=ITERS, **adamw):
    """surf_wcs_init: [F, 32, 32, 3] (or [F, P, 3]) world-space surface points (utils.py:727-743); face_edges: list of F
    tensors [n_f, 32, 3] -- the fitted boundary edges of each face (utils.py:718-722).  Returns (surf_wcs shaped like the
    input, offsets [F,3], per-face Chamfer sums of the last iteration) -- ``surf_wcs`` is utils.py:770's ``surf_updated``."""
    if not surf_wcs_init.is_cuda:
        raise _lib.BrepgenHipError(f"brepgen_amd runs on the MI355X only (tensor on {surf_wcs_init.device}); no CPU fallback")
    hp = dict(ADAMW, **adamw)
    shape = surf_wcs_init.shape
    F = shape[0]
    if len(face_edges) != F:
        raise ValueError(f"{F} surfaces but {len(face_edges)} edge sets")
    surf = surf_wcs_init.detach().to(torch.float32).reshape(F, -1, 3).contiguous()
    P = surf.shape[1]
    dev = surf.device
    pts = [e.detach().to(device=dev, dtype=torch.float32).reshape(-1, 3) for e in face_edges]
    counts = [p.shape[0] for p in pts]
    edge_off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0).tolist()), dtype=torch.int32).to(dev)
    edge_pts = torch.cat(pts).contiguous() if sum(counts) else torch.zeros(1, 3, device=dev)
    offsets = torch.empty(F, 3, device=dev, dtype=torch.float32)
    out = torch.empty_like(surf)
    loss = torch.empty(F, device=dev, dtype=torch.float32)
    check(_lib.load().bg_chamfer_offset_fit(ptr(surf), ptr(edge_pts), ptr(edge_off), F, P, int(iters), hp["lr"], hp["beta1"],
                                            hp["beta2"], hp["weight_decay"], hp["eps"], ptr(offsets), ptr(out), ptr(loss),
                                            stream()), "bg_chamfer_offset_fit")
    return out.reshape(shape), offsets, loss
