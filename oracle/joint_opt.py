"""CPU restatement of the GPU loop of the reference's ``joint_optimize`` (utils.py:746-772): per-face 3-D offsets that pull
each decoded surface point grid onto its boundary edges, fitted with 200 AdamW steps on a one-directional Chamfer loss.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED for the loss: it comes from the third-party package
``chamferdist`` (krrish94; ``pip install chamferdist``, README.md:33 -- no version pinned, not vendored, not installable
here).  Restated from its published 1.0.x semantics: ``ChamferDistance()(source, target, bidirectional=False,
reverse=True)`` = for every *target* point the squared L2 distance to its nearest *source* point, summed over the points,
mean over the batch (the call site, utils.py:761, passes one cloud pair at a time, so the batch mean is the identity).
The optimiser is ``torch.optim.AdamW(lr=1e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)`` (utils.py:680-686);
its arithmetic is restated below and pinned against torch's own AdamW + autograd by tests/test_oracle_joint_opt.py.

Reference mapping:
    model.surf_st[:, 1:]  -> offsets [F,3] (initial 0; the scale column surf_st[:, 0] never enters the forward pass)
    surf                  -> surf  [F,P,3] initial world-space surface points (utils.py:727-743)
    face_edges[f]         -> edges[f] [Q_f,3] boundary points of face f (constant, .detach()-ed at utils.py:760)
    loss                  -> mean over faces of sum_e min_s |e - (s + offset_f)|^2          (utils.py:756-762)
"""
import numpy as np

LR, BETA1, BETA2, WEIGHT_DECAY, EPS, ITERS = 1e-3, 0.95, 0.999, 1e-6, 1e-8, 200


def chamfer_reverse(surf_pts, edge_pts):
    """sum over edge points of the squared distance to the nearest surface point; also returns the argmin indices."""
    d = ((edge_pts[:, None, :] - surf_pts[None, :, :]) ** 2).sum(-1)          # [Q,P]
    idx = d.argmin(1)
    return d[np.arange(len(edge_pts)), idx].sum(), idx


def optimize_surface_offsets(surf, edges, iters=ITERS, lr=LR, beta1=BETA1, beta2=BETA2, weight_decay=WEIGHT_DECAY, eps=EPS):
    """surf [F,P,3] float32, edges: list of F arrays [Q_f,3].  Returns (surf + offsets [F,P,3], offsets [F,3], losses).

    Like the reference, the returned surface is the one evaluated in the LAST iteration (``surf_updated``, utils.py:751,
    770), i.e. with the offsets *before* the final optimiser step."""
    surf = np.asarray(surf, np.float32)
    F = surf.shape[0]
    off = np.zeros((F, 3), np.float32)
    m = np.zeros((F, 3), np.float32)
    v = np.zeros((F, 3), np.float32)
    losses = []
    used = off.copy()
    for it in range(1, iters + 1):
        used = off.copy()
        grad = np.zeros((F, 3), np.float32)
        loss = 0.0
        for f in range(F):
            s = surf[f] + off[f][None, :]
            e = np.asarray(edges[f], np.float32)
            val, idx = chamfer_reverse(s, e)
            loss += float(val)
            grad[f] = (-2.0 * (e - s[idx])).sum(0) / F                        # d/d off_f of (1/F) sum_e |e - s* - off|^2
        losses.append(loss / F)
        # torch.optim.AdamW (decoupled weight decay, bias-corrected moments)
        off = off * np.float32(1.0 - lr * weight_decay)
        m = m + (grad - m) * np.float32(1.0 - beta1)                          # exp_avg.lerp_(grad, 1 - beta1)
        v = v * np.float32(beta2) + grad * grad * np.float32(1.0 - beta2)
        bc1, bc2 = 1.0 - beta1 ** it, 1.0 - beta2 ** it
        denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
        off = (off - np.float32(lr / bc1) * m / denom).astype(np.float32)
    return surf + used[:, None, :], used, losses
