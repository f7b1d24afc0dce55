"""oracle/joint_opt.py (Chamfer offset fit of utils.py:746-772) pinned against torch autograd + torch.optim.AdamW driving
the same loss written with torch ops.  (The loss semantics themselves come from the un-vendored `chamferdist` package:
parity unpinned, see the oracle's header.)"""
import numpy as np
import torch

from oracle import joint_opt as jo


def _case(F=3, P=40, seed=0):
    g = np.random.default_rng(seed)
    surf = g.normal(size=(F, P, 3)).astype(np.float32)
    edges = [(g.normal(size=(int(g.integers(5, 20)), 3)) * 0.7 + g.normal(size=(1, 3)) * 0.3).astype(np.float32) for _ in range(F)]
    return surf, edges


def test_offsets_follow_torch_adamw_and_autograd():
    surf, edges = _case()
    iters = 60
    got_surf, got_off, losses = jo.optimize_surface_offsets(surf, edges, iters=iters)
    # the reference's own loop (utils.py:746-770) with the Chamfer loss spelled out in torch
    st = torch.nn.Parameter(torch.tensor([1.0, 0, 0, 0]).unsqueeze(0).repeat(len(surf), 1))
    opt = torch.optim.AdamW([st], lr=1e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    S = torch.from_numpy(surf)
    ref_losses = []
    for _ in range(iters):
        upd = S + st[:, 1:].reshape(-1, 1, 3)
        loss = 0
        for s, e in zip(upd, edges):
            d = torch.cdist(torch.from_numpy(e)[None], s[None])[0] ** 2
            loss = loss + d.min(1).values.sum()
        loss = loss / len(upd)
        ref_losses.append(float(loss.detach()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert np.allclose(got_surf, upd.detach().numpy(), atol=2e-5)
    assert np.allclose(losses, ref_losses, rtol=2e-4)
    assert losses[-1] < losses[0]                                             # the fit makes progress


def test_single_point_case_has_a_closed_form_first_step():
    # one surface point at the origin, one edge point at (1,0,0): grad = -2, first AdamW step moves by exactly lr
    surf = np.zeros((1, 1, 3), np.float32)
    _, off, losses = jo.optimize_surface_offsets(surf, [np.array([[1.0, 0, 0]], np.float32)], iters=2)
    assert abs(losses[0] - 1.0) < 1e-7 and abs(off[0, 0] - 1e-3) < 1e-7 and off[0, 1] == 0 and off[0, 2] == 0
