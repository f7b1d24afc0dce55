"""Numpy restatement of the reference's host-side bbox de-duplication loops (sample.py:159-183, 242-261).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): the checker of the device kernels ``bg_dedup_surfaces`` /
``bg_dedup_edges`` (csrc/dedup.hip), which must reproduce these loops bit for bit.  PINNED by construction to the
reference's arithmetic: ``np.round(., 4)`` on the face bboxes, greedy keep-list, L-infinity threshold on both corner
orders.  Nothing under ``brepgen_amd/`` imports this module.
"""
import numpy as np
import torch


def dedup_surfaces_host(surfPos, threshold):
    """sample.py:159-183.  surfPos [B,S,6] -> (surfPos padded with 0 [B,S,6], surfMask bool [B,S], True = padded)."""
    B, S, _ = surfPos.shape
    host = np.round(surfPos.detach().float().cpu().numpy().reshape(B, S, 2, 3), 4)
    pos = np.zeros((B, S, 6), dtype=np.float32)
    mask = np.ones((B, S), dtype=bool)
    for b in range(B):
        keep = [host[b, 0]]
        for bbox in host[b]:
            cur = np.stack(keep)
            same = np.abs(cur - bbox).max(-1).max(-1) < threshold
            same_rev = np.abs(cur - bbox[::-1]).max(-1).max(-1) < threshold
            if not (same.any() or same_rev.any()):
                keep.append(bbox)
        k = len(keep)
        pos[b, :k] = np.stack(keep).reshape(k, 6)
        mask[b, :k] = False
    return torch.from_numpy(pos).to(surfPos.device), torch.from_numpy(mask).to(surfPos.device)


def dedup_edges_host(edgePos, surfMask, threshold):
    """sample.py:242-261.  -> edgeM bool [B,S,E], True = padded face or duplicate edge."""
    B, S, E, _ = edgePos.shape
    host = edgePos.detach().float().cpu().numpy().reshape(B, S, E, 2, 3)
    smask = surfMask.cpu().numpy()
    edgeM = np.repeat(smask[:, :, None], E, axis=2).copy()
    for b in range(B):
        valid_faces = np.nonzero(~smask[b])[0]
        # the reference indexes edgeM with the position inside the list of valid faces (sample.py:246,257);
        # valid faces are left-aligned after dedup_surfaces, so position == face index
        for idx, s in enumerate(valid_faces):
            keep = [host[b, s, 0]]
            for e in range(E):
                bbox = host[b, s, e]
                cur = np.stack(keep)
                same = np.abs(cur - bbox).max(-1).max(-1) < threshold
                same_rev = np.abs(cur - bbox[::-1]).max(-1).max(-1) < threshold
                if same.any() or same_rev.any():
                    edgeM[b, idx, e] = True
                else:
                    keep.append(bbox)
            edgeM[b, idx, 0] = False
    return torch.from_numpy(edgeM).to(edgePos.device)
