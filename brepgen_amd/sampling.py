"""The denoising cascade of sample.py:120-286 on the HIP path, batch-sharded across the GPUs of one node.

Stages (reference lines):  surfPos: PNDM[:158] -> late doubling -> DDPM[-250:]   (sample.py:126-153)
                           bbox de-dup on the host                               (sample.py:159-183)
                           surfZ:  PNDM x209                                     (sample.py:189-202)
                           edgePos: PNDM[:158] -> DDPM[-250:]                    (sample.py:208-236)
                           edge de-dup on the host                               (sample.py:242-261)
                           edgeZV: PNDM x209, zero removed                       (sample.py:267-286)

Multi-GPU: every op of the path is per-sample, so the batch is cut into contiguous per-rank slices
(`shard_range`), the initial / ancestral noise for the WHOLE batch is drawn once from one seeded CPU generator
and sliced (`sharded_randn`; an N-GPU run therefore reproduces the 1-GPU run sample for sample), and there is
exactly ONE exchange: `gather_latents`, a single flat all_gather (RCCL over xGMI) of the finished latents.
The VAE decode and the OpenCascade B-rep reconstruction that follow are outside this path.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .utils import randn_tensor


# --------------------------------------------------------------------------------------------------
# sharding / the one collective
# --------------------------------------------------------------------------------------------------
def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n samples owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_randn(shape, generator, rank, world, device):
    """Whole-batch CPU draw (reference seed semantics, utils.py:62-97), then this rank's slice."""
    full = randn_tensor(tuple(shape), generator=generator)          # CPU
    lo, hi = shard_range(shape[0], rank, world)
    return full[lo:hi].to(device)


def gather_latents(tensors, dist=None, group=None):
    """All-gather a dict of per-rank tensors (batch on dim 0, equal per-rank batch) with ONE collective.

    Everything is packed into one flat byte buffer (bool/uint8/fp32 alike) so the ring runs once with a large
    message instead of once per tensor.  Returns the dict with the full batch on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(tensors)
    world = dist.get_world_size(group)
    names = sorted(tensors)
    flat, meta = [], []
    for k in names:
        t = tensors[k].contiguous()
        b = t.view(torch.uint8) if t.dtype != torch.bool else t.view(torch.uint8)
        flat.append(b.reshape(-1))
        meta.append((k, t.dtype, tuple(t.shape), b.numel()))
    send = torch.cat(flat)
    pad = (-send.numel()) % 16
    if pad:
        send = torch.cat([send, send.new_zeros(pad)])
    recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, -1)
    out, off = {}, 0
    for k, dt, shape, nb in meta:
        piece = recv[:, off:off + nb].contiguous().view(dt)
        out[k] = piece.reshape((world * shape[0],) + shape[1:])
        off += nb
    return out


# --------------------------------------------------------------------------------------------------
# de-duplication between the stages.  On the device (bg_dedup_*: no host sync inside the cascade) for CUDA tensors;
# the numpy restatement of the reference's host loops is kept (`*_host`) as the checker of the device kernels.  Per sample -> shards with the batch.
# --------------------------------------------------------------------------------------------------
def dedup_surfaces(surfPos, threshold):
    """sample.py:159-183.  surfPos [B,S,6] -> (surfPos padded with 0 [B,S,6], surfMask bool [B,S])."""
    if not surfPos.is_cuda:
        raise _lib.BrepgenHipError("dedup_surfaces runs on the MI355X (use dedup_surfaces_host for host tensors)")
    B, S, _ = surfPos.shape
    x = surfPos.detach().to(torch.float32).contiguous()
    pos = torch.empty_like(x)
    mask = torch.empty(B, S, dtype=torch.uint8, device=x.device)
    check(_lib.load().bg_dedup_surfaces(ptr(x), float(np.float32(threshold)), ptr(pos), ptr(mask), B, S, stream()),
          "bg_dedup_surfaces")
    return pos, mask.view(torch.bool)


def dedup_edges(edgePos, surfMask, threshold):
    """sample.py:242-261.  -> edgeM bool [B,S,E], True = padded face or duplicate edge."""
    if not edgePos.is_cuda:
        raise _lib.BrepgenHipError("dedup_edges runs on the MI355X (use dedup_edges_host for host tensors)")
    B, S, E, _ = edgePos.shape
    x = edgePos.detach().to(torch.float32).contiguous()
    sm = surfMask.contiguous()
    sm = sm.view(torch.uint8) if sm.dtype == torch.bool else sm.to(torch.uint8)
    em = torch.empty(B, S, E, dtype=torch.uint8, device=x.device)
    check(_lib.load().bg_dedup_edges(ptr(x), ptr(sm), float(np.float32(threshold)), ptr(em), B, S, E, stream()),
          "bg_dedup_edges")
    return em.view(torch.bool)


def dedup_surfaces_host(surfPos, threshold):
    """sample.py:159-183.  surfPos [B,S,6] (device) -> (surfPos padded with 0 [B,S,6], surfMask bool [B,S])."""
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


# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def decode_latents(surf_vae, edge_vae, latents):
    """Stage 5 of sample.py (lines 286-294): VAE-decode the cascade's latents on the device.

    latents: the dict CascadeSampler.sample returns.  Adds surf_ncs [B,S,32,32,3], edge_ncs [B,S,E,32,3] and
    edgeV [B,S,E,6] (the vertex half of edgeZV, sample.py:286) and returns the dict.  The token layout of the latents
    (position-major, channel-minor) is the channels-last layout of the VAE kernels, so no permutes are needed."""
    out = dict(latents)
    out["surf_ncs"] = surf_vae.decode_tokens(latents["surfZ"])
    out["edge_ncs"] = edge_vae.decode_tokens(latents["edgeZV"][..., :12])
    out["edgeV"] = latents["edgeZV"][..., 12:].contiguous()
    return out


class CascadeSampler:
    """Runs stages 1-4 of sample.py on this rank's slice of the batch and all-gathers the latents."""

    def __init__(self, surfpos, surfz, edgepos, edgez, pndm, ddpm, *, use_cf=False, class_id=0, guidance=0.6,
                 bbox_threshold=0.08, dist=None, autocast=True):
        self.nets = (surfpos, surfz, edgepos, edgez)
        self.pndm, self.ddpm = pndm, ddpm
        self.use_cf, self.class_id, self.w = use_cf, class_id, guidance
        self.thr = bbox_threshold
        self.dist = dist
        self.autocast = autocast
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    # one guided / unguided eps evaluation + scheduler step
    def _step(self, sched, net_call, x, t, t_dev, **kw):
        if self.use_cf:
            eps = net_call(True)
            return sched.step(eps, t, x, guidance=self.w, **kw).prev_sample
        return sched.step(net_call(False), t, x, **kw).prev_sample

    def _labels(self, b, device):
        if not self.use_cf:
            return None
        return torch.tensor([self.class_id] * b + [0] * b, dtype=torch.int64, device=device).reshape(-1, 1)

    @staticmethod
    def _rep(t, n):
        return t.repeat(n, *([1] * (t.dim() - 1))).contiguous()

    @torch.no_grad()
    def sample(self, batch_size, num_surfaces, num_edges, generator=None, device="cuda",
               pndm_pos_steps=158, ddpm_pos_steps=250, pndm_z_steps=None, stop_after=None):
        surfpos_net, surfz_net, edgepos_net, edgez_net = self.nets
        dev = torch.device(device)
        lo, hi = shard_range(batch_size, self.rank, self.world)
        b = hi - lo
        cl = self._labels(b, dev)
        # autocast=True -> bf16 operands; a torch dtype (torch.float16: the reference's own autocast dtype) selects it
        if self.autocast is True:
            ctx = torch.autocast("cuda", dtype=torch.bfloat16)
        elif self.autocast:
            ctx = torch.autocast("cuda", dtype=self.autocast)
        else:
            ctx = torch.autocast("cuda", enabled=False)
        with ctx:
            # ---- 1-1 surface positions ----
            S = num_surfaces
            x = sharded_randn((batch_size, S, 6), generator, self.rank, self.world, dev)
            self.pndm.set_timesteps(200)
            for t in self.pndm.timesteps[:pndm_pos_steps]:
                td = t.reshape(-1).to(dev)
                x = self._step(self.pndm, lambda g: surfpos_net(self._rep(x, 2) if g else x, td, cl), x, t, td)
            if not self.use_cf:                                     # late doubling, sample.py:140-142
                x = x.repeat(1, 2, 1).contiguous()
                S *= 2
            self.ddpm.set_timesteps(1000)
            for t in self.ddpm.timesteps[-ddpm_pos_steps:]:
                td = t.reshape(-1).to(dev)
                z = sharded_randn((batch_size, S, 6), generator, self.rank, self.world, dev) if int(t) > 0 else None
                x = self._step(self.ddpm, lambda g: surfpos_net(self._rep(x, 2) if g else x, td, cl), x, t, td, noise=z)
            surfPos, surfMask = dedup_surfaces(x, self.thr)
            out = {"surfPos": surfPos, "surfMask": surfMask}
            if stop_after == "surfPos":
                return gather_latents(out, self.dist)

            # ---- 1-3 surface latents ----
            surfZ = sharded_randn((batch_size, S, 48), generator, self.rank, self.world, dev)
            sp2, sm2 = (self._rep(surfPos, 2), self._rep(surfMask, 2)) if self.use_cf else (surfPos, surfMask)
            self.pndm.set_timesteps(200)
            for t in self.pndm.timesteps[:pndm_z_steps]:
                td = t.reshape(-1).to(dev)
                surfZ = self._step(self.pndm, lambda g: surfz_net(self._rep(surfZ, 2) if g else surfZ, td, sp2, sm2, cl),
                                   surfZ, t, td)
            out["surfZ"] = surfZ
            if stop_after == "surfZ":
                return gather_latents(out, self.dist)

            # ---- 2-1 edge positions ----
            E = num_edges
            edgePos = sharded_randn((batch_size, S, E, 6), generator, self.rank, self.world, dev)
            sz2 = self._rep(surfZ, 2) if self.use_cf else surfZ
            self.pndm.set_timesteps(200)
            for t in self.pndm.timesteps[:pndm_pos_steps]:
                td = t.reshape(-1).to(dev)
                edgePos = self._step(self.pndm, lambda g: edgepos_net(self._rep(edgePos, 2) if g else edgePos, td, sp2,
                                                                       sz2, sm2, cl), edgePos, t, td)
            self.ddpm.set_timesteps(1000)
            for t in self.ddpm.timesteps[-ddpm_pos_steps:]:
                td = t.reshape(-1).to(dev)
                z = sharded_randn((batch_size, S, E, 6), generator, self.rank, self.world, dev) if int(t) > 0 else None
                edgePos = self._step(self.ddpm, lambda g: edgepos_net(self._rep(edgePos, 2) if g else edgePos, td, sp2,
                                                                       sz2, sm2, cl), edgePos, t, td, noise=z)
            edgeM = dedup_edges(edgePos, surfMask, self.thr)
            out.update(edgePos=edgePos, edgeM=edgeM)
            if stop_after == "edgePos":
                return gather_latents(out, self.dist)

            # ---- 2-3 edge latents + vertices ----
            edgeZV = sharded_randn((batch_size, S, E, 18), generator, self.rank, self.world, dev)
            ep2, em2 = (self._rep(edgePos, 2), self._rep(edgeM, 2)) if self.use_cf else (edgePos, edgeM)
            self.pndm.set_timesteps(200)
            for t in self.pndm.timesteps[:pndm_z_steps]:
                td = t.reshape(-1).to(dev)
                edgeZV = self._step(self.pndm, lambda g: edgez_net(self._rep(edgeZV, 2) if g else edgeZV, td, ep2, sp2,
                                                                    sz2, em2, cl), edgeZV, t, td)
            edgeZV = edgeZV.masked_fill(edgeM.unsqueeze(-1), 0.0)   # sample.py:284
            out["edgeZV"] = edgeZV
        return gather_latents(out, self.dist)
