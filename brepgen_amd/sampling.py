"""The denoising cascade of sample.py:120-286 on the HIP path, batch-sharded across the GPUs of one node.

Stages (reference lines):  surfPos: PNDM[:158] -> late doubling -> DDPM[-250:]   (sample.py:126-153)
                           bbox de-dup (on the device here)                      (sample.py:159-183)
                           surfZ:  PNDM x209                                     (sample.py:189-202)
                           edgePos: PNDM[:158] -> DDPM[-250:]                    (sample.py:208-236)
                           edge de-dup (on the device here)                      (sample.py:242-261)
                           edgeZV: PNDM x209, zero removed                       (sample.py:267-286)

Multi-GPU: every op of the path is per-sample, so the batch is cut into contiguous per-rank slices (`shard_range`;
sizes may differ by one, a rank may own nothing).  Noise:
  * the four INITIAL latents keep the reference's seed semantics (utils.py:62-97): one seeded CPU generator draws the
    whole batch, each rank keeps its slice (`sharded_randn`);
  * the ANCESTRAL noise of the 2 x 250 DDPM steps is drawn on the device, as upstream does (sample.py:153), by a
    counter-based generator keyed on (seed, draw number, GLOBAL sample index) (`device_randn` -> bg_philox_randn): a rank
    draws only its own rows, nothing crosses PCIe, and an N-GPU run still reproduces the 1-GPU run sample for sample.
    `noise_mode="reference"` restores the whole-batch CPU draw per step (parity mode against a CPU-driven cascade).
There is exactly ONE exchange: `gather_latents`, a single flat all_gather (RCCL over xGMI) of the finished latents
(and, when the caller decodes first, of the decoded point grids).  No step of the loops synchronises with the host.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream
from .utils import randn_tensor


# --------------------------------------------------------------------------------------------------
# sharding / noise / the one collective
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


def noise_key(generator=None):
    """64-bit key of the device-side ancestral noise of ONE sample() call.

    Derived from the generator's current STATE (the global CPU generator when none is given), not from its initial
    seed: the state has advanced by the initial-latent draws of every earlier call, so successive calls that share one
    generator get independent noise (upstream draws fresh noise from an advancing RNG, sample.py:153), while two
    generators seeded alike still reproduce each other.  Nothing is consumed -- the four initial-latent draws keep the
    reference's seed semantics -- and every rank holds the same state, so the key is rank-independent."""
    import hashlib
    st = (generator if generator is not None else torch.default_generator).get_state()
    return int.from_bytes(hashlib.blake2b(st.cpu().numpy().tobytes(), digest_size=8).digest(), "little")


def device_randn(shape, seed, draw_id, first_sample, device):
    """N(0,1) of `shape` (= this rank's rows) drawn on the device; row b is global sample first_sample + b."""
    out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if out.numel() == 0:
        return out
    if not out.is_cuda:
        raise _lib.BrepgenHipError("device_randn runs on the MI355X only; there is no CPU fallback")
    per = out[0].numel()
    check(_lib.load().bg_philox_randn(ptr(out), shape[0], per, int(seed) & 0xFFFFFFFFFFFFFFFF, int(draw_id) & 0xFFFFFFFF,
                                      int(first_sample), 0, stream()), "bg_philox_randn")
    return out


def gather_latents(tensors, dist=None, group=None, batch_size=None, single_rank_collective=False):
    """All-gather a dict of per-rank tensors (this rank's `shard_range` rows of a batch of `batch_size`, batch on
    dim 0) with ONE collective; returns the dict with the full batch on every rank.

    Everything is packed into one flat byte buffer (bool / uint8 / fp32 alike) so the ring runs once with a large
    message instead of once per tensor.  Ranks may own different numbers of rows (or none): every rank pads its rows
    to ceil(batch_size / world) before the collective -- all_gather needs equal contributions -- and the padding is
    trimmed with `shard_range` afterwards.  batch_size=None means equal shards (world * local rows).
    single_rank_collective: run the pack -> all_gather -> unpack path even with one rank (how a 1-GPU box tests the RCCL leg)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not single_rank_collective):
        return dict(tensors)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    names = sorted(tensors)
    local = tensors[names[0]].shape[0]
    if batch_size is None:
        batch_size = local * world
    lo, hi = shard_range(batch_size, rank, world)
    if hi - lo != local:
        raise ValueError(f"rank {rank} holds {local} rows but owns {hi - lo} of a batch of {batch_size}")
    rows = -(-batch_size // world)                                   # per-rank rows after padding
    flat, meta = [], []
    for k in names:
        t = tensors[k].contiguous()
        if t.shape[0] != local:
            raise ValueError(f"tensor {k!r} has {t.shape[0]} rows, expected {local}")
        if local < rows:
            t = torch.cat([t, t.new_zeros((rows - local,) + tuple(t.shape[1:]))])
        b = t.view(torch.uint8)
        flat.append(b.reshape(-1))
        meta.append((k, t.dtype, tuple(t.shape[1:]), b.numel()))
    send = torch.cat(flat)
    pad = (-send.numel()) % 16
    if pad:
        send = torch.cat([send, send.new_zeros(pad)])
    dev = send.device
    if dist.get_backend(group) == "nccl":                            # RCCL: device buffers, one ring pass over xGMI
        recv = torch.empty(world * send.numel(), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(recv, send, group=group)
    else:                                                            # gloo (CPU tests): host staging
        pieces = [torch.empty(send.numel(), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(pieces, send.cpu(), group=group)
        recv = torch.cat(pieces).to(dev)
    recv = recv.view(world, -1)
    counts = [b - a for a, b in (shard_range(batch_size, r, world) for r in range(world))]
    out, off = {}, 0
    for k, dt, tail, nb in meta:
        piece = recv[:, off:off + nb].contiguous().view(dt).reshape((world, rows) + tail)
        out[k] = piece.reshape((world * rows,) + tail) if rows * world == batch_size else \
            torch.cat([piece[r, :counts[r]] for r in range(world)])
        off += nb
    return out


# --------------------------------------------------------------------------------------------------
# de-duplication between the stages, on the device (bg_dedup_*: no host sync inside the cascade).  Per sample ->
# shards with the batch.  (The numpy restatement of the reference's host loops that checks these kernels lives in
# oracle/dedup.py.)
# --------------------------------------------------------------------------------------------------
def dedup_surfaces(surfPos, threshold):
    """sample.py:159-183.  surfPos [B,S,6] -> (surfPos padded with 0 [B,S,6], surfMask bool [B,S])."""
    if not surfPos.is_cuda:
        raise _lib.BrepgenHipError("dedup_surfaces runs on the MI355X only; there is no CPU fallback")
    B, S, _ = surfPos.shape
    x = surfPos.detach().to(torch.float32).contiguous()
    pos = torch.empty_like(x)
    mask = torch.empty(B, S, dtype=torch.uint8, device=x.device)
    if B > 0:
        check(_lib.load().bg_dedup_surfaces(ptr(x), float(np.float32(threshold)), ptr(pos), ptr(mask), B, S, stream()),
              "bg_dedup_surfaces")
    return pos, mask.view(torch.bool)


def dedup_edges(edgePos, surfMask, threshold):
    """sample.py:242-261.  -> edgeM bool [B,S,E], True = padded face or duplicate edge."""
    if not edgePos.is_cuda:
        raise _lib.BrepgenHipError("dedup_edges runs on the MI355X only; there is no CPU fallback")
    B, S, E, _ = edgePos.shape
    x = edgePos.detach().to(torch.float32).contiguous()
    sm = surfMask.contiguous()
    sm = sm.view(torch.uint8) if sm.dtype == torch.bool else sm.to(torch.uint8)
    em = torch.empty(B, S, E, dtype=torch.uint8, device=x.device)
    if B > 0:
        check(_lib.load().bg_dedup_edges(ptr(x), ptr(sm), float(np.float32(threshold)), ptr(em), B, S, E, stream()),
              "bg_dedup_edges")
    return em.view(torch.bool)


# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def decode_latents(surf_vae, edge_vae, latents, concurrent=True):
    """Stage 5 of sample.py (lines 286-294): VAE-decode the cascade's latents on the device.

    latents: the dict CascadeSampler.sample returns.  Adds surf_ncs [B,S,32,32,3], edge_ncs [B,S,E,32,3] and
    edgeV [B,S,E,6] (the vertex half of edgeZV, sample.py:286) and returns the dict.  The token layout of the latents
    (position-major, channel-minor) is the channels-last layout of the VAE kernels, so no permutes are needed."""
    out = dict(latents)
    surf_z, edge_z = latents["surfZ"], latents["edgeZV"][..., :12]
    if concurrent and surf_z.is_cuda and surf_z.numel() and edge_z.numel():
        # The two decodes are independent: the edge pass runs on a forked stream beside the surface pass and is joined before the
        # function returns (same kernels, same results).  Each pass alternates MFMA-bound convolutions with HBM-bound GroupNorm /
        # activation passes and tile-round tails; two of them in flight fill each other's gaps (the n_split of the denoisers).
        cur = torch.cuda.current_stream(surf_z.device)
        side = _side_stream(surf_z.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            edge = edge_vae.decode_tokens(edge_z)
        out["surf_ncs"] = surf_vae.decode_tokens(surf_z)
        cur.wait_stream(side)
        edge.record_stream(cur)                       # allocated on the side stream, consumed on the caller's
        out["edge_ncs"] = edge
    else:
        out["surf_ncs"] = surf_vae.decode_tokens(surf_z)
        out["edge_ncs"] = edge_vae.decode_tokens(edge_z)
    out["edgeV"] = latents["edgeZV"][..., 12:].contiguous()
    return out


_SIDE_STREAMS = {}


def _side_stream(device):
    """One helper stream per device for decode_latents' forked pass (created once: a stream per call would leak handles)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


class _GraphedEval:
    """One eps-evaluation captured into a hipGraph (via torch.cuda.graph): static latent / timestep buffers, one graph launch
    per step instead of ~90 kernel launches.  Everything else the evaluation reads (bbox / latent conditioning, masks, class
    labels) is captured by address and must stay alive and unchanged for the stage -- which is how the cascade uses it."""

    def __init__(self, fn, x, td):
        self.x, self.td = x.clone(), td.clone()
        cur, side = torch.cuda.current_stream(), torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            fn(self.x, self.td)                      # outside the capture: weight packing, workspace sizing
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn(self.x, self.td)

    def __call__(self, x, td):
        self.x.copy_(x)
        self.td.copy_(td)
        self.graph.replay()
        return self.out                              # consumed by the scheduler step before the next replay (same stream)


class CascadeSampler:
    """Runs stages 1-4 of sample.py on this rank's slice of the batch and (by default) all-gathers the latents.

    noise_mode: "device" (default) -- ancestral DDPM noise from the counter-based device generator, keyed on the global
    sample index (reproducible for any number of ranks); "reference" -- whole-batch draw from the seeded CPU generator at
    every step, sliced by rank (the reference's utils.randn_tensor semantics; what a CPU-driven oracle cascade can
    reproduce bit for bit, at the price of B x S x E x 6 host randoms and a PCIe copy per step on every rank).

    graphs: False (default) -- kernel-by-kernel launches; True -- every stage replays ONE captured hipGraph per step;
    "auto" -- only stages of <= GRAPH_MAX_TOKENS tokens.  Same kernels in the same order: results are bit-identical
    either way.  Off by default because it measured flat on the MI355X (whole DeepCAD cascade at the reference's batch
    16: 2.17 s without, 2.24 s with graphs; profiles/r02/cascade_b16_*.log): at that size a step is ~90 dependent
    kernels of a few microseconds each and the time is the GPU's own dispatch-to-dispatch latency, not the host's
    launch rate -- the option is for hosts that ARE launch-bound (slow CPU, many ranks per socket)."""

    GRAPH_MAX_TOKENS = 32768

    def __init__(self, surfpos, surfz, edgepos, edgez, pndm, ddpm, *, use_cf=False, class_id=0, guidance=0.6,
                 bbox_threshold=0.08, dist=None, autocast=True, noise_mode="device", graphs=False):
        if noise_mode not in ("device", "reference"):
            raise ValueError("noise_mode must be 'device' or 'reference'")
        if graphs not in ("auto", True, False):
            raise ValueError("graphs must be 'auto', True or False")
        self.graphs = graphs
        self.nets = (surfpos, surfz, edgepos, edgez)
        self.pndm, self.ddpm = pndm, ddpm
        # the nets' precomputed time-embedding tables must cover every timestep these schedulers can hand out
        t_max = max(int(sch.config.num_train_timesteps) for sch in (pndm, ddpm))
        for net in self.nets:
            if net is not None and getattr(net, "time_table_steps", 0) and net.time_table_steps < t_max:
                net.time_table_steps = t_max
        self.use_cf, self.class_id, self.w = use_cf, class_id, guidance
        self.thr = bbox_threshold
        self.dist = dist
        self.autocast = autocast
        self.noise_mode = noise_mode
        self.rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1

    # one guided / unguided eps evaluation + scheduler step
    def _step(self, sched, net_call, x, t, **kw):
        if self.use_cf:
            eps = net_call(True)
            return sched.step(eps, t, x, guidance=self.w, **kw).prev_sample
        return sched.step(net_call(False), t, x, **kw).prev_sample

    def _evaluator(self, fn, x, td):
        """fn(latent, timestep) -> eps, as is or captured into a graph (x / td: example inputs of the stage)."""
        tokens = x.numel() // max(1, x.shape[-1]) * (2 if self.use_cf else 1)
        use = self.graphs is True or (self.graphs == "auto" and tokens <= self.GRAPH_MAX_TOKENS)
        if not use or x.shape[0] == 0:
            return fn
        return _GraphedEval(fn, self._rep(x, 2) if self.use_cf else x, td)

    def _labels(self, b, device):
        if not self.use_cf:
            return None
        return torch.tensor([self.class_id] * b + [0] * b, dtype=torch.int64, device=device).reshape(-1, 1)

    @staticmethod
    def _rep(t, n):
        return t.repeat(n, *([1] * (t.dim() - 1))).contiguous()

    def _empty(self, S2, E, dev, stop_after):
        """What a rank that owns no sample contributes (it skips the compute but still joins the collective)."""
        f = lambda *s: torch.empty((0,) + s, dtype=torch.float32, device=dev)
        m = lambda *s: torch.empty((0,) + s, dtype=torch.bool, device=dev)
        out = {"surfPos": f(S2, 6), "surfMask": m(S2)}
        if stop_after == "surfPos":
            return out
        out["surfZ"] = f(S2, 48)
        if stop_after == "surfZ":
            return out
        out.update(edgePos=f(S2, E, 6), edgeM=m(S2, E))
        if stop_after == "edgePos":
            return out
        out["edgeZV"] = f(S2, E, 18)
        return out

    @torch.no_grad()
    def sample(self, batch_size, num_surfaces, num_edges, generator=None, device="cuda",
               pndm_pos_steps=158, ddpm_pos_steps=250, pndm_z_steps=None, stop_after=None, gather=True, timings=None):
        """-> dict of latents: the whole batch on every rank (gather=True), or this rank's rows only (gather=False;
        pass them -- plus anything derived per sample, e.g. the VAE decode -- to `gather_latents(..., batch_size=)`).
        timings: a dict to receive the wall-clock seconds of each stage (adds one device synchronisation per stage)."""
        import time
        clock = [None]

        def mark(stage):
            if timings is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                if stage is not None:
                    timings[stage] = now - clock[0]
                clock[0] = now

        mark(None)
        surfpos_net, surfz_net, edgepos_net, edgez_net = self.nets
        dev = torch.device(device)
        lo, hi = shard_range(batch_size, self.rank, self.world)
        b = hi - lo
        finish = (lambda o: gather_latents(o, self.dist, batch_size=batch_size)) if gather else (lambda o: o)
        # every rank consumes the CPU generator identically (4 whole-batch draws), whether or not it owns samples
        seed = noise_key(generator)
        draw = [0]

        def ancestral(shape, t):
            """noise of one DDPM step for this rank's rows (None at t == 0, where upstream adds no variance)."""
            draw[0] += 1
            if int(t) == 0:
                return None
            if self.noise_mode == "reference":                      # every rank consumes the generator identically
                return sharded_randn((batch_size,) + shape, generator, self.rank, self.world, dev)
            return device_randn((b,) + shape, seed, draw[0], lo, dev)

        cl = self._labels(b, dev)
        # autocast=True -> bf16 operands; a torch dtype (torch.float16: the reference's own autocast dtype) selects it
        if self.autocast is True:
            ctx = torch.autocast("cuda", dtype=torch.bfloat16)
        elif self.autocast:
            ctx = torch.autocast("cuda", dtype=self.autocast)
        else:
            ctx = torch.autocast("cuda", enabled=False)
        skip = b == 0
        with ctx:
            # ---- 1-1 surface positions ----
            S = num_surfaces
            x = sharded_randn((batch_size, S, 6), generator, self.rank, self.world, dev)
            self.pndm.set_timesteps(200)
            pndm_ts, pndm_dev = self.pndm.timesteps, self.pndm.timesteps.to(dev)     # ONE copy; steps take views
            ev = None if skip else self._evaluator(lambda xi, ti: surfpos_net(xi, ti, cl), x, pndm_dev[:1])
            for i, t in enumerate(pndm_ts[:pndm_pos_steps]):
                if skip:
                    break
                td = pndm_dev[i:i + 1]
                x = self._step(self.pndm, lambda g: ev(self._rep(x, 2) if g else x, td), x, t)
            if not self.use_cf:                                     # late doubling, sample.py:140-142
                x = x.repeat(1, 2, 1).contiguous()
                S *= 2
            self.ddpm.set_timesteps(1000)
            ddpm_ts = self.ddpm.timesteps[-ddpm_pos_steps:]
            ddpm_dev = ddpm_ts.to(dev)
            if not skip and not self.use_cf:                        # the token count doubled: a new capture
                ev = self._evaluator(lambda xi, ti: surfpos_net(xi, ti, cl), x, ddpm_dev[:1])
            for i, t in enumerate(ddpm_ts):
                z = ancestral((S, 6), t)
                if skip:
                    continue
                td = ddpm_dev[i:i + 1]
                x = self._step(self.ddpm, lambda g: ev(self._rep(x, 2) if g else x, td), x, t, noise=z)
            surfPos, surfMask = dedup_surfaces(x, self.thr)
            out = {"surfPos": surfPos, "surfMask": surfMask}
            mark("surfPos")
            if stop_after == "surfPos":
                return finish(out)

            # ---- 1-3 surface latents ----
            surfZ = sharded_randn((batch_size, S, 48), generator, self.rank, self.world, dev)
            sp2, sm2 = (self._rep(surfPos, 2), self._rep(surfMask, 2)) if self.use_cf else (surfPos, surfMask)
            self.pndm.set_timesteps(200)
            ev = None if skip else self._evaluator(lambda xi, ti: surfz_net(xi, ti, sp2, sm2, cl), surfZ, pndm_dev[:1])
            for i, t in enumerate(pndm_ts[:pndm_z_steps]):
                if skip:
                    break
                td = pndm_dev[i:i + 1]
                surfZ = self._step(self.pndm, lambda g: ev(self._rep(surfZ, 2) if g else surfZ, td), surfZ, t)
            out["surfZ"] = surfZ
            mark("surfZ")
            if stop_after == "surfZ":
                return finish(out)

            # ---- 2-1 edge positions ----
            E = num_edges
            edgePos = sharded_randn((batch_size, S, E, 6), generator, self.rank, self.world, dev)
            sz2 = self._rep(surfZ, 2) if self.use_cf else surfZ
            self.pndm.set_timesteps(200)
            ev = None if skip else self._evaluator(lambda xi, ti: edgepos_net(xi, ti, sp2, sz2, sm2, cl), edgePos, pndm_dev[:1])
            for i, t in enumerate(pndm_ts[:pndm_pos_steps]):
                if skip:
                    break
                td = pndm_dev[i:i + 1]
                edgePos = self._step(self.pndm, lambda g: ev(self._rep(edgePos, 2) if g else edgePos, td), edgePos, t)
            self.ddpm.set_timesteps(1000)
            for i, t in enumerate(ddpm_ts):
                z = ancestral((S, E, 6), t)
                if skip:
                    continue
                td = ddpm_dev[i:i + 1]
                edgePos = self._step(self.ddpm, lambda g: ev(self._rep(edgePos, 2) if g else edgePos, td), edgePos, t, noise=z)
            edgeM = dedup_edges(edgePos, surfMask, self.thr)
            out.update(edgePos=edgePos, edgeM=edgeM)
            mark("edgePos")
            if stop_after == "edgePos":
                return finish(out)

            # ---- 2-3 edge latents + vertices ----
            edgeZV = sharded_randn((batch_size, S, E, 18), generator, self.rank, self.world, dev)
            ep2, em2 = (self._rep(edgePos, 2), self._rep(edgeM, 2)) if self.use_cf else (edgePos, edgeM)
            self.pndm.set_timesteps(200)
            ev = None if skip else self._evaluator(lambda xi, ti: edgez_net(xi, ti, ep2, sp2, sz2, em2, cl), edgeZV, pndm_dev[:1])
            for i, t in enumerate(pndm_ts[:pndm_z_steps]):
                if skip:
                    break
                td = pndm_dev[i:i + 1]
                edgeZV = self._step(self.pndm, lambda g: ev(self._rep(edgeZV, 2) if g else edgeZV, td), edgeZV, t)
            edgeZV = edgeZV.masked_fill(edgeM.unsqueeze(-1), 0.0)   # sample.py:284
            out["edgeZV"] = edgeZV
            mark("edgeZV")
        return finish(out)
