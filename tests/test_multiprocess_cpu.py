"""N>1 path on CPU: world_size-2 gloo run of the sharding helpers and the single all-gather
(the per-rank compute itself has no CPU fallback, so a per-sample stand-in function plays the denoiser)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.dedup import dedup_edges_host as dedup_edges
from oracle.dedup import dedup_surfaces_host as dedup_surfaces
from brepgen_amd.sampling import gather_latents, shard_range, sharded_randn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stage(x):
    """Stand-in for a per-sample denoising stage: any function with no cross-sample mixing."""
    return torch.tanh(x * 1.7) + x.flip(-1) * 0.25


def _single_process(B):
    g = torch.Generator().manual_seed(2024)
    a = _stage(sharded_randn((B, 6, 5), g, 0, 1, "cpu"))
    b = _stage(sharded_randn((B, 6, 3, 4), g, 0, 1, "cpu"))
    return {"a": a, "b": b, "m": a[..., 0] > 0}


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(2024)          # same seed on every rank: whole-batch draw, then slice
        a = _stage(sharded_randn((B, 6, 5), g, rank, world, "cpu"))
        b = _stage(sharded_randn((B, 6, 3, 4), g, rank, world, "cpu"))
        out = gather_latents({"a": a, "b": b, "m": a[..., 0] > 0}, dist)
        # by value (numpy pickles its bytes): a torch tensor travels as a shared-memory handle the parent has to fetch from
        # this process, which may have exited by then
        q.put((rank, {k: v.numpy().copy() for k, v in out.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(420)
def test_two_rank_gloo_run_equals_single_process():
    B, world = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _single_process(B)
    for r in range(world):
        for k in want:
            assert got[r][k].dtype == want[k].dtype and got[r][k].shape == want[k].shape
            assert torch.equal(got[r][k], want[k]), f"rank {r} tensor {k}"


def test_shard_range_partitions():
    for n in (1, 7, 8, 512, 4096):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_is_identity_without_process_group():
    t = {"x": torch.arange(6.0).reshape(2, 3)}
    assert gather_latents(t, None)["x"] is t["x"]


def test_dedup_surfaces_matches_reference_semantics():
    # 5 boxes: #2 duplicates #0 within the threshold, #3 is #1 with its corners swapped
    base = torch.tensor([[0.0, 0, 0, 1, 1, 1], [2, 2, 2, 3, 3, 3], [0.05, 0, 0, 1, 1, 1.05], [3, 3, 3, 2, 2, 2],
                         [-1, -1, -1, 0, 0, 0]])
    pos, mask = dedup_surfaces(base[None], 0.08)
    assert mask[0].tolist() == [False, False, False, True, True]
    assert torch.allclose(pos[0, :3], base[[0, 1, 4]]) and float(pos[0, 3:].abs().max()) == 0


def test_dedup_edges_marks_duplicates_and_padded_faces():
    e = torch.zeros(1, 2, 4, 6)
    e[0, 0, 0] = torch.tensor([0.0, 0, 0, 1, 1, 1])
    e[0, 0, 1] = torch.tensor([1.0, 1, 1, 0, 0, 0])       # reversed duplicate of edge 0
    e[0, 0, 2] = torch.tensor([5.0, 5, 5, 6, 6, 6])
    e[0, 0, 3] = torch.tensor([5.0, 5, 5.01, 6, 6, 6])    # duplicate of edge 2
    smask = torch.tensor([[False, True]])
    m = dedup_edges(e, smask, 0.08)
    assert m[0, 0].tolist() == [False, True, False, True]
    assert m[0, 1].all()


def test_noise_key_advances_with_the_generator_and_consumes_nothing():
    """the device-side ancestral noise must differ between successive sample() calls that share one
    generator (pipeline.main --batches N), reproduce for equally seeded generators, and leave the generator untouched
    (the initial latents keep the reference's seed semantics)."""
    from brepgen_amd.sampling import noise_key
    g1, g2 = torch.Generator().manual_seed(77), torch.Generator().manual_seed(77)
    k1 = noise_key(g1)
    assert k1 == noise_key(g2) and 0 <= k1 < 2 ** 64
    st = g1.get_state().clone()
    assert torch.equal(st, g1.get_state()) and noise_key(g1) == k1          # nothing consumed, deterministic
    a = sharded_randn((4, 3), g1, 0, 1, "cpu")                            # what one sample() call draws
    k1b = noise_key(g1)
    assert k1b != k1                                                       # the next call gets fresh noise
    assert torch.equal(a, sharded_randn((4, 3), g2, 0, 1, "cpu")) and noise_key(g2) == k1b
    assert noise_key(torch.Generator().manual_seed(78)) != k1
    torch.manual_seed(5)                                                   # generator=None -> the global CPU generator
    ka = noise_key(None)
    torch.randn(3)
    assert noise_key(None) != ka
