"""BASELINE.json configs[2..4] at their FULL shapes, checked through size-independent properties (the oracle cannot
finish these sizes): per-sample independence of the denoisers (a batch row equals the same sample run alone), mask
invariants of the cascade, layout equivalence of the VAE token decode, finiteness.  Step counts are shortened -- the
properties do not depend on them -- so the whole file stays within about a minute of GPU time.
"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

F16, BF16 = torch.float16, torch.bfloat16


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import parity_cases
    return parity_cases


def _cuda(args):
    return [a.cuda() if torch.is_tensor(a) else a for a in args]


def _row(args, b, keep):
    """Sample b of every per-sample argument (timesteps / None stay as they are)."""
    return [a[b:b + 1].contiguous() if (torch.is_tensor(a) and i in keep) else a for i, a in enumerate(args)]


@pytest.mark.parametrize("cfg", ["cfg3_deepcad_edgez", "cfg4_abc_edgepos", "cfg5_furniture_cfg_fp16"])
def test_edge_nets_full_size_properties(pc, cfg):
    """cfg3: EdgeZNet B=256, S=60, E=30 (1800 tokens) bf16; cfg4: EdgePosNet 512/rank, S=100, E=40 (4000 tokens) bf16;
    cfg5: EdgeZNet with class labels, 256/rank doubled to 512 rows, S=60, E=40 (2400 tokens), fp16."""
    net, B, S, E, dt, cf = {"cfg3_deepcad_edgez": ("EdgeZNet", 256, 60, 30, BF16, False),
                            "cfg4_abc_edgepos": ("EdgePosNet", 512, 100, 40, BF16, False),
                            "cfg5_furniture_cfg_fp16": ("EdgeZNet", 512, 60, 40, F16, True)}[cfg]
    m, _ = pc.build_net(net, 9, cf, dt)
    args = _cuda(pc.synth_inputs(net, B, S, E, cf))
    per_sample = {0, 2, 3, 4} if net == "EdgePosNet" else {0, 2, 3, 4, 5}
    mask = args[4][:, :, None].expand(B, S, E) if net == "EdgePosNet" else args[5]
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        full = m(*args)
        torch.cuda.synchronize()
        dt_full = time.perf_counter() - t0
        assert full.shape == args[0].shape and torch.isfinite(full).all()
        worst = 0.0
        for b in (0, B // 2 - 1, B - 1):
            a1 = _row(args, b, per_sample)
            if cf:
                a1[-1] = args[-1][b:b + 1].contiguous()
            one = m(*a1)
            worst = max(worst, float((one[0] - full[b])[~mask[b]].abs().max()))
    print(f"{cfg}: {net} B={B} tokens={S * E} {dt}: {dt_full * 1e3:.0f} ms per eval, batch-vs-single max |d| = {worst:.2e}")
    assert worst < 2e-3          # same kernels, same K order: expected 0; bound = one 16-bit rounding flip downstream


def test_cfg3_cascade_and_decode_full_shapes(pc):
    """configs[2]: B=256, 30 faces doubled to 60, 30 edges per face, bf16; then VAE decode of all 15 360 faces and
    460 800 edges.  Schedules are cut to 13 PNDM evaluations (12 PRK + 1 PLMS) and 3 DDPM steps per stage."""
    import brepgen_amd as bga
    from brepgen_amd.sampling import CascadeSampler, decode_latents
    from oracle import vae as ov
    B, S, E = 256, 30, 30
    nets = [pc.build_net(n, 70 + i, False, None)[0] for i, n in enumerate(["SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet"])]
    for n in nets:
        n.compute_dtype = None                      # follow autocast, as sample.py does
    kw = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001, beta_end=0.02)
    sampler = CascadeSampler(*nets, bga.PNDMScheduler(**kw), bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **kw),
                             autocast=True)
    t0 = time.perf_counter()
    lat = sampler.sample(B, S, E, generator=torch.Generator().manual_seed(5), pndm_pos_steps=13, ddpm_pos_steps=3,
                         pndm_z_steps=13)
    torch.cuda.synchronize()
    t_cascade = time.perf_counter() - t0
    S2 = 2 * S
    assert lat["surfPos"].shape == (B, S2, 6) and lat["surfZ"].shape == (B, S2, 48)
    assert lat["edgePos"].shape == (B, S2, E, 6) and lat["edgeZV"].shape == (B, S2, E, 18)
    sm, em = lat["surfMask"], lat["edgeM"]
    assert sm.dtype == torch.bool and em.dtype == torch.bool
    assert bool((~sm).any(1).all())                                   # every sample keeps at least one face
    assert bool(em[sm].all())                                         # edges of padded faces are padded
    assert bool((~em[~sm]).any(-1).all())                             # every valid face keeps at least one edge
    assert bool((lat["edgeZV"][em] == 0).all())                       # sample.py:284
    for k in ("surfPos", "surfZ", "edgePos", "edgeZV"):
        assert torch.isfinite(lat[k]).all(), k

    surf_vae = bga.AutoencoderKLFastDecode(**pc.SURF_CFG)
    surf_vae.load_state_dict(ov.seeded_state_dict(ov.surf_decoder_spec(), 31), strict=True)
    edge_vae = bga.AutoencoderKL1DFastDecode(**pc.EDGE_CFG)
    edge_vae.load_state_dict(ov.seeded_state_dict(ov.edge_decoder_spec(), 41), strict=True)
    surf_vae, edge_vae = surf_vae.cuda().eval(), edge_vae.cuda().eval()
    surf_vae.compute_dtype = edge_vae.compute_dtype = BF16
    t0 = time.perf_counter()
    dec = decode_latents(surf_vae, edge_vae, lat)
    torch.cuda.synchronize()
    t_decode = time.perf_counter() - t0
    assert dec["surf_ncs"].shape == (B, S2, 32, 32, 3) and dec["edge_ncs"].shape == (B, S2, E, 32, 3)
    assert dec["edgeV"].shape == (B, S2, E, 6)
    assert torch.isfinite(dec["surf_ncs"]).all() and torch.isfinite(dec["edge_ncs"]).all()
    # a15: the token decode equals the reference's permute chain through the NCHW call surface (sample.py:289-294)
    with torch.no_grad():
        z = lat["surfZ"][:2]
        ref = surf_vae(z.unflatten(-1, torch.Size([16, 3])).flatten(0, 1).permute(0, 2, 1).unflatten(-1, torch.Size([4, 4])))
        ref = ref.permute(0, 2, 3, 1).unflatten(0, torch.Size([2, S2]))
        d_s = float((ref - dec["surf_ncs"][:2]).abs().max())
        ez = lat["edgeZV"][:1, :, :, :12]
        ref = edge_vae(ez.unflatten(-1, torch.Size([4, 3])).reshape(-1, 4, 3).permute(0, 2, 1))
        ref = ref.permute(0, 2, 1).reshape(1, S2, E, 32, 3)
        d_e = float((ref - dec["edge_ncs"][:1]).abs().max())
    assert d_s < 5e-3 and d_e < 5e-3, (d_s, d_e)      # a layout slip would be O(1); expected 0 (same kernels)
    print(f"cfg3 full shapes: cascade (13+3+13 / 13+3+13 evaluations) {t_cascade:.2f} s, "
          f"VAE decode 15360 faces + 460800 edges {t_decode:.2f} s; token-vs-NCHW decode |d| = {d_s:.1e} / {d_e:.1e}")
