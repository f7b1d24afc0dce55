"""-m gpu: the whole denoising cascade (sample.py:120-286) on the HIP path vs the same cascade driven by the oracle
(oracle nets -- their fp32 forwards evaluated with plain torch on the device -- + restated schedulers, de-duplication and
the same seeded noise on the CPU), fp32, shortened schedules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _oracle_cascade(sds, B, S, E, gen, use_cf, class_id, w, n_pos, n_ddpm, n_z, thr=0.08):
    """Reference order of operations, CPU: sample.py:126-286 with the oracle in place of network.py/diffusers."""
    from oracle.dedup import dedup_edges_host as dedup_edges
    from oracle.dedup import dedup_surfaces_host as dedup_surfaces
    from brepgen_amd.utils import randn_tensor
    from oracle import denoisers as orc
    from oracle.schedulers import OracleDDPM, OraclePNDM
    from parity_cases import oracle_on_device as on_dev           # the oracle's fp32 forwards run on the GPU (~70 tiny evaluations)
    pndm, ddpm = OraclePNDM(), OracleDDPM(clip_sample=True, clip_sample_range=3)
    cl = torch.tensor([class_id] * B + [0] * B).reshape(-1, 1) if use_cf else None
    rep = (lambda t: t.repeat(2, *([1] * (t.dim() - 1)))) if use_cf else (lambda t: t)

    def guided(e):
        return e[:B] * (1 + w) - e[B:] * w if use_cf else e

    x = randn_tensor((B, S, 6), generator=gen)
    pndm.set_timesteps(200)
    for t in pndm.timesteps[:n_pos]:
        x = pndm.step(guided(on_dev(orc.surfpos_forward, sds[0], rep(x), t.reshape(-1), cl)), t, x)
    if not use_cf:
        x = x.repeat(1, 2, 1)
        S *= 2
    ddpm.set_timesteps(1000)
    for t in ddpm.timesteps[-n_ddpm:]:
        z = randn_tensor((B, S, 6), generator=gen) if int(t) > 0 else None
        x = ddpm.step(guided(on_dev(orc.surfpos_forward, sds[0], rep(x), t.reshape(-1), cl)), t, x, noise=z)
    surfPos, surfMask = dedup_surfaces(x, thr)
    surfZ = randn_tensor((B, S, 48), generator=gen)
    pndm.set_timesteps(200)
    for t in pndm.timesteps[:n_z]:
        surfZ = pndm.step(guided(on_dev(orc.surfz_forward, sds[1], rep(surfZ), t.reshape(-1), rep(surfPos), rep(surfMask), cl)),
                          t, surfZ)
    edgePos = randn_tensor((B, S, E, 6), generator=gen)
    pndm.set_timesteps(200)
    for t in pndm.timesteps[:n_pos]:
        e = on_dev(orc.edgepos_forward, sds[2], rep(edgePos), t.reshape(-1), rep(surfPos), rep(surfZ), rep(surfMask), cl)
        edgePos = pndm.step(guided(e), t, edgePos)
    ddpm.set_timesteps(1000)
    for t in ddpm.timesteps[-n_ddpm:]:
        z = randn_tensor((B, S, E, 6), generator=gen) if int(t) > 0 else None
        e = on_dev(orc.edgepos_forward, sds[2], rep(edgePos), t.reshape(-1), rep(surfPos), rep(surfZ), rep(surfMask), cl)
        edgePos = ddpm.step(guided(e), t, edgePos, noise=z)
    edgeM = dedup_edges(edgePos, surfMask, thr)
    edgeZV = randn_tensor((B, S, E, 18), generator=gen)
    pndm.set_timesteps(200)
    for t in pndm.timesteps[:n_z]:
        e = on_dev(orc.edgez_forward, sds[3], rep(edgeZV), t.reshape(-1), rep(edgePos), rep(surfPos), rep(surfZ), rep(edgeM), cl)
        edgeZV = pndm.step(guided(e), t, edgeZV)
    edgeZV = edgeZV.masked_fill(edgeM.unsqueeze(-1), 0.0)
    return dict(surfPos=surfPos, surfMask=surfMask, surfZ=surfZ, edgePos=edgePos, edgeM=edgeM, edgeZV=edgeZV)


@pytest.mark.parametrize("use_cf,varlen", [(False, False), (True, True),       # (the mixed pairs repeat the same code paths: BG_RUN_SLOW=1)
                                           pytest.param(True, False, marks=pytest.mark.slow), pytest.param(False, True, marks=pytest.mark.slow)])
def test_cascade_matches_oracle_cascade(use_cf, varlen):
    """varlen=False: dense execution, every position of every latent compared (the reference computes padded positions
    too); varlen=True (the product default): only valid tokens run through the nets -- masks must still be identical and
    the latents are compared where the reference's pipeline reads them (valid faces / edges; sample.py:284, 307-314)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import brepgen_amd as bga
    from brepgen_amd.sampling import CascadeSampler
    from oracle import denoisers as orc
    names = ["SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet"]
    sds = [orc.seeded_state_dict(n, 50 + i, use_cf) for i, n in enumerate(names)]
    nets = []
    for n, sd in zip(names, sds):
        m = getattr(bga, n)(use_cf)
        m.load_state_dict(sd, strict=True)
        m.varlen = varlen
        nets.append(m.cuda().eval())
    B, S, E = 2, 4, 3
    n_pos, n_ddpm, n_z = 14, 6, 14          # 12 PRK + 2 PLMS evaluations, 6 ancestral steps (the last has t = 0)
    kw = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001,
              beta_end=0.02)
    sampler = CascadeSampler(*nets, bga.PNDMScheduler(**kw), bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **kw),
                             use_cf=use_cf, class_id=6, guidance=0.6, autocast=False, noise_mode="reference")
    with torch.no_grad():
        got = sampler.sample(B, S, E, generator=torch.Generator().manual_seed(11), pndm_pos_steps=n_pos,
                             ddpm_pos_steps=n_ddpm, pndm_z_steps=n_z)
        want = _oracle_cascade(sds, B, S, E, torch.Generator().manual_seed(11), use_cf, 6, 0.6, n_pos, n_ddpm, n_z)
    assert set(got) == set(want)
    for k in ("surfMask", "edgeM"):
        assert torch.equal(got[k].cpu(), want[k]), k
    valid_face = ~want["surfMask"]
    for k in ("surfPos", "surfZ", "edgePos", "edgeZV"):
        assert got[k].shape == want[k].shape, k
        diff = (got[k].cpu() - want[k]).abs()
        if varlen and k in ("surfZ", "edgePos"):
            diff = diff[valid_face]                       # padded faces: never read downstream, not computed here
        d = float(diff.max())
        assert np.isfinite(d) and d < 5e-4, (k, d)     # fp32 path: per-step 1e-5 compounded over ~50 steps


def test_device_dedup_is_bit_identical_to_the_numpy_loops():
    """bg_dedup_surfaces / bg_dedup_edges vs the reference-order numpy code, incl. near-threshold and swapped boxes."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from brepgen_amd.sampling import dedup_edges, dedup_surfaces
    from oracle.dedup import dedup_edges_host, dedup_surfaces_host
    g = torch.Generator().manual_seed(4)
    for B, S, E in [(5, 60, 30), (3, 100, 40), (2, 7, 3), (4, 64, 64)]:
        base = torch.randn(B, 6, 6, generator=g).clamp(-3, 3)                     # 6 prototypes per sample
        idx = torch.randint(0, 6, (B, S), generator=g)
        pos = torch.gather(base, 1, idx.unsqueeze(-1).expand(B, S, 6)).clone()
        jitter = (torch.rand(B, S, 6, generator=g) - 0.5) * 0.17                  # straddles the 0.08 threshold
        pos = pos + jitter
        swap = torch.rand(B, S, generator=g) < 0.3                                # corner-swapped duplicates
        pos[swap] = torch.cat([pos[swap][:, 3:], pos[swap][:, :3]], dim=1)
        hp, hm = dedup_surfaces_host(pos, 0.08)
        dp, dm = dedup_surfaces(pos.cuda(), 0.08)
        assert torch.equal(dm.cpu(), hm) and torch.equal(dp.cpu(), hp), (B, S)
        ebase = torch.randn(B, S, 5, 6, generator=g).clamp(-3, 3)
        eidx = torch.randint(0, 5, (B, S, E), generator=g)
        ep = torch.gather(ebase, 2, eidx.unsqueeze(-1).expand(B, S, E, 6)) + (torch.rand(B, S, E, 6, generator=g) - 0.5) * 0.17
        he = dedup_edges_host(ep, hm, 0.08)
        de = dedup_edges(ep.cuda(), dm, 0.08)
        assert torch.equal(de.cpu(), he), (B, S, E)
        # a mask that is not left-aligned: the reference writes rows by position among the valid faces
        odd = torch.zeros(B, S, dtype=torch.bool)
        odd[:, ::3] = True
        assert torch.equal(dedup_edges(ep.cuda(), odd.cuda(), 0.08).cpu(), dedup_edges_host(ep, odd, 0.08))


def test_pipeline_driver_end_to_end(tmp_path):
    """brepgen_amd.pipeline: eval_config.yaml + .pt files on disk -> cascade + VAE decode -> the arrays of sample.py:286-299.
    Weights are seeded reference-keyed state dicts written with torch.save (the VAE files hold encoder + decoder)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import yaml
    from brepgen_amd import pipeline
    from oracle import denoisers as orc
    from oracle import vae as ov
    names = {"surfpos_weight": "SurfPosNet", "surfz_weight": "SurfZNet", "edgepos_weight": "EdgePosNet", "edgez_weight": "EdgeZNet"}
    args = {"batch_size": 2, "z_threshold": 0.2, "bbox_threshold": 0.08, "num_surfaces": 3, "num_edges": 3, "use_cf": True,
            "class_label": "table", "save_folder": str(tmp_path / "out")}
    for i, (k, net) in enumerate(names.items()):
        args[k] = f"{k}.pt"
        torch.save(orc.seeded_state_dict(net, 90 + i, True), tmp_path / args[k])
    for k, dec, enc in (("surfvae_weight", ov.surf_decoder_spec(), ov.surf_encoder_spec()),
                        ("edgevae_weight", ov.edge_decoder_spec(), ov.edge_encoder_spec())):
        full = dict(ov.seeded_state_dict(dec, 5))
        full.update(ov.seeded_state_dict(enc, 6))
        args[k] = f"{k}.pt"
        torch.save(full, tmp_path / args[k])
    cfg = tmp_path / "eval_config.yaml"
    cfg.write_text(yaml.safe_dump({"furniture": args}))
    eval_args = pipeline.load_eval_args(str(cfg), "furniture")
    sampler, surf_vae, edge_vae = pipeline.build(eval_args, "cuda", None, torch.float16, str(tmp_path))
    assert sampler.use_cf and sampler.class_id == 10
    out = pipeline.sample_batch(sampler, surf_vae, edge_vae, eval_args, torch.Generator().manual_seed(4),
                                pndm_pos_steps=13, ddpm_pos_steps=3, pndm_z_steps=13)
    B, S, E = 2, 3, 3                                   # use_cf: no late doubling of the faces (sample.py:140-142)
    assert out["surfPos"].shape == (B, S, 6) and out["surf_ncs"].shape == (B, S, 32, 32, 3)
    assert out["edge_pos"].shape == (B, S, E, 6) and out["edge_ncs"].shape == (B, S, E, 32, 3)
    assert out["edge_z"].shape == (B, S, E, 12) and out["edgeV"].shape == (B, S, E, 6)
    assert out["surfMask"].dtype == np.bool_ and out["edge_mask"].shape == (B, S, E)
    assert all(np.isfinite(v).all() for k, v in out.items() if v.dtype != np.bool_)


def test_chamfer_offset_fit_matches_the_oracle_loop():
    """bg_chamfer_offset_fit (one launch) vs oracle/joint_opt.py (itself pinned to torch AdamW + autograd on the CPU):
    the 200-iteration surface-offset fit of utils.py:746-772 on 32x32 point grids with ragged edge sets."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from brepgen_amd import postprocess
    from oracle import joint_opt as jo
    g = np.random.default_rng(5)
    F = 4
    surf = (g.normal(size=(F, 32, 32, 3)) * 0.5).astype(np.float32)
    edges = [(g.normal(size=(int(g.integers(2, 7)), 32, 3)) * 0.4 + g.normal(size=(1, 1, 3)) * 0.2).astype(np.float32)
             for _ in range(F)]
    edges[2] = edges[2][:1]                                  # a face with a single edge
    want_surf, want_off, losses = jo.optimize_surface_offsets(surf.reshape(F, -1, 3), [e.reshape(-1, 3) for e in edges])
    got_surf, got_off, got_loss = postprocess.optimize_surface_offsets(torch.from_numpy(surf).cuda(),
                                                                       [torch.from_numpy(e).cuda() for e in edges])
    assert got_surf.shape == (F, 32, 32, 3)
    assert float(np.abs(got_off.cpu().numpy() - want_off).max()) < 1e-4, (got_off.cpu().numpy(), want_off)
    assert float(np.abs(got_surf.cpu().numpy().reshape(F, -1, 3) - want_surf).max()) < 1e-4
    assert abs(float(got_loss.sum()) / F - losses[-1]) < 1e-3 * max(1.0, losses[-1])
    assert float(np.abs(want_off).max()) > 1e-2              # the fit actually moved the surfaces (~200 * lr)


@pytest.mark.parametrize("use_cf,dt", [(False, False), (True, torch.bfloat16),
                                       pytest.param(True, False, marks=pytest.mark.slow), pytest.param(False, torch.bfloat16, marks=pytest.mark.slow)])
def test_graph_replayed_cascade_is_bit_identical(use_cf, dt):
    """graphs=True: every stage replays one captured hipGraph per step (static latent / timestep buffers, conditioning
    captured by address); graphs=False launches kernel by kernel.  Same kernels in the same order -> the same bits, in
    fp32 and bf16, guided and unguided, variable-length execution and the device noise generator included."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import brepgen_amd as bga
    from brepgen_amd.sampling import CascadeSampler
    from oracle import denoisers as orc
    names = ["SurfPosNet", "SurfZNet", "EdgePosNet", "EdgeZNet"]
    nets = []
    for i, n in enumerate(names):
        m = getattr(bga, n)(use_cf)
        m.load_state_dict(orc.seeded_state_dict(n, 70 + i, use_cf), strict=True)
        nets.append(m.cuda().eval())
    kw = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon", beta_start=0.0001,
              beta_end=0.02)
    outs = []
    for graphs in (False, True):
        sampler = CascadeSampler(*nets, bga.PNDMScheduler(**kw), bga.DDPMScheduler(clip_sample=True, clip_sample_range=3, **kw),
                                 use_cf=use_cf, class_id=3, guidance=0.6, autocast=dt, graphs=graphs)
        with torch.no_grad():
            outs.append(sampler.sample(3, 5, 4, generator=torch.Generator().manual_seed(21), pndm_pos_steps=14,
                                       ddpm_pos_steps=6, pndm_z_steps=14))
    torch.cuda.synchronize()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
        assert outs[0][k].dtype == torch.bool or torch.isfinite(outs[0][k]).all(), k
