"""Parity measurements: HIP path (through the C ABI) vs the CPU oracle / plain torch fp32 restatements.

Every function returns a dict of error metrics; ``tests/test_gpu_*.py`` assert on them and
``tests/gpu_check.py`` prints them all (one gpurun call gives the whole picture).  Needs a GPU.
"""
import json
import math
import os

import numpy as np
import torch

import brepgen_amd as bga
import hip_ops as ops
from oracle import denoisers as orc
from oracle.schedulers import OracleDDPM, OraclePNDM

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def _err(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    d = (got - want).abs()
    return {"max_abs": float(d.max()), "mean_abs": float(d.mean()), "ref_absmax": float(want.abs().max()),
            "finite": bool(torch.isfinite(got).all())}


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------------------------------------------
def layernorm_case(M, out_dtype, silu, seed=0):
    g = gen(seed)
    x = torch.randn(M, 768, generator=g) * 2 + 0.3
    w = 1 + 0.1 * torch.randn(768, generator=g)
    b = 0.1 * torch.randn(768, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (768,), w.double(), b.double(), 1e-5)
    if silu:
        want = torch.nn.functional.silu(want)
    got = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), out_dtype=out_dtype, silu=silu)
    return _err(got.float(), want)


def sincos_case():
    t = torch.tensor([0, 1, 10, 249, 255, 500, 980, 995, 999])
    return _err(ops.sincos_embed(t.to(DEV)), orc.sincos_embedding(t))


def gemm_case(M, N, K, dtype, n_valid=None, bias=True, act=0, add_mode=None, out_dtype=torch.float32, seed=0):
    """add_mode: None | 'resid' (in place, add_div=1) | int d (broadcast rows m//d)."""
    g = gen(seed)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    bvec = torch.randn(N, generator=g) if bias else None
    nv = N if n_valid is None else n_valid
    a_d, w_d = a.to(dtype), w.to(dtype)
    ref = a_d.double() @ w_d.double().t()
    if bias:
        ref = ref + bvec.double()
    if act:
        ref = ref.clamp_min(0)
    ref = ref[:, :nv]
    add = None
    out = None
    add_div = 1
    if add_mode == "resid":
        add = torch.randn(M, nv, generator=g)
        ref = ref + add.double()
        out = add.clone().to(DEV)
        add_t = out
    elif isinstance(add_mode, int):
        rows = (M + add_mode - 1) // add_mode
        add = torch.randn(rows, nv, generator=g)
        ref = ref + add.double().repeat_interleave(add_mode, 0)[:M]
        add_t = add.to(DEV)
        add_div = add_mode
    else:
        add_t = None
    got = ops.linear(a_d.to(DEV), w_d.to(DEV), bvec.to(DEV) if bias else None, out_dtype=out_dtype, act=act,
                     add=add_t, add_div=add_div, n_valid=nv, out=out)
    if out_dtype == torch.bfloat16:
        ref = ref.to(torch.bfloat16).double() if False else ref
    return _err(got.float(), ref)


def _attn_ref(qkv, mask, B, N):
    """fp64 reference on the (already rounded) operands; q is pre-scaled."""
    q, k, v = qkv.double().reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s.masked_fill(mask.reshape(B, 1, 1, N), float("-inf"))
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(B * N, 768)


def attn_case(B, N, dtype, mask_kind="ragged", seed=0, scale=1.0):
    g = gen(seed)
    qkv = torch.randn(B * N, 2304, generator=g) * scale
    qkv[:, :768] *= 0.125 * 2.0          # pre-scaled q, with some spread in the logits
    mask = None
    if mask_kind == "ragged":
        mask = torch.ones(B, N, dtype=torch.bool)
        for b in range(B):
            mask[b, : int(torch.randint(1, N + 1, (1,), generator=g))] = False
    elif mask_kind == "random":
        mask = torch.rand(B, N, generator=g) < 0.5
        mask[:, 0] = False
    qd = qkv.to(dtype)
    want = _attn_ref(qd, mask, B, N)
    got = ops.attention(qd.to(DEV), mask.to(DEV) if mask is not None else None, B, N)
    return _err(got.float(), want)


# ---------------------------------------------------------------------------------------------------
def ddpm_case(t, shape=(4, 60, 48), guidance=None, clip=True, seed=0):
    g = gen(seed)
    x = torch.randn(*shape, generator=g) * 1.5
    B = shape[0]
    eps = torch.randn(*((2 * B,) + shape[1:]) if guidance else shape, generator=g)
    noise = torch.randn(*shape, generator=g)
    o = OracleDDPM(clip_sample=clip, clip_sample_range=3)
    o.set_timesteps(1000)
    s = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                          beta_start=0.0001, beta_end=0.02, clip_sample=clip, clip_sample_range=3)
    s.set_timesteps(1000)
    e_eff = eps if not guidance else eps[:B] * (1 + guidance) - eps[B:] * guidance
    want = o.step(e_eff, t, x, noise=noise)
    got = s.step(eps.to(DEV), torch.tensor(t), x.to(DEV), noise=noise.to(DEV), guidance=guidance).prev_sample
    return _err(got, want)


def pndm_case(n_steps=209, shape=(3, 30, 6), guidance=None, seed=0):
    """Drive both PNDM implementations with the same pseudo-model eps = f(x, t) and compare every step."""
    g = gen(seed)
    x0 = torch.randn(*shape, generator=g)
    B = shape[0]
    o = OraclePNDM()
    o.set_timesteps(200)
    s = bga.PNDMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                          beta_start=0.0001, beta_end=0.02)
    s.set_timesteps(200)
    assert s.timesteps.tolist() == o.timesteps.tolist()
    xo, xs = x0.clone(), x0.clone().to(DEV)
    worst = 0.0
    for i, t in enumerate(s.timesteps[:n_steps]):
        noise_like = torch.randn(*((2 * B,) + shape[1:]) if guidance else shape, generator=g)
        # pseudo network: depends on the current sample so errors would propagate
        def net(x):
            rep = x.repeat(2, *([1] * (x.dim() - 1))) if guidance else x
            return 0.5 * torch.tanh(rep) + 0.3 * noise_like.to(x.device)
        eo = net(xo)
        if guidance:
            eo = eo[:B] * (1 + guidance) - eo[B:] * guidance
        xo = o.step(eo, t, xo)
        xs = s.step(net(xs), t, xs, guidance=guidance).prev_sample
        worst = max(worst, float((xs.cpu() - xo).abs().max()))
    return {"max_abs": worst, "ref_absmax": float(xo.abs().max()), "finite": bool(torch.isfinite(xs).all()),
            "mean_abs": worst}


def embed_case(rows, k, out_dtype, lda=None, col0=0, seed=0):
    """Fused Linear(k) + LayerNorm + SiLU vs plain torch fp64 math."""
    g = gen(seed)
    lda = lda or k
    xfull = torch.randn(rows, lda, generator=g) * 1.5
    w0, b0 = torch.randn(768, k, generator=g) * 0.3, torch.randn(768, generator=g) * 0.1
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    xd = xfull.to(DEV)
    got = ops.embed_ln_silu(xd[:, col0:], k, w0.to(DEV), b0.to(DEV), gamma.to(DEV), beta.to(DEV), out_dtype)
    x = xfull[:, col0:col0 + k].double()
    h = torch.nn.functional.layer_norm(x @ w0.double().t() + b0.double(), (768,), gamma.double(), beta.double(), 1e-5)
    return _err(got.float(), torch.nn.functional.silu(h).float())


def _split(x, dt):
    hi = x.to(dt)
    return hi, (x - hi.float()).to(dt)


def gemm_split_case(M, dtype, K=768, N=768, with_res=True, seed=0):
    """Split-residual producer epilogue: hi/lo planes + per-64-column row statistics vs plain torch fp32 math."""
    g = gen(seed)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dtype)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    bias = torch.randn(N, generator=g)
    x = torch.randn(M, N, generator=g) * 3 + 0.7
    hi, lo = _split(x, dtype)
    kw = dict(res=(hi.to(DEV), lo.to(DEV))) if with_res else dict(add=x.to(DEV))
    r = ops.linear_ex(a.to(DEV), w.to(DEV), bias.to(DEV), split_out=True, want_stats=True, **kw)
    want = a.double() @ w.double().t() + bias.double() + ((hi.double() + lo.double()) if with_res else x.double())
    got = r["out"].double().cpu() + r["lo"].double().cpu()
    st = r["stats"].double().cpu().permute(1, 0, 2)             # part-major on the device
    grp = want.reshape(M, N // 64, 64)
    return {"hi": r["out"], "lo": r["lo"], "stats": r["stats"],
            "max_abs": float((got - want).abs().max()),
            "hi_is_rounding": bool((r["out"].cpu().float() - want.float().to(dtype).float()).abs().max() <= 2 * float(want.abs().max()) * 2 ** -8),
            "stats_sum_err": float((st[..., 0] - grp.sum(-1)).abs().max()),
            "stats_sq_rel": float(((st[..., 1] - (grp * grp).sum(-1)).abs() / (grp * grp).sum(-1)).max())}


def gemm_fold_case(M, N, dtype, act=0, seed=0):
    """LayerNorm-fold consumer epilogue vs (a) the same algebra in fp64 and (b) LayerNorm(x) @ W^T + b itself."""
    g = gen(seed)
    K = 768
    x = torch.randn(M, K, generator=g) * 2.5 + torch.randn(M, 1, generator=g)      # per-row mean up to ~1 sigma/2
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    W, b = torch.randn(N, K, generator=g) * 0.04, torch.randn(N, generator=g) * 0.1
    hi, _ = _split(x, dtype)
    grp = x.reshape(M, K // 64, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous()    # [12, M, 2]
    Wp = (W * gamma[None]).to(dtype)
    colsum = Wp.float().sum(1)
    c = b + W @ beta
    r = ops.linear_ex(hi.to(DEV), Wp.to(DEV), c.to(DEV), act=act, stats_in=stats.to(DEV), colsum=colsum.to(DEV))
    got = r["out"].double().cpu()
    mean = x.double().mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.double().var(-1, unbiased=False, keepdim=True) + 1e-5)
    alg = rstd * (hi.double() @ Wp.double().t()) - mean * rstd * colsum.double()[None] + c.double()[None]
    ln = ((x.double() - mean) * rstd * gamma.double() + beta.double()) @ W.double().t() + b.double()
    if act:
        alg, ln = alg.clamp(min=0), ln.clamp(min=0)
    scale = float(ln.abs().max())
    return {"out": r["out"], "vs_algebra": float((got - alg).abs().max()) / scale,
            "vs_layernorm": float((got - ln).abs().max()) / scale}


def layernorm_split_case(M, dtype, seed=0):
    g = gen(seed)
    x = torch.randn(M, 768, generator=g) * 2 + 0.3
    hi, lo = _split(x, dtype)
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    got = ops.layernorm_split(hi.to(DEV), lo.to(DEV), gamma.to(DEV), beta.to(DEV))
    want = torch.nn.functional.layer_norm(hi.float() + lo.float(), (768,), gamma, beta, 1e-5)
    return _err(got.float(), want)


# ---------------------------------------------------------------------------------------------------
NETS = {"SurfPosNet": bga.SurfPosNet, "SurfZNet": bga.SurfZNet, "EdgePosNet": bga.EdgePosNet, "EdgeZNet": bga.EdgeZNet}
MANIFEST = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))


def build_net(net, seed, use_cf, dtype, varlen=False, weights="seeded"):
    """varlen=False: dense execution (every position as the reference computes it) -- what the position-exact parity
    tests check; varlen=True: the product default (valid tokens only, 0 at padded positions)."""
    sd = orc.make_state_dict(weights, net, seed, use_cf)
    m = NETS[net](use_cf)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = dtype
    m.varlen = varlen
    return m, sd


def golden_case(name, dtype, varlen=False, fold=True, center=True, autocast_ref=False):
    """HIP denoiser vs the golden output written by the reference's own class (tests/golden/gen_golden.py).
    autocast_ref: also run oracle/ref_formulation.py (the reference's formulation: stock nn.TransformerEncoder) under
    torch.autocast('cuda', dtype) on the same inputs -- how sample.py:121 runs the reference -- and report its error."""
    meta = MANIFEST["cases"][name]
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    args = [torch.from_numpy(z[k]).to(DEV) if k in z.files else None for k in meta["args"]]
    m, sd = build_net(meta["net"], meta["weight_seed"], meta["use_cf"], dtype, varlen, meta.get("weights", "seeded"))
    m.fold_layernorm, m.center_stream = fold, center
    with torch.no_grad():
        got = m(*args)
    want = torch.from_numpy(z["out"])
    e = _err(got, want)
    e["got"] = got.cpu()
    mk = "surf_mask" if "surf_mask" in meta["args"] else ("mask" if "mask" in meta["args"] else None)
    vsel = ~torch.from_numpy(z[mk]) if mk is not None else torch.ones(want.shape[:-1], dtype=torch.bool)
    if meta["net"] == "EdgePosNet" and mk is not None:
        vsel = vsel.unsqueeze(-1).expand(want.shape[:-1])
    e["mean_abs_valid"] = float((got.cpu() - want)[vsel].abs().mean())
    if autocast_ref:
        from oracle import ref_formulation as rf
        ref = rf.build(meta["net"], sd, meta["use_cf"]).to(DEV)
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
            ra = ref(*args).float().cpu()
        e["autocast_max_abs_valid"] = float((ra - want)[vsel].abs().max())
        e["autocast_mean_abs_valid"] = float((ra - want)[vsel].abs().mean())
    mask_key = "surf_mask" if "surf_mask" in meta["args"] else ("mask" if "mask" in meta["args"] else None)
    if mask_key is not None and meta["net"] != "EdgePosNet":
        valid = ~torch.from_numpy(z[mask_key])
        e["max_abs_valid"] = float((got.cpu() - want)[valid].abs().max())
    elif mask_key is not None:
        valid = ~torch.from_numpy(z[mask_key])
        e["max_abs_valid"] = float((got.cpu() - want)[valid].abs().max())
    else:
        e["max_abs_valid"] = e["max_abs"]
    return e


def synth_inputs(net, B, S, E, use_cf, seed=1234):
    """Synthetic DeepCAD-shaped inputs per SURVEY.md section 8(d)."""
    g = gen(seed)
    R = lambda *s: torch.randn(*s, generator=g)
    t = torch.tensor([249])
    cl = None
    if use_cf:
        cl = torch.cat([torch.full((B // 2, 1), 6), torch.zeros(B - B // 2, 1, dtype=torch.long)]).long()
    smask = torch.ones(B, S, dtype=torch.bool)
    for b in range(B):
        smask[b, : int(torch.randint(min(8, S), S + 1, (1,), generator=g))] = False
    if net == "SurfPosNet":
        return [R(B, S, 6).clamp(-3, 3), t, cl]
    if net == "SurfZNet":
        return [R(B, S, 48), t, R(B, S, 6).clamp(-3, 3), smask, cl]
    if net == "EdgePosNet":
        return [R(B, S, E, 6).clamp(-3, 3), t, R(B, S, 6).clamp(-3, 3), R(B, S, 48), smask, cl]
    em = torch.rand(B, S, E, generator=g) < 0.4
    em[:, :, 0] = False
    em = em | smask.unsqueeze(-1)
    em[:, 0, 0] = False
    return [R(B, S, E, 18), t, R(B, S, E, 6).clamp(-3, 3), R(B, S, 6).clamp(-3, 3), R(B, S, 48), em, cl]


def oracle_case(net, B, S, E, dtype, use_cf=False, seed=7, varlen=False):
    """HIP denoiser vs the CPU oracle on seeded inputs at a size the oracle finishes in seconds."""
    m, sd = build_net(net, seed, use_cf, dtype, varlen)
    args = synth_inputs(net, B, S, E, use_cf)
    with torch.no_grad():
        want = orc.FORWARD[net](sd, *args)
        got = m(*[a.to(DEV) if torch.is_tensor(a) else a for a in args])
    e = _err(got, want)
    mask = args[3] if net == "SurfZNet" else (args[4] if net == "EdgePosNet" else (args[5] if net == "EdgeZNet" else None))
    if mask is not None:
        valid = ~mask if net != "EdgePosNet" else (~mask)[:, :, None].expand(B, S, E)
        e["max_abs_valid"] = float((got.cpu() - want)[valid].abs().max())
        e["padded_absmax"] = float(got.cpu()[~valid].abs().max()) if bool((~valid).any()) else 0.0
    else:
        e["max_abs_valid"] = e["max_abs"]
        e["padded_absmax"] = 0.0
    return e


_SD_DEV = {}


def oracle_on_device(fn, sd, *args):
    """The fp32 oracle forward `fn(sd, *args)` evaluated on the GPU (plain torch; gfx950 has no TF32) and returned on the CPU:
    the chain / cascade tests call the oracle hundreds of times on tiny batches, which costs ~0.2 s per call on the host."""
    key = id(sd)
    if key not in _SD_DEV:
        _SD_DEV[key] = ({k: v.to(DEV) for k, v in sd.items()}, sd)      # (keep sd alive: the cache is keyed on its id)
    mv = lambda a: a.to(DEV) if torch.is_tensor(a) else a
    with torch.no_grad():
        return fn(_SD_DEV[key][0], *[mv(a) for a in args]).cpu()


def ddpm_chain_case(dtype, steps=50, B=1, N=60, seed=3, last=None, oracle_dev=True):
    """BASELINE configs[0]: B=1 face-LDM, `steps` DDPM steps of SurfZNet with injected noise, HIP vs oracle.
    Both chains are fed the ORACLE's trajectory (per-step parity: same x_t in, compare x_{t-1} out)."""
    m, sd = build_net("SurfZNet", seed, False, dtype)
    g = gen(99)
    surfPos = torch.randn(B, N, 6, generator=g).clamp(-3, 3)
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[:, 40:] = True
    x = torch.randn(B, N, 48, generator=g)
    o = OracleDDPM(clip_sample=True, clip_sample_range=3)
    o.set_timesteps(steps)
    s = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                          beta_start=0.0001, beta_end=0.02, clip_sample=True, clip_sample_range=3)
    s.set_timesteps(steps)
    worst_eps = worst_x = 0.0
    sp_d, mk_d = surfPos.to(DEV), mask.to(DEV)
    with torch.no_grad():
        for t in (s.timesteps if last is None else s.timesteps[-last:]):
            noise = torch.randn(B, N, 48, generator=g)
            tt = t.reshape(-1)
            eo = oracle_on_device(orc.surfz_forward, sd, x, tt, surfPos, mask) if oracle_dev else orc.surfz_forward(sd, x, tt, surfPos, mask)
            xo = o.step(eo, t, x, noise=noise)
            eh = m(x.to(DEV), tt.to(DEV), sp_d, mk_d, None)
            xh = s.step(eh, t, x.to(DEV), noise=noise.to(DEV)).prev_sample
            worst_eps = max(worst_eps, float((eh.cpu() - eo)[~mask].abs().max()))
            worst_x = max(worst_x, float((xh.cpu() - xo)[~mask].abs().max()))
            x = xo
    return {"max_abs_eps": worst_eps, "max_abs_x": worst_x, "finite": bool(torch.isfinite(xh).all())}


# ---------------------------------------------------------------------------------------------------
SURF_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownEncoderBlock2D"] * 4,
                up_block_types=["UpDecoderBlock2D"] * 4, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                act_fn="silu", latent_channels=3, norm_num_groups=32, sample_size=512)          # sample.py:72-82
EDGE_CFG = dict(in_channels=3, out_channels=3, down_block_types=["DownBlock1D"] * 3, up_block_types=["UpBlock1D"] * 3,
                block_out_channels=[128, 256, 512], layers_per_block=2, act_fn="silu", latent_channels=3,
                norm_num_groups=32, sample_size=512)                                            # sample.py:86-97


def vae_case(kind, n, dtype, seed=0):
    """HIP VAE decoder vs the oracle restatement (oracle/vae.py) on seeded weights and latents."""
    from oracle import vae as ov
    g = gen(100 + seed)
    if kind == "surf":
        sd = ov.seeded_state_dict(ov.surf_decoder_spec(), 31 + seed)
        m = bga.AutoencoderKLFastDecode(**SURF_CFG)
        z = torch.randn(n, 3, 4, 4, generator=g)
        with torch.no_grad():
            want = ov.surf_decode(sd, z)
    elif kind == "surf_enc":
        sd = ov.seeded_state_dict(ov.surf_encoder_spec(), 51 + seed)
        m = bga.AutoencoderKLFastEncode(**SURF_CFG)
        z = torch.randn(n, 3, 32, 32, generator=g)
        with torch.no_grad():
            want = ov.surf_encode(sd, z)
    elif kind == "edge_enc":
        sd = ov.seeded_state_dict(ov.edge_encoder_spec(), 61 + seed)
        m = bga.AutoencoderKL1DFastEncode(**EDGE_CFG)
        z = torch.randn(n, 3, 32, generator=g)
        with torch.no_grad():
            want = ov.edge_encode(sd, z)
    else:
        sd = ov.seeded_state_dict(ov.edge_decoder_spec(), 41 + seed)
        m = bga.AutoencoderKL1DFastDecode(**EDGE_CFG)
        z = torch.randn(n, 3, 4, generator=g)
        with torch.no_grad():
            want = ov.edge_decode(sd, z)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.compute_dtype = dtype
    with torch.no_grad():
        got = m(z.to(DEV))
    assert got.shape == want.shape
    return _err(got, want)


def upsample1d_case(S=3, L=8, C=12):
    from oracle import vae as ov
    from brepgen_amd import _lib
    g = gen(5)
    x = torch.randn(S, C, L, generator=g)
    want = ov.upsample1d_cubic(x)                                  # [S, C, 2L]
    xc = x.permute(0, 2, 1).contiguous().to(DEV)
    y = torch.empty(S, 2 * L, C, device=DEV)
    _lib.check(_lib.load().bg_upsample1d_cubic(xc.data_ptr(), y.data_ptr(), S, L, C, _lib.stream()), "upsample")
    return _err(y.permute(0, 2, 1), want)


def downsample1d_case(S=3, L=16, C=12):
    from oracle import vae as ov
    from brepgen_amd import _lib
    g = gen(6)
    x = torch.randn(S, C, L, generator=g)
    want = ov.downsample1d_cubic(x)
    xc = x.permute(0, 2, 1).contiguous().to(DEV)
    y = torch.empty(S, L // 2, C, device=DEV)
    _lib.check(_lib.load().bg_downsample1d_cubic(xc.data_ptr(), y.data_ptr(), S, L, C, _lib.stream()), "downsample")
    return _err(y.permute(0, 2, 1), want)
