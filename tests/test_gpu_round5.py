"""-m gpu: round-5 additions -- the fused tail of the output MLP (csrc/out_tail.hip) against plain torch fp32 math, the final
LayerNorm folded into fc_out.0, the precomputed time-embedding table, and oracle-backed checks that drive the round-4 kernels
(fused QKV + attention, pipelined split-residual GEMM) DIRECTLY on exactly representable operands, so that their oracle leg is
tighter than 16-bit noise.  Measured numbers -> gpurun_out/parity_r05.json."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

F32, F16, BF16 = torch.float32, torch.float16, torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import parity_cases
    return parity_cases


def _record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_r05.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    d[key] = value
    with open(path, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


# ---- out_tail.hip: W3 . SiLU(LayerNorm(t0)) + b3 ---------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("n_out", [6, 18, 48])
@pytest.mark.parametrize("rows", [1, 17, 1000, 4099])
def test_ln_silu_out_vs_torch_fp32(pc, dt, n_out, rows):
    import hip_ops as ops
    g = torch.Generator().manual_seed(rows * 100 + n_out)
    t0 = (torch.randn(rows, 768, generator=g) * 1.7 + 0.3).to(dt)
    gamma, beta = 1 + 0.2 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    w3 = (torch.randn(64, 768, generator=g) * 0.05).to(dt)
    b3 = torch.randn(64, generator=g)
    got = ops.ln_silu_out(t0.cuda(), gamma.cuda(), beta.cuda(), w3.cuda(), b3.cuda(), n_out).cpu()
    # the kernel's arithmetic in plain torch: fp32 LayerNorm of the 16-bit rows, SiLU, rounded to the operand dtype, fp32 product
    x = t0.double()
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    h = (x - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    h = (h * torch.sigmoid(h)).float().to(dt)
    want = (h.double() @ w3[:n_out].double().T + b3[:n_out].double()).float()
    err = float((got - want).abs().max())
    # (a SiLU value within an ulp of a 16-bit rounding boundary may round the other way: one flip moves an output by 2^-9 * |h w|)
    assert torch.isfinite(got).all() and err < (4e-3 if dt == BF16 else 6e-4) * max(1.0, float(want.abs().max())), err


@pytest.mark.parametrize("net", ["SurfPosNet", "SurfZNet", "EdgeZNet"])
def test_fused_output_path_vs_unfused_and_oracle(pc, net):
    """net.fuse_output (final LayerNorm folded into fc_out.0 + the one-launch tail) against the four-launch path it replaces and
    against the fp32 oracle: both 16-bit paths sit at the same distance from the truth."""
    S, E = (8, 12) if net == "EdgeZNet" else (60, 1)
    m, sd = pc.build_net(net, 11, False, BF16)
    args = pc.synth_inputs(net, 3, S, E, False)
    cu = [a.cuda() if torch.is_tensor(a) else a for a in args]
    with torch.no_grad():
        want = pc.orc.FORWARD[net](sd, *args)
        fused = m(*cu).cpu()
        m.fuse_output = False
        plain = m(*cu).cpu()
    mask = args[3] if net == "SurfZNet" else (args[5] if net == "EdgeZNet" else None)
    valid = ~mask if mask is not None else torch.ones(want.shape[:-1], dtype=torch.bool)
    ef, ep = float((fused - want)[valid].abs().max()), float((plain - want)[valid].abs().max())
    mf, mp = float((fused - want)[valid].abs().mean()), float((plain - want)[valid].abs().mean())
    _record("fused_output_vs_unfused_" + net, {"fused_max": ef, "unfused_max": ep, "fused_mean": mf, "unfused_mean": mp})
    assert not torch.equal(fused, plain)
    assert ef < 3e-2 and ep < 3e-2 and mf < 1.25 * mp + 1e-4, (ef, ep, mf, mp)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_time_table_equals_per_call_time_mlp(pc, dt):
    """The precomputed time-embedding table against the per-call MLP (time_table_steps = 0): same eps up to the last bits of two
    GEMM orders (M = 1 GEMV vs M = 1000 MFMA rows); per-sample timesteps look the same rows up as a shared one."""
    m, sd = pc.build_net("SurfPosNet", 5, False, dt)
    x = torch.randn(4, 30, 6, generator=torch.Generator().manual_seed(3)).cuda()
    t1 = torch.tensor([437]).cuda()
    with torch.no_grad():
        a = m(x, t1, None)
        b = m(x, t1.repeat(4), None)
        m.time_table_steps = 0
        c = m(x, t1, None)
    assert torch.equal(a, b)
    d = float((a - c).abs().max())
    assert d < (5e-6 if dt == F32 else 2e-2), d      # (fp32: the two GEMM orders differ in the last bits; |eps| ~ 1-2)
    # a timestep outside the table fails loudly (NaN), it is never clamped
    m.time_table_steps = 1000
    with torch.no_grad():
        bad = m(x, torch.tensor([1000]).cuda(), None)
    assert torch.isnan(bad).all()


# ---- qkv_attn.hip: the XCD-pinned tile walk -----------------------------------------------------------------------------------
@pytest.fixture
def tune():
    from brepgen_amd import _lib
    lib = _lib.load()
    yield lib.bg_tune_set
    for k in (12, 13, 14):
        lib.bg_tune_set(k, 0)


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,N", [(512, 60), (300, 48), (37, 60), (130, 30), (8, 60), (67, 64)])
def test_qkv_attn_pinned_walk_is_bit_identical(pc, tune, dt, B, N):
    """bg_tune key 14: 1 = plain walk, 2 = XCD-pinned head halves wherever the grid allows (a different ORDER of the same tiles)."""
    import hip_ops as ops
    g = torch.Generator().manual_seed(B * 7 + N)
    M = B * N
    x = torch.randn(M, 768, generator=g) * 2
    grp = x.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().cuda()
    w = (torch.randn(2304, 768, generator=g) * 0.04).to(dt).cuda()
    b = torch.randn(2304, generator=g).cuda()
    cs = w.float().sum(1).contiguous()
    a = x.to(dt).cuda()
    outs = []
    for walk in (1, 2, 0):
        tune(14, walk)
        outs.append(ops.qkv_attention(a, w, b, cs, stats, B, N))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("ns", [1, 2])
def test_pinned_walk_inside_the_denoisers(pc, tune, ns):
    """SurfPosNet (equal lengths) and SurfZNet (ragged, slot-packed PAIR variant, device-side tile count) at a size that takes the
    pinned walk by default: eps bit-identical to the plain walk."""
    for net, B in (("SurfPosNet", 256), ("SurfZNet", 300)):
        m, _ = pc.build_net(net, 4, False, BF16, varlen=True)
        m.n_split = ns
        args = pc.synth_inputs(net, B, 60, 1, False)
        cu = [a.cuda() if torch.is_tensor(a) else a for a in args]
        res = []
        with torch.no_grad():
            for walk in (1, 2, 0):
                tune(14, walk)
                res.append(m(*cu))
        torch.cuda.synchronize()
        assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]), net


# ---- oracle legs tighter than 16-bit noise for the round-4 kernels ---------------------------------------------
# The round-4 tests prove these kernels EQUAL to the launches they replace; the chain to the oracle ran only through whole-network
# goldens at 16-bit tolerance.  Here each is driven directly and checked against fp64 torch math of its own definition: one 16-bit
# rounding of the q|k|v image, fp32-accumulation noise on the residual stream, an exactly representable softmax.
def _fold_case(M, dt, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, 768, generator=g) * 2).to(dt)             # the raw residual rows the GEMM reads (hi plane)
    xf = x.double()
    grp = xf.reshape(M, 12, 64)
    stats = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2).contiguous().float()
    w = torch.randn(2304, 768, generator=g) * 0.04
    w[:768] *= 0.125
    w = w.to(dt)
    b = torch.randn(2304, generator=g)
    return x, stats, w, b


def _qkv_fp64(x, w, b):
    """q|k|v of the LayerNorm-fold GEMM in fp64: rstd (x W'^T) - mean rstd colsum + b' (bg_common.h: ln_fold_coeffs / ln_fold_apply)."""
    xf, wf = x.double(), w.double()
    mean = xf.mean(-1, keepdim=True)
    var = (xf * xf).mean(-1, keepdim=True) - mean * mean
    rstd = 1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)
    return rstd * (xf @ wf.T) - mean * rstd * wf.sum(1)[None, :] + b.double()[None, :]


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("B,N", [(64, 60), (300, 30), (9, 64)])
def test_fused_qkv_image_is_one_rounding_from_fp64(pc, dt, B, N):
    """The q|k|v image of bg_qkv_attn_fwd against the fp64 value of its definition: within ONE rounding of the operand dtype (half an
    ulp = 2^-8 / 2^-11 relative, plus the fp32 accumulation noise of a 768-term product) -- not the 1e-2 of a whole network."""
    import hip_ops as ops
    x, stats, w, b = _fold_case(B * N, dt, B + N)
    cs = w.float().sum(1)
    out, img = ops.qkv_attention(x.cuda(), w.cuda(), b.cuda(), cs.cuda(), stats.cuda(), B, N, want_qkv=True)
    want = _qkv_fp64(x, w, b)
    err = (img.cpu().double() - want).abs()
    half_ulp = 2.0 ** (-8 if dt == BF16 else -11)                # relative half-spacing of 8 / 11 significant bits
    bound = half_ulp * want.abs() * 1.001 + 3e-5                  # (fp16 subnormals: values below 6e-5 carry an absolute step)
    assert bool((err <= bound).all()), float((err - bound).max())
    # and the attention, from the kernel's OWN q|k|v image in fp64 (q carries the 1/8): only P's and the output's roundings remain
    q, k, v = (img.cpu().double().reshape(B, N, 3, 12, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    p = torch.softmax(q @ k.transpose(-1, -2), -1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B * N, 768)
    ea = float((out.cpu().double() - ref).abs().max())
    _record(f"fused_qkv_attention_vs_fp64_{str(dt)[6:]}_B{B}_N{N}", {"image_max_err_over_bound": float((err / bound).max()), "attn_max_abs": ea})
    assert ea < (2e-2 if dt == BF16 else 3e-3) * max(1.0, float(ref.abs().max())), ea


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("N", [32, 64])
def test_fused_attention_with_exact_softmax(pc, dt, N):
    """q = 0 (zero W'_q rows and bias): every score is 0, p = 1 / N exactly (N a power of two), the output is the mean of v over the
    sample's keys -- P carries no rounding, so the launch must match fp64 to the output's own rounding."""
    import hip_ops as ops
    B = 40
    x, stats, w, b = _fold_case(B * N, dt, 5 * N)
    w[:768] = 0
    b[:768] = 0
    cs = w.float().sum(1)
    out, img = ops.qkv_attention(x.cuda(), w.cuda(), b.cuda(), cs.cuda(), stats.cuda(), B, N, want_qkv=True)
    v = img.cpu().double().reshape(B, N, 3, 768)[:, :, 2]
    ref = v.mean(1, keepdim=True).expand(B, N, 768).reshape(B * N, 768)
    err = (out.cpu().double() - ref).abs()
    half_ulp = 2.0 ** (-8 if dt == BF16 else -11)                # relative half-spacing of 8 / 11 significant bits
    assert float(img[:, :768].float().abs().max()) == 0.0
    assert bool((err <= half_ulp * ref.abs() * 1.01 + 2e-5).all()), float(err.max())


@pytest.mark.parametrize("dt", [BF16, F16])
@pytest.mark.parametrize("K", [768, 1024])
def test_split_pipe_gemm_vs_fp64(pc, dt, K):
    """The pipelined split-residual GEMM (out-proj / FFN2 form; 10 000 rows -> the gemm_split.hip kernel, ragged last panel) against
    fp64: hi + lo reproduces v = a W^T + b + (res_hi + res_lo) to fp32-accumulation noise plus the lo plane's rounding (2^-16 / 2^-22
    relative), hi is the correctly rounded v wherever v is not within that noise of a rounding boundary, and the row statistics are
    the sums of v."""
    import hip_ops as ops
    M = 10000
    g = torch.Generator().manual_seed(K)
    rn = lambda *s: torch.randn(*s, generator=g)
    xr = rn(M, 768) * 2
    hi = xr.to(dt)
    lo = (xr - hi.float()).to(dt)
    a, w, b = (rn(M, K) * 0.5).to(dt), (rn(768, K) * 0.04).to(dt), rn(768)
    r = ops.linear_ex(a.cuda(), w.cuda(), b.cuda(), split_out=True, res=(hi.cuda(), lo.cuda()), want_stats=True)
    v = a.double() @ w.double().T + b.double() + hi.double() + lo.double()
    got = r["out"].cpu().double() + r["lo"].cpu().double()
    lo_ulp = 2.0 ** (-16 if dt == BF16 else -22)                  # half-spacing of the lo plane relative to v
    err = (got - v).abs()
    assert bool((err <= lo_ulp * v.abs() + 2e-5).all()), float(err.max())
    hi_want = v.float().to(dt)
    frac = float((r["out"].cpu() != hi_want).float().mean())
    assert frac < 2e-3, frac                                      # (only values within fp32 noise of a rounding boundary)
    grp = v.reshape(M, 12, 64)
    st = torch.stack([grp.sum(-1), (grp * grp).sum(-1)], -1).permute(1, 0, 2)
    es = float(((r["stats"].cpu().double() - st).abs() / (1 + st.abs())).max())
    _record(f"split_pipe_vs_fp64_{str(dt)[6:]}_K{K}", {"hi_plus_lo_max_err": float(err.max()), "hi_mismatch_fraction": frac, "stats_rel": es})
    assert es < 2e-5, es


def test_exact_row_plan_skips_nothing_but_launches(pc):
    """rows_plan (the exact device-side row count, known to the module from its asynchronous count of the mask): the launcher may then
    run a GEMM on the 256 x 256 kernel alone instead of that kernel + a 128 x 128 tail that finds nothing to do.  eps must not change,
    with one and two sample groups, and the plan must equal what the device counted."""
    from brepgen_amd import _lib
    for ns in (1, 2):
        m, _ = pc.build_net("SurfZNet", 6, False, BF16, varlen=True)
        m.n_split = ns
        args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
        with torch.no_grad():
            first = m(*args).clone()                              # first sight of the mask: no numbers
            second = m(*args).clone()                             # count started
            torch.cuda.synchronize()
            third = m(*args).clone()                              # count arrived: planned launches
            with _lib.profile() as prof:
                m(*args)
        assert torch.equal(first, second) and torch.equal(first, third)
        hc = m._hint_cache
        pk = (ns, (True,) * ns)                                    # (slot-packed in every sample group: the library's own predicate per group)
        assert hc["counts"] is not None and pk in hc["plans"]
        counts = (~args[3]).sum(1).tolist()
        assert hc["counts"] == counts
        want = [m._slot_rows(counts[lo:hi]) for lo, hi in m._group_ranges(512, ns)]
        assert list(hc["plans"][pk]) == [float(v) for v in want]
        # the residual-stream GEMMs (24 per evaluation and group): where the rule gives every panel of the planned count to the
        # 256 x 256 kernel it runs alone, where it gives it none the pipelined 128 x 128 kernel does -- never both
        lib = _lib.load()
        names = {r["kernel"]: r["launches"] for r in prof.rows}
        all256 = [lib.bg_gemm_p256_rows(int(v), 768, 1, int(ns > 1)) >= v for v in want]
        none256 = [lib.bg_gemm_p256_rows(int(v), 768, 1, int(ns > 1)) == 0 for v in want]
        assert all(a or n for a, n in zip(all256, none256)), (want, "the test shape should not need both kernels")
        assert names.get("gemm16_p256_kernel(256x256, split-residual launches)", 0) == 24 * sum(all256), names
        assert names.get("gemm16_split_pipe_kernel(128x128)", 0) == 24 * sum(none256), names
