"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, the golden vectors written by the
reference's own classes, and fp64 restatements of the single ops.

Tolerances (measured round 1 on MI355X; the asserted bounds leave ~2-4x headroom):
  fp32 mode  : eps within 1e-5 of the reference classes (measured 2.5e-6 .. 4.4e-6)   [north_star: 1e-5 fp32]
  bf16 mode  : GEMM/attention operands carry 8 mantissa bits, so a 12-layer stack lands at ~1.3e-2 max-abs on
               eps (0.7 % of |eps|max ~ 1.9).  north_star's "1e-3 bf16" is met for the per-step sample update
               x_{t-1} at the sampling schedule (eps enters with a coefficient of ~0.007-0.02), not for eps itself;
               both are asserted below with the measured scale.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16


@pytest.fixture(scope="module")
def pc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import parity_cases
    return parity_cases


def test_native_library_is_loaded(pc):
    from brepgen_amd import _lib
    lib = _lib.load()
    assert lib.bg_abi_version() == _lib.ABI_VERSION
    maps = open("/proc/self/maps").read()
    assert "libbrepgen_hip.so" in maps


# ---- single kernels ---------------------------------------------------------------------------------
def test_sincos(pc):
    assert pc.sincos_case()["max_abs"] < 1e-4          # arguments reach 999 rad: 1 ulp of freq = 6e-5


@pytest.mark.parametrize("silu", [False, True])
def test_layernorm_fp32(pc, silu):
    assert pc.layernorm_case(61, F32, silu)["max_abs"] < 5e-6
    assert pc.layernorm_case(1, F32, silu)["max_abs"] < 5e-6


@pytest.mark.parametrize("silu", [False, True])
def test_layernorm_bf16(pc, silu):
    e = pc.layernorm_case(130, BF16, silu)
    assert e["max_abs"] < 2e-2 * max(1.0, e["ref_absmax"] / 4)    # one bf16 rounding of the output


@pytest.mark.parametrize("t", [999, 249, 1, 0])
def test_ddpm_step_bit_exact_vs_oracle(pc, t):
    assert pc.ddpm_case(t)["max_abs"] <= 1e-6
    assert pc.ddpm_case(t, shape=(3, 7, 6))["max_abs"] <= 1e-6      # non-multiple-of-4 tail
    assert pc.ddpm_case(t, guidance=0.6)["max_abs"] <= 2e-6
    assert pc.ddpm_case(t, clip=False)["max_abs"] <= 1e-6


def test_pndm_full_schedule(pc):
    e = pc.pndm_case()
    assert e["finite"] and e["max_abs"] <= 2e-6 * max(1.0, e["ref_absmax"])
    e = pc.pndm_case(n_steps=40, guidance=0.6)
    assert e["finite"] and e["max_abs"] <= 4e-6 * max(1.0, e["ref_absmax"])


@pytest.mark.parametrize("shape", [(1, 768, 768), (60, 768, 6), (130, 768, 48), (61, 768, 12), (77, 6, 768),
                                   (64, 2304, 768), (0 + 5, 18, 768)])
def test_gemm_fp32(pc, shape):
    assert pc.gemm_case(*shape, F32)["max_abs"] < 2e-5
    assert pc.gemm_case(*shape, F32, act=1, add_mode="resid")["max_abs"] < 2e-5


def test_gemm_fp32_broadcast_add(pc):
    assert pc.gemm_case(100, 768, 48, F32, add_mode=30)["max_abs"] < 2e-5


@pytest.mark.parametrize("shape", [(60, 768, 768), (128, 2304, 768), (257, 1024, 768), (1000, 768, 1024),
                                   (1, 768, 768), (4099, 2304, 768)])
def test_gemm_bf16_exact_products(pc, shape):
    # operands are rounded once on the host; the kernel's fp32 accumulation must match fp64 to fp32 noise
    assert pc.gemm_case(*shape, BF16)["max_abs"] < 2e-5


def test_gemm_bf16_epilogues(pc):
    assert pc.gemm_case(300, 768, 1024, BF16, add_mode="resid")["max_abs"] < 2e-5
    assert pc.gemm_case(300, 768, 768, BF16, add_mode=60)["max_abs"] < 2e-5
    assert pc.gemm_case(130, 768, 768, BF16, bias=False)["max_abs"] < 2e-5
    e = pc.gemm_case(300, 1024, 768, BF16, act=1, out_dtype=BF16)
    assert e["max_abs"] < 4e-3 * e["ref_absmax"] + 1e-6                 # bf16 output rounding only
    for nv in (6, 18, 48):
        assert pc.gemm_case(300, 64, 768, BF16, n_valid=nv)["max_abs"] < 2e-5


@pytest.mark.parametrize("N", [1, 17, 30, 32, 33, 60, 64, 65, 100, 128, 130, 257])
@pytest.mark.parametrize("mask", ["ragged", "random", None])
def test_attention_bf16(pc, N, mask):
    e = pc.attn_case(3, N, BF16, mask)
    assert e["finite"] and e["max_abs"] < 3e-2 and e["mean_abs"] < 3e-3   # P and O carry bf16 roundings


def test_attention_bf16_long(pc):
    e = pc.attn_case(1, 1800, BF16, "random")
    assert e["finite"] and e["max_abs"] < 3e-2
    e = pc.attn_case(1, 4000, BF16, "ragged")                              # ABC edge-net length
    assert e["finite"] and e["max_abs"] < 3e-2


def test_attention_online_softmax_rescale(pc):
    """Large logits spread across key tiles force the running-max rescale branch every tile."""
    e = pc.attn_case(2, 300, BF16, None, seed=5, scale=3.0)
    assert e["finite"] and e["max_abs"] < 0.12 and e["mean_abs"] < 8e-3


@pytest.mark.parametrize("N", [1, 17, 60, 130, 300])
def test_attention_fp32(pc, N):
    e = pc.attn_case(2, N, F32, "ragged")
    assert e["finite"] and e["max_abs"] < 1e-5


# ---- fused input-embed kernel: Linear(k) + LayerNorm + SiLU (csrc/embed.hip) --------------------------------
@pytest.mark.parametrize("k", [6, 12, 48])
@pytest.mark.parametrize("rows", [1, 33, 1000])
def test_embed_ln_silu_fp32(pc, k, rows):
    e = pc.embed_case(rows, k, F32)
    assert e["finite"] and e["max_abs"] < 2e-5


@pytest.mark.parametrize("dtype,tol", [(BF16, 2e-2), (F16, 3e-3)])
def test_embed_ln_silu_16bit_and_strided_input(pc, dtype, tol):
    assert pc.embed_case(257, 48, dtype)["max_abs"] < tol
    assert pc.embed_case(100, 6, dtype, lda=18, col0=12)["max_abs"] < tol      # vertp_fc reads x[:, 12:18]
    assert pc.embed_case(100, 12, F32, lda=18)["max_abs"] < 2e-5               # edgez_embed reads x[:, :12]


# ---- split residual stream + LayerNorm fold (16-bit modes) ---------------------------
@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("with_res", [True, False])
def test_gemm_split_residual_and_row_stats(pc, dtype, with_res):
    small = pc.gemm_split_case(300, dtype, with_res=with_res)          # generic kernel (18 tiles)
    big = pc.gemm_split_case(2000, dtype, with_res=with_res)           # persistent kernel (96 tiles)
    ragged = pc.gemm_split_case(1999, dtype, K=1024, with_res=with_res)
    for e in (small, big, ragged):
        assert e["max_abs"] < (2e-3 if dtype == BF16 else 2e-4)       # hi + lo carries ~16 mantissa bits (values ~10)
        assert e["hi_is_rounding"]
        assert e["stats_sum_err"] < 2e-3 and e["stats_sq_rel"] < 1e-4
    # the two kernels agree bit for bit (same inputs for the first 300 rows: same generator seed and draw order? no --
    # so compare a persistent run against the generic run of ITS first rows)
    g = pc.gen(0)
    a = (torch.randn(2000, 768, generator=g) * 0.5).to(dtype).cuda()
    w = (torch.randn(768, 768, generator=g) * 0.05).to(dtype).cuda()
    x = (torch.randn(2000, 768, generator=g) * 3).cuda()
    hi, lo = x.to(dtype), (x - x.to(dtype).float()).to(dtype)
    full = pc.ops.linear_ex(a, w, None, split_out=True, want_stats=True, res=(hi, lo))
    part = pc.ops.linear_ex(a[:300].contiguous(), w, None, split_out=True, want_stats=True,
                            res=(hi[:300].contiguous(), lo[:300].contiguous()))
    for k in ("out", "lo"):
        assert torch.equal(full[k][:300], part[k]), k
    assert torch.equal(full["stats"][:, :300], part["stats"])


@pytest.mark.parametrize("dtype", [BF16, F16])
@pytest.mark.parametrize("N,act", [(2304, 0), (1024, 1)])
def test_gemm_layernorm_fold(pc, dtype, N, act):
    tol_alg = 6e-3 if dtype == BF16 else 8e-4        # output rounding of a 16-bit result (relative to max |y|)
    tol_ln = 1.2e-2 if dtype == BF16 else 1.6e-3     # + operand rounding of x and gamma*W
    small = pc.gemm_fold_case(300, N, dtype, act)
    big = pc.gemm_fold_case(2000, N, dtype, act)
    for e in (small, big):
        assert e["vs_algebra"] < tol_alg and e["vs_layernorm"] < tol_ln, e
    again = pc.gemm_fold_case(300, N, dtype, act)
    assert torch.equal(small["out"], again["out"])
    first = pc.gemm_fold_case(2000, N, dtype, act)
    assert torch.equal(first["out"], big["out"])


@pytest.mark.parametrize("dtype", [BF16, F16])
def test_layernorm_split_input(pc, dtype):
    e = pc.layernorm_split_case(130, dtype)
    assert e["finite"] and e["max_abs"] < (3e-2 if dtype == BF16 else 4e-3)


# bf16 eps against the fp32 oracle on the small seeded cases below: 2 x the largest value the MI355X measured on them
# (0.8-1.5e-2, gpurun_out/parity_r05.json -> profiles/r05/; the per-case golden bounds further down are per case)
BF16_EPS_BOUND = 3e-2


def test_fold_and_unfolded_paths_agree(pc):
    """fold_layernorm=False keeps the fp32 residual stream and the LayerNorm kernels; both must sit within the same
    distance of the oracle."""
    m, sd = pc.build_net("SurfZNet", 7, False, BF16)
    args = pc.synth_inputs("SurfZNet", 4, 60, 1, False)
    cu = [a.cuda() if torch.is_tensor(a) else a for a in args]
    with torch.no_grad():
        want = pc.orc.FORWARD["SurfZNet"](sd, *args)
        folded = m(*cu).cpu()
        m.fold_layernorm = False
        plain = m(*cu).cpu()
    valid = ~args[3]
    ef, ep, efp = (float((a - b)[valid].abs().max()) for a, b in ((folded, want), (plain, want), (folded, plain)))
    _record("fold_vs_unfolded_surfz_b4_bf16", folded_vs_oracle=ef, plain_vs_oracle=ep, folded_vs_plain=efp)
    assert ef < BF16_EPS_BOUND and ep < BF16_EPS_BOUND and efp < BF16_EPS_BOUND and not torch.equal(folded, plain)


def _record(key, **vals):
    """Measured parity numbers of this run -> gpurun_out/parity_r05.json (copied to profiles/ by hand: the bounds below are 2 x them)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_r05.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        d = {}
    d[key] = vals
    with open(path, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


# ---- whole denoisers ----------------------------------------------------------------------------------
GOLDEN = ["surfpos_b2_n30", "surfpos_cf_b2_n60", "surfz_b3_n60", "surfz_cf_b2_n17", "edgepos_b2_s6_e5",
          "edgez_b2_s7_e9", "edgez_cf_b2_s4_e40"]


@pytest.mark.parametrize("name", GOLDEN)
def test_denoiser_fp32_vs_reference_golden(pc, name):
    e = pc.golden_case(name, F32)
    assert e["finite"] and e["max_abs"] < 1e-5


# bf16 operands against the reference's own fp32 outputs: (max, mean) abs error bound per case = 2 x what the MI355X measured
# (profiles/r04/parity_r04.json, key golden_bf16_<case>; the values differ per case with |eps| and the token count).  north_star's
# "1e-3 bf16" is NOT met on eps itself -- by construction: 8 mantissa bits through 12 layers, the reference's own bf16 autocast sits
# at 2.4-3.3e-2 -- it is asserted where it is meaningful, on the per-step sample update (test_baseline_config0_ddpm_chain).
GOLDEN_BF16_BOUND = {"_default": (4e-2, 8e-3),
                     "edgepos_b2_s6_e5": (1.6e-2, 4.6e-3), "edgez_b2_s7_e9": (2.2e-2, 5.5e-3), "edgez_cf_b2_s4_e40": (2.6e-2, 5.7e-3),
                     "surfpos_b2_n30": (2.1e-2, 5.9e-3), "surfpos_cf_b2_n60": (2.5e-2, 6.8e-3), "surfz_b3_n60": (2.3e-2, 5.4e-3),
                     "surfz_cf_b2_n17": (2.9e-2, 7.2e-3)}


@pytest.mark.parametrize("name", GOLDEN)
def test_denoiser_bf16_vs_reference_golden(pc, name):
    e = pc.golden_case(name, BF16)
    _record("golden_bf16_" + name, max_abs_valid=e["max_abs_valid"], mean_abs=e["mean_abs"], ref_absmax=e["ref_absmax"])
    bmax, bmean = GOLDEN_BF16_BOUND.get(name, GOLDEN_BF16_BOUND["_default"])
    assert e["finite"] and e["max_abs_valid"] < bmax and e["mean_abs"] < bmean, e


def test_denoiser_vs_oracle_larger(pc):
    assert pc.oracle_case("SurfZNet", 4, 60, 1, F32)["max_abs"] < 1e-5
    assert pc.oracle_case("SurfPosNet", 5, 30, 1, F32, use_cf=True)["max_abs"] < 1e-5
    ez = pc.oracle_case("EdgeZNet", 1, 10, 20, BF16)["max_abs_valid"]
    ep = pc.oracle_case("EdgePosNet", 2, 8, 20, BF16, use_cf=True)["max_abs_valid"]
    _record("oracle_larger_bf16", edgez_b1_s10_e20=ez, edgepos_cf_b2_s8_e20=ep)
    assert ez < BF16_EPS_BOUND and ep < BF16_EPS_BOUND


def test_baseline_config0_ddpm_chain(pc):
    """BASELINE configs[0]: B=1 face-LDM, 50 DDPM steps with injected noise, per-step parity."""
    e = pc.ddpm_chain_case(F32, steps=50, oracle_dev=False)                      # (fp32 bound: against the oracle on the HOST;
    assert e["finite"] and e["max_abs_eps"] < 1e-5 and e["max_abs_x"] < 1e-5     # north_star: 1e-5 fp32 per step   the 16-bit
    #                                                                              runs below evaluate it with torch on the device)
    # bf16 operands: eps carries ~1.5e-2 (8 mantissa bits through 12 layers); with the 50-step schedule eps enters
    # x_{t-1} with a coefficient of up to ~0.3, with the sampling schedule (1000 steps, sample.py:144) ~0.007-0.02
    e = pc.ddpm_chain_case(BF16, steps=50)
    _record("ddpm_chain_bf16_50_steps", **e)
    assert e["finite"] and e["max_abs_eps"] < BF16_EPS_BOUND and e["max_abs_x"] < 1.2e-2
    e = pc.ddpm_chain_case(BF16, steps=1000, last=12)
    _record("ddpm_chain_bf16_last_12_of_1000", **e)
    assert e["finite"] and e["max_abs_eps"] < BF16_EPS_BOUND and e["max_abs_x"] < 1e-3     # north_star: 1e-3 bf16 per step


def test_conditioning_cache_and_masked_rows(pc):
    """Second call with the same conditioning tensors takes the cached-embed path and must not change eps;
    changing padded tokens must not move valid outputs (network.py:1196: padded tokens are only masked as keys)."""
    m, sd = pc.build_net("SurfZNet", 21, False, F32)
    z, t, pos, mask, _ = pc.synth_inputs("SurfZNet", 3, 60, 1, False)
    z, t, pos, mask = z.cuda(), t.cuda(), pos.cuda(), mask.cuda()
    with torch.no_grad():
        a = m(z, t, pos, mask, None)
        b = m(z, t, pos, mask, None)            # cache hit
        assert m._cond["valid"] and torch.equal(a, b)
        z2 = z.clone()
        z2[mask] = 77.0
        c = m(z2, t, pos, mask, None)
        assert float((a - c)[~mask].abs().max()) < 1e-5
        pos2 = pos.clone()                      # new conditioning tensor -> cache miss, different result
        pos2[~mask] += 0.5
        d = m(z, t, pos2, mask, None)
        assert float((a - d).abs().max()) > 1e-3


def test_full_size_properties(pc):
    """BASELINE configs[1] shape (B=512, N=60, bf16): size-independent properties instead of an oracle run --
    per-sample independence (a batch row equals the same sample run alone) and finiteness."""
    m, _ = pc.build_net("SurfZNet", 5, False, BF16)
    args = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 512, 60, 1, False)]
    with torch.no_grad():
        full = m(*args)
        assert torch.isfinite(full).all()
        for b in (0, 255, 511):
            one = m(args[0][b:b + 1].contiguous(), args[1], args[2][b:b + 1].contiguous(),
                    args[3][b:b + 1].contiguous(), None)
            valid = ~args[3][b]
            assert float((one[0] - full[b])[valid].abs().max()) < 1e-6     # bit-stable across batch sizes


# ---- VAE decoders (oracle parity-unpinned: diffusers blocks restated, see oracle/vae.py) ---------------------------
def test_upsample1d_cubic(pc):
    assert pc.upsample1d_case()["max_abs"] < 1e-5
    assert pc.upsample1d_case(S=2, L=4, C=512)["max_abs"] < 1e-5


@pytest.mark.parametrize("kind,n", [("surf", 3), ("edge", 7)])
def test_vae_decode_fp32(pc, kind, n):
    e = pc.vae_case(kind, n, F32)
    assert e["finite"] and e["max_abs"] < 2e-4 * max(1.0, e["ref_absmax"])


@pytest.mark.parametrize("kind,n", [("surf", 3), ("edge", 7)])
def test_vae_decode_bf16(pc, kind, n):
    # measured on MI355X (tools/vae_small_parity.py, profiles/r03/vae_small_parity.log): max 0.7-0.9 % of |ref|max, mean
    # 2.7-4.1e-3; asserted at 2x.  The same passes at sizes that take the implicit-GEMM path, next to torch.autocast of the
    # oracle: tests/test_gpu_round3.py::test_vae_decode_large_vs_oracle_and_torch_autocast
    e = pc.vae_case(kind, n, BF16)
    assert e["finite"] and e["max_abs"] < 0.02 * max(1.0, e["ref_absmax"]) and e["mean_abs"] < 0.009


def test_vae_decode_batch_independence(pc):
    """Chunked execution: a large batch equals the per-sample results (GroupNorm statistics are per sample)."""
    import brepgen_amd as bga
    from oracle import vae as ov
    sd = ov.seeded_state_dict(ov.edge_decoder_spec(), 77)
    m = bga.AutoencoderKL1DFastDecode(**pc.EDGE_CFG)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    m.compute_dtype = F32
    z = torch.randn(40, 3, 4, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        full = m(z)
        m.WS_BUDGET = 1 << 22                           # force several chunks inside bg_vae_run
        m.release_workspace()
        chunked = m(z)
        one = m(z[17:18])
    assert torch.equal(full, chunked) and float((full[17:18] - one).abs().max()) < 1e-5


# ---- fp16 operand mode (the reference's own autocast dtype, sample.py:121; BASELINE configs[4]) --------------------
@pytest.mark.parametrize("shape", [(60, 768, 768), (257, 1024, 768), (1000, 768, 1024), (300, 2304, 768)])
def test_gemm_fp16_exact_products(pc, shape):
    assert pc.gemm_case(*shape, F16)["max_abs"] < 2e-5
    assert pc.gemm_case(*shape, F16, add_mode="resid")["max_abs"] < 2e-5
    e = pc.gemm_case(*shape, F16, act=1, out_dtype=F16)
    assert e["max_abs"] < 6e-4 * e["ref_absmax"] + 1e-6                 # fp16 output rounding only (11-bit mantissa)


@pytest.mark.parametrize("N", [17, 60, 64, 130, 300])
def test_attention_fp16(pc, N):
    e = pc.attn_case(3, N, F16, "ragged")
    assert e["finite"] and e["max_abs"] < 5e-3 and e["mean_abs"] < 5e-4


@pytest.mark.parametrize("name", GOLDEN)
def test_denoiser_fp16_vs_reference_golden(pc, name):
    e = pc.golden_case(name, F16)
    assert e["finite"] and e["max_abs_valid"] < 8e-3 and e["mean_abs"] < 1.5e-3


def test_fp16_ddpm_chain_meets_1e3_per_step(pc):
    e = pc.ddpm_chain_case(F16, steps=50)
    assert e["finite"] and e["max_abs_eps"] < 8e-3 and e["max_abs_x"] < 2e-3
    e = pc.ddpm_chain_case(F16, steps=1000, last=12)
    assert e["finite"] and e["max_abs_x"] < 2e-4


def test_autocast_selects_operand_dtype(pc):
    m, _ = pc.build_net("SurfPosNet", 9, False, None)
    x, t, _ = pc.synth_inputs("SurfPosNet", 2, 30, 1, False)
    x, t = x.cuda(), t.cuda()
    with torch.no_grad():
        a = m(x, t, None)                                   # no autocast -> exact fp32
        with torch.autocast("cuda", dtype=torch.float16):
            b = m(x, t, None)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            c = m(x, t, None)
    assert {k[0] for k in m._packs} == {torch.float32, torch.float16, torch.bfloat16}
    assert 0 < float((a - b).abs().max()) < 8e-3 < 1e9 and float((a - b).abs().max()) < float((a - c).abs().max())


@pytest.mark.parametrize("kind,n", [("surf", 2), ("edge", 5)])
def test_vae_decode_fp16(pc, kind, n):
    e = pc.vae_case(kind, n, F16)                                  # measured 0.09-0.14 % of |ref|max, mean 3.7-5.3e-4
    assert e["finite"] and e["max_abs"] < 0.003 * max(1.0, e["ref_absmax"]) and e["mean_abs"] < 0.0012


# ---- VAE encoders (training-time API surface) ------------------------------------------
def test_downsample1d_cubic(pc):
    assert pc.downsample1d_case()["max_abs"] < 1e-5
    assert pc.downsample1d_case(S=2, L=32, C=128)["max_abs"] < 1e-5


@pytest.mark.parametrize("kind,n", [("surf_enc", 2), ("edge_enc", 6)])
def test_vae_encode_fp32(pc, kind, n):
    e = pc.vae_case(kind, n, F32)
    assert e["finite"] and e["max_abs"] < 2e-4 * max(1.0, e["ref_absmax"])


@pytest.mark.parametrize("kind,n", [("surf_enc", 2), ("edge_enc", 6)])
def test_vae_encode_16bit(pc, kind, n):
    e = pc.vae_case(kind, n, BF16)                                 # measured 0.66 / 1.03 % of |ref|max, mean 3.2-3.5e-3
    assert e["finite"] and e["max_abs"] < 0.022 * max(1.0, e["ref_absmax"]) and e["mean_abs"] < 0.008
    e = pc.vae_case(kind, n, F16)                                  # measured 0.10 / 0.15 %, mean 4.9-5.7e-4
    assert e["finite"] and e["max_abs"] < 0.003 * max(1.0, e["ref_absmax"]) and e["mean_abs"] < 0.0012


def test_per_sample_timesteps_training_style(pc):
    """timesteps of shape [B] (trainer.py:346-351 passes one t per sample) -> one time embedding per sample."""
    from oracle import denoisers as orc
    m, sd = pc.build_net("SurfZNet", 33, True, F32)
    z, _, pos, mask, cl = pc.synth_inputs("SurfZNet", 4, 20, 1, True)
    t = torch.tensor([3, 250, 999, 0])
    with torch.no_grad():
        want = orc.surfz_forward(sd, z, t, pos, mask, cl)
        got = m(z.cuda(), t.cuda(), pos.cuda(), mask.cuda(), cl.cuda())
    assert float((got.cpu() - want)[~mask].abs().max()) < 1e-5


# ---- training / validation forward (brepgen_amd/training.py) -------------------------------
def test_masked_mse_matches_torch(pc):
    from brepgen_amd import training
    g = pc.gen(8)
    pred, tgt = torch.randn(4, 60, 18, generator=g), torch.randn(4, 60, 18, generator=g)
    mask = torch.rand(4, 60, generator=g) < 0.3
    for m, cols in ((mask, None), (None, None), (mask, (0, 12)), (mask, (12, 18))):
        r = training.masked_mse(pred.cuda(), tgt.cuda(), None if m is None else m.cuda(), cols)
        sel = slice(None) if cols is None else slice(*cols)
        a, b = (pred, tgt) if m is None else (pred[~m], tgt[~m])
        a, b = a.reshape(-1, 18)[:, sel].double(), b.reshape(-1, 18)[:, sel].double()
        assert abs(float(r["mean"]) - float(((a - b) ** 2).mean())) < 1e-6
        assert abs(float(r["row_mean_sum"]) - float(((a - b) ** 2).mean(-1).sum())) < 1e-3
        assert int(r["rows"]) == a.shape[0]
    again = training.masked_mse(pred.cuda(), tgt.cuda(), mask.cuda())
    assert float(again["mean"]) == float(training.masked_mse(pred.cuda(), tgt.cuda(), mask.cuda())["mean"])   # deterministic


def test_encode_tokens_equal_the_trainer_permute_chains(pc):
    import brepgen_amd as bga
    from oracle import vae as ov
    g = pc.gen(9)
    se = bga.AutoencoderKLFastEncode(**pc.SURF_CFG)
    se.load_state_dict(ov.seeded_state_dict(ov.surf_encoder_spec(), 51), strict=True)
    ee = bga.AutoencoderKL1DFastEncode(**pc.EDGE_CFG)
    ee.load_state_dict(ov.seeded_state_dict(ov.edge_encoder_spec(), 61), strict=True)
    se, ee = se.cuda().eval(), ee.cuda().eval()
    surfPnt = torch.randn(2, 3, 32, 32, 3, generator=g).cuda()
    edgePnt = torch.randn(2, 3, 4, 32, 3, generator=g).cuda()
    with torch.no_grad():
        z = se(surfPnt.flatten(0, 1).permute(0, 3, 1, 2))                                            # trainer.py:519-524
        want = z.unflatten(0, (2, -1)).flatten(-2, -1).permute(0, 1, 3, 2).flatten(-2, -1)
        assert torch.equal(se.encode_tokens(surfPnt), want)
        z = ee(edgePnt.flatten(0, 1).flatten(0, 1).permute(0, 2, 1))                                 # trainer.py:924-929
        want = z.unflatten(0, (-1, 4)).unflatten(0, (2, -1)).permute(0, 1, 2, 4, 3).flatten(-2, -1)
        assert torch.equal(ee.encode_tokens(edgePnt), want)


def test_ldm_loss_vs_oracle_pipeline(pc):
    """add_noise (per-sample t) -> SurfZNet -> masked MSE against the same three steps done by the CPU oracle."""
    import brepgen_amd as bga
    from brepgen_amd import training
    from oracle.schedulers import OracleDDPM
    m, sd = pc.build_net("SurfZNet", 41, False, F32)
    z, _, pos, mask, _ = pc.synth_inputs("SurfZNet", 4, 30, 1, False)
    g = pc.gen(10)
    noise = torch.randn(z.shape, generator=g)
    t = torch.tensor([9, 49, 199, 499])
    ddpm = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                             beta_start=0.0001, beta_end=0.02, clip_sample=False)
    got = training.ldm_loss(m, ddpm, z.cuda(), t.cuda(), noise.cuda(), (pos.cuda(),), mask.cuda())
    x_t = OracleDDPM().add_noise(z, noise, t)
    pred = pc.orc.surfz_forward(sd, x_t, t, pos, mask)
    want = ((pred[~mask] - noise[~mask]) ** 2).mean()
    assert abs(float(got["mean"]) - float(want)) < 1e-5 * max(1.0, float(want))
    vals = training.validation_losses(m, ddpm, z.cuda(), (pos.cuda(),), mask.cuda(), generator=pc.gen(11))
    assert len(vals) == 5 and all(torch.isfinite(v["mean"]) for v in vals)


# ---- stand-alone entry points named by SURVEY.md section 8(b): bg_embed_mlp_fwd, bg_encoder_layer_fwd ---------------------
@pytest.mark.parametrize("dtype,tol", [(F32, 1e-5), (BF16, 3e-2), (F16, 4e-3)])
def test_encoder_layer_and_embed_mlp_entry_points(pc, dtype, tol):
    m, sd = pc.build_net("SurfZNet", 13, False, dtype)
    m.fold_layernorm = False                       # the stand-alone layer takes unfolded weights + an fp32 stream
    w, _keep = m._pack(dtype)
    g = pc.gen(12)
    B, N = 3, 41
    x = torch.randn(B, N, 768, generator=g)
    key_pad = torch.zeros(B, N, dtype=torch.bool)
    key_pad[0, 30:] = True
    key_pad[2, 5:] = True
    got = pc.ops.encoder_layer(w.layers[2], dtype, x.reshape(B * N, 768).cuda(), key_pad.cuda(), B, N).cpu().reshape(B, N, 768)
    want = pc.orc.encoder_layer(sd, 2, x, key_pad)
    assert float((got - want)[~key_pad].abs().max()) < tol * max(1.0, float(want.abs().max()))
    z = torch.randn(50, 48, generator=g)
    got = pc.ops.embed_mlp(w.embed[0], dtype, z.cuda()).cpu()                     # z_embed: 48 -> 768 -> 768
    want = pc.orc.embed_mlp(sd, "z_embed", z)
    assert float((got - want).abs().max()) < tol * max(1.0, float(want.abs().max()))
    m.fold_layernorm = True
    wf, _ = m._pack(dtype)
    if dtype != F32:                               # folded weights are refused with an explanation
        with pytest.raises(Exception, match="unfolded"):
            pc.ops.encoder_layer(wf.layers[2], dtype, x.reshape(B * N, 768).cuda(), None, B, N)


def test_hip_graph_capture_of_one_eps_eval(pc):
    """The whole-net call is graph-safe (no allocation, no synchronisation, no host round trip inside bg_denoiser_fwd):
    torch.cuda.graph captures one eps-evaluation + DDPM update and replays it bit-identically -- the launch-bound
    small-batch regime (BASELINE configs[0], B = 1: ~60 launches per step) can run as one graph launch."""
    import time
    import brepgen_amd as bga
    m, _ = pc.build_net("SurfZNet", 17, False, BF16)
    z, t, pos, mask, _ = [a.cuda() if torch.is_tensor(a) else a for a in pc.synth_inputs("SurfZNet", 1, 60, 1, False)]
    noise = torch.randn(z.shape, generator=pc.gen(13)).cuda()
    ddpm = bga.DDPMScheduler(num_train_timesteps=1000, beta_schedule="linear", prediction_type="epsilon",
                             beta_start=0.0001, beta_end=0.02, clip_sample=True, clip_sample_range=3)
    ddpm.set_timesteps(1000)

    def step():
        return ddpm.step(m(z, t, pos, mask, None), 249, z, noise=noise).prev_sample

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream: workspace, packs, embed cache
            for _ in range(3):
                want = step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            got = step()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        z.add_(0.25)                                       # static input buffers: new contents, same graph
        want2 = step()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want2) and not torch.equal(want2, want)

        def timed(fn, n=50):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        print(f"B=1 step: eager {timed(step):.0f} us, graph replay {timed(graph.replay):.0f} us")
