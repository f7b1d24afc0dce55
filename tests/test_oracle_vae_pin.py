"""Independent pin of the restated 2-D (surface) VAE encoder / decoder.

diffusers==0.27 -- whose `Decoder` / `Encoder` / `ResnetBlock2D` / `Attention` / `Upsample2D` / `Downsample2D` the reference
instantiates (network.py:12-14, 861-1040) -- is neither vendored nor installable offline, so `oracle/vae.py` is a
restatement.  The architecture, however, is the published latent-diffusion auto-encoder (Rombach et al. 2022; the
"taming" encoder/decoder), and the image carries an INDEPENDENT implementation of exactly that network: Hugging Face
`transformers`' `JanusVQVAEEncoder` / `JanusVQVAEDecoder` (ResnetBlock: GroupNorm(32, eps 1e-6) -> swish -> conv3x3, twice,
+ 1x1 shortcut; single-head AttnBlock with 1x1 q/k/v/proj and scale C^-1/2; nearest-2x + conv3x3 up-sampler; pad(0,1,0,1)
+ stride-2 conv down-sampler; mid = block, attn, block; GroupNorm -> swish -> conv_out).  With BrepGen's hyper-parameters
(sample.py:72-82: channels 128/256/512/512, 2 layers per block, 3 latent channels) and the per-level attention blocks
Janus adds at the lowest resolution removed, the two networks are the same function.  This test maps a seeded
diffusers-keyed state dict onto the transformers modules and requires the restatement to agree to fp32 round-off:
the surface decoder and the 2-D half of the encoders (a14) are thereby pinned to third-party code, not
only to the author's reading of diffusers.  (The 1-D edge VAE uses diffusers' dance-diffusion blocks, for which the image
holds no second implementation: it stays anchored on network.py:30-299 + parameter counts + the cubic-kernel constants.)
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import vae as ov

janus = pytest.importorskip("transformers.models.janus.modeling_janus")
from transformers.models.janus.configuration_janus import JanusVQVAEConfig  # noqa: E402


def _cfg():
    return JanusVQVAEConfig(latent_channels=3, in_channels=3, out_channels=3, base_channels=128,
                            channel_multiplier=(1, 2, 4, 4), num_res_blocks=2, double_latent=True, dropout=0.0)


def _copy(mod, sd, prefix):
    with torch.no_grad():
        mod.weight.copy_(sd[prefix + ".weight"].reshape(mod.weight.shape))       # Linear [C,C] <-> 1x1 conv [C,C,1,1]
        mod.bias.copy_(sd[prefix + ".bias"])


def _load_resnet(blk, sd, p):
    for n in ("norm1", "conv1", "norm2", "conv2"):
        _copy(getattr(blk, n), sd, p + n)
    if blk.in_channels != blk.out_channels:
        _copy(blk.nin_shortcut, sd, p + "conv_shortcut")


def _load_mid(mid, sd, p):
    _load_resnet(mid.block_1, sd, p + "resnets.0.")
    _load_resnet(mid.block_2, sd, p + "resnets.1.")
    a = p + "attentions.0."
    _copy(mid.attn_1.norm, sd, a + "group_norm")
    for theirs, ours in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")):
        _copy(getattr(mid.attn_1, theirs), sd, a + ours)


def test_surface_decoder_restatement_equals_the_transformers_ldm_decoder():
    sd = ov.seeded_state_dict(ov.surf_decoder_spec(), 31)
    dec = janus.JanusVQVAEDecoder(_cfg()).eval()
    dec.up[0].attn = nn.ModuleList()                          # diffusers' UpDecoderBlock2D has no attention
    _copy(dec.conv_in, sd, "decoder.conv_in")
    _load_mid(dec.mid, sd, "decoder.mid_block.")
    for b in range(4):
        for r in range(3):
            _load_resnet(dec.up[b].block[r], sd, f"decoder.up_blocks.{b}.resnets.{r}.")
        if b != 3:
            _copy(dec.up[b].upsample.conv, sd, f"decoder.up_blocks.{b}.upsamplers.0.conv")
    _copy(dec.norm_out, sd, "decoder.conv_norm_out")
    _copy(dec.conv_out, sd, "decoder.conv_out")
    z = torch.randn(3, 3, 4, 4, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = dec(F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"]))     # network.py:1033-1035
        got = ov.surf_decode(sd, z)
    assert got.shape == want.shape == (3, 3, 32, 32)
    assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_surface_encoder_restatement_equals_the_transformers_ldm_encoder():
    sd = ov.seeded_state_dict(ov.surf_encoder_spec(), 51)
    enc = janus.JanusVQVAEEncoder(_cfg()).eval()
    enc.down[3].attn = nn.ModuleList()                        # diffusers' DownEncoderBlock2D has no attention
    _copy(enc.conv_in, sd, "encoder.conv_in")
    for b in range(4):
        for r in range(2):
            _load_resnet(enc.down[b].block[r], sd, f"encoder.down_blocks.{b}.resnets.{r}.")
        if b != 3:
            _copy(enc.down[b].downsample.conv, sd, f"encoder.down_blocks.{b}.downsamplers.0.conv")
    _load_mid(enc.mid, sd, "encoder.mid_block.")
    _copy(enc.norm_out, sd, "encoder.conv_norm_out")
    _copy(enc.conv_out, sd, "encoder.conv_out")
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        moments = F.conv2d(enc(x), sd["quant_conv.weight"], sd["quant_conv.bias"])            # network.py:941-944
        want = moments[:, :3]                                 # DiagonalGaussianDistribution(moments).mode() = mean
        got = ov.surf_encode(sd, x)
    assert got.shape == want.shape == (2, 3, 4, 4)
    assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
