"""CPU restatement of the two VAE *decoders* BrepGen samples with (sample.py:72-99, 289-294).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinning: the 2-D (surface) encoder / decoder restatement is PINNED to an
independent implementation of the same published network (``transformers``' Janus VQ-VAE encoder / decoder,
``tests/test_oracle_vae_pin.py``); the 1-D (edge) blocks are *** PARITY UNPINNED ***: the decoder blocks live in the
third-party ``diffusers==0.27`` package (``network.py:12-13`` imports ``Decoder``, ``ResConvBlock``,
``SelfAttention1d``, ``Upsample1d`` from it), which is neither vendored under /root/reference nor installable
offline.  The repo-side wiring that IS in the reference is followed line by line:

  AutoencoderKLFastDecode      network.py:948-1040  (post_quant_conv 1x1 -> diffusers ``Decoder``)
  AutoencoderKL1DFastDecode    network.py:786-858   (post_quant_conv 1x1 -> ``Decoder1D``)
  Decoder1D                    network.py:188-299
  UNetMidBlock1D               network.py:51-83     (6 x [ResConvBlock -> SelfAttention1d])
  UpBlock1D                    network.py:30-48     (3 ResConvBlocks + Upsample1d("cubic"))
  constructor arguments        sample.py:72-82 (block_out_channels [128,256,512,512], layers_per_block 2,
                               32 groups) and sample.py:86-97 ([128,256,512])

and the diffusers blocks are restated from the published 0.27 sources: ``ResnetBlock2D``
(GroupNorm(32, eps 1e-6) -> SiLU -> conv3x3, twice, + 1x1 shortcut when channels change), the single-head
mid-block ``Attention`` (GroupNorm, q/k/v/out Linear, scale 1/sqrt(C), residual), ``Upsample2D`` (nearest x2 +
conv3x3), ``ResConvBlock`` (conv k5 -> GroupNorm(1) -> GELU, twice, + 1x1 skip), ``SelfAttention1d`` (GroupNorm(1),
heads = C/32, scale d^-1/4 on q and k) and ``Upsample1d("cubic")`` (reflect pad 2, stride-2 transposed conv with the
8-tap cubic kernel x2, padding 7).  State-dict keys follow the diffusers module tree so a BrepGen ``*_vae_*.pt``
loads with ``strict=False`` as at sample.py:83,98.
"""
import math

import torch
import torch.nn.functional as F

CUBIC = [-0.01171875, -0.03515625, 0.11328125, 0.43359375, 0.43359375, 0.11328125, -0.03515625, -0.01171875]


# ------------------------------------------------------------------------------------------------
# 2-D surface decoder
# ------------------------------------------------------------------------------------------------
def _resnet2d(sd, p, x, groups=32, eps=1e-6):
    h = F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    h = F.conv2d(F.silu(h), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
    h = F.conv2d(F.silu(h), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def _attn2d(sd, p, x, groups=32, eps=1e-6):
    B, C, H, W = x.shape
    h = F.group_norm(x.reshape(B, C, H * W), groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], eps)
    h = h.transpose(1, 2)                                                   # [B, HW, C]
    q = F.linear(h, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(h, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(h, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) * (1.0 / math.sqrt(C)), dim=-1)  # heads = 1, dim_head = C
    o = F.linear(a @ v, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def surf_decode(sd, z, n_up=4, layers_per_block=2, groups=32):
    """AutoencoderKLFastDecode.forward: z [F,3,4,4] -> [F,3,32,32]."""
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _resnet2d(sd, "decoder.mid_block.resnets.0.", x, groups)
    x = _attn2d(sd, "decoder.mid_block.attentions.0.", x, groups)
    x = _resnet2d(sd, "decoder.mid_block.resnets.1.", x, groups)
    for b in range(n_up):
        for r in range(layers_per_block + 1):
            x = _resnet2d(sd, f"decoder.up_blocks.{b}.resnets.{r}.", x, groups)
        key = f"decoder.up_blocks.{b}.upsamplers.0.conv."
        if key + "weight" in sd:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[key + "weight"], sd[key + "bias"], padding=1)
    x = F.group_norm(x, groups, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6)
    return F.conv2d(F.silu(x), sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def surf_decoder_spec(block_out=(128, 256, 512, 512), layers_per_block=2, latent=3, out_ch=3):
    """Ordered {key: shape} of the decoder half of a surface-VAE checkpoint (diffusers AutoencoderKL layout)."""
    spec = {"post_quant_conv.weight": (latent, latent, 1, 1), "post_quant_conv.bias": (latent,)}
    top = block_out[-1]
    spec["decoder.conv_in.weight"] = (top, latent, 3, 3)
    spec["decoder.conv_in.bias"] = (top,)

    def resnet(p, cin, cout):
        spec[p + "norm1.weight"] = (cin,); spec[p + "norm1.bias"] = (cin,)
        spec[p + "conv1.weight"] = (cout, cin, 3, 3); spec[p + "conv1.bias"] = (cout,)
        spec[p + "norm2.weight"] = (cout,); spec[p + "norm2.bias"] = (cout,)
        spec[p + "conv2.weight"] = (cout, cout, 3, 3); spec[p + "conv2.bias"] = (cout,)
        if cin != cout:
            spec[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); spec[p + "conv_shortcut.bias"] = (cout,)

    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    spec[a + "group_norm.weight"] = (top,); spec[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        spec[a + n + ".weight"] = (top, top); spec[a + n + ".bias"] = (top,)
    resnet("decoder.mid_block.resnets.1.", top, top)
    rev = list(reversed(block_out))
    prev = rev[0]
    for b, ch in enumerate(rev):
        for r in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{b}.resnets.{r}.", prev if r == 0 else ch, ch)
        if b != len(rev) - 1:
            spec[f"decoder.up_blocks.{b}.upsamplers.0.conv.weight"] = (ch, ch, 3, 3)
            spec[f"decoder.up_blocks.{b}.upsamplers.0.conv.bias"] = (ch,)
        prev = ch
    spec["decoder.conv_norm_out.weight"] = (block_out[0],); spec["decoder.conv_norm_out.bias"] = (block_out[0],)
    spec["decoder.conv_out.weight"] = (out_ch, block_out[0], 3, 3); spec["decoder.conv_out.bias"] = (out_ch,)
    return spec


# ------------------------------------------------------------------------------------------------
# 1-D edge decoder
# ------------------------------------------------------------------------------------------------
def _resconv(sd, p, x):
    res = F.conv1d(x, sd[p + "conv_skip.weight"]) if p + "conv_skip.weight" in sd else x
    h = F.conv1d(x, sd[p + "conv_1.weight"], sd[p + "conv_1.bias"], padding=2)
    h = F.gelu(F.group_norm(h, 1, sd[p + "group_norm_1.weight"], sd[p + "group_norm_1.bias"], 1e-5))
    h = F.conv1d(h, sd[p + "conv_2.weight"], sd[p + "conv_2.bias"], padding=2)
    h = F.gelu(F.group_norm(h, 1, sd[p + "group_norm_2.weight"], sd[p + "group_norm_2.bias"], 1e-5))
    return h + res


def _attn1d(sd, p, x):
    B, C, L = x.shape
    nh = C // 32                                                            # network.py:68-75: heads = channels // 32
    h = F.group_norm(x, 1, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-5).transpose(1, 2)
    def heads(t):
        return t.reshape(B, L, nh, C // nh).permute(0, 2, 1, 3)
    q = heads(F.linear(h, sd[p + "query.weight"], sd[p + "query.bias"]))
    k = heads(F.linear(h, sd[p + "key.weight"], sd[p + "key.bias"]))
    v = heads(F.linear(h, sd[p + "value.weight"], sd[p + "value.bias"]))
    s = 1.0 / math.sqrt(math.sqrt(C // nh))
    a = torch.softmax((q * s) @ (k.transpose(-1, -2) * s), dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, L, C)
    o = F.linear(o, sd[p + "proj_attn.weight"], sd[p + "proj_attn.bias"])
    return x + o.transpose(1, 2)


def upsample1d_cubic(x):
    """diffusers Upsample1d("cubic") exactly as upstream builds it: dense diagonal transposed conv."""
    C = x.shape[1]
    k = torch.tensor(CUBIC, dtype=x.dtype, device=x.device) * 2
    xp = F.pad(x, (2, 2), mode="reflect")
    w = x.new_zeros(C, C, 8)
    idx = torch.arange(C, device=x.device)
    w[idx, idx] = k
    return F.conv_transpose1d(xp, w, stride=2, padding=7)


def upsample1d_cubic_taps(x):
    """The same operator written as the 4-tap depthwise gather the HIP kernel implements (checked == above)."""
    B, C, L = x.shape
    k = torch.tensor(CUBIC, dtype=x.dtype) * 2
    xp = F.pad(x, (2, 2), mode="reflect")                                   # hp[i] = x[reflect(i - 2)]
    out = x.new_zeros(B, C, 2 * L)
    for o in range(2 * L):
        for i in range((o + 1) // 2, (o + 7) // 2 + 1):
            kk = o + 7 - 2 * i
            if 0 <= kk < 8 and 0 <= i < L + 4:
                out[:, :, o] += xp[:, :, i] * k[kk]
    return out


def edge_decode(sd, z, n_up=3):
    """AutoencoderKL1DFastDecode.forward: z [G,3,4] -> [G,3,32]."""
    x = F.conv1d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv1d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    for i in range(6):
        x = _resconv(sd, f"decoder.mid_block.resnets.{i}.", x)
        x = _attn1d(sd, f"decoder.mid_block.attentions.{i}.", x)
    for b in range(n_up):
        for r in range(3):
            x = _resconv(sd, f"decoder.up_blocks.{b}.resnets.{r}.", x)
        x = upsample1d_cubic(x)
    x = F.group_norm(x, 32, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6)
    return F.conv1d(F.silu(x), sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def edge_decoder_spec(block_out=(128, 256, 512), latent=3, out_ch=3):
    spec = {"post_quant_conv.weight": (latent, latent, 1), "post_quant_conv.bias": (latent,)}
    top = block_out[-1]
    spec["decoder.conv_in.weight"] = (top, latent, 3); spec["decoder.conv_in.bias"] = (top,)

    def resconv(p, cin, mid, cout):
        if cin != cout:
            spec[p + "conv_skip.weight"] = (cout, cin, 1)
        spec[p + "conv_1.weight"] = (mid, cin, 5); spec[p + "conv_1.bias"] = (mid,)
        spec[p + "group_norm_1.weight"] = (mid,); spec[p + "group_norm_1.bias"] = (mid,)
        spec[p + "conv_2.weight"] = (cout, mid, 5); spec[p + "conv_2.bias"] = (cout,)
        spec[p + "group_norm_2.weight"] = (cout,); spec[p + "group_norm_2.bias"] = (cout,)

    for i in range(6):
        resconv(f"decoder.mid_block.resnets.{i}.", top, top, top)
        a = f"decoder.mid_block.attentions.{i}."
        spec[a + "group_norm.weight"] = (top,); spec[a + "group_norm.bias"] = (top,)
        for n in ("query", "key", "value", "proj_attn"):
            spec[a + n + ".weight"] = (top, top); spec[a + n + ".bias"] = (top,)
    rev = list(reversed(block_out))
    prev = rev[0]
    for b, ch in enumerate(rev):                                            # network.py:223-236
        resconv(f"decoder.up_blocks.{b}.resnets.0.", prev, prev, prev)
        resconv(f"decoder.up_blocks.{b}.resnets.1.", prev, prev, prev)
        resconv(f"decoder.up_blocks.{b}.resnets.2.", prev, prev, ch)
        spec[f"decoder.up_blocks.{b}.up.kernel"] = (8,)
        prev = ch
    spec["decoder.conv_norm_out.weight"] = (block_out[0],); spec["decoder.conv_norm_out.bias"] = (block_out[0],)
    spec["decoder.conv_out.weight"] = (out_ch, block_out[0], 3); spec["decoder.conv_out.bias"] = (out_ch,)
    return spec


# ------------------------------------------------------------------------------------------------
# encoders (training-time API surface: AutoencoderKLFastEncode network.py:861-945, AutoencoderKL1DFastEncode 690-783)
# ------------------------------------------------------------------------------------------------
def surf_encode(sd, x, n_down=4, layers_per_block=2, groups=32, latent=3):
    """AutoencoderKLFastEncode.forward: x [F,3,32,32] -> DiagonalGaussian mode [F,3,4,4] (network.py:941-945)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for b in range(n_down):
        for r in range(layers_per_block):
            h = _resnet2d(sd, f"encoder.down_blocks.{b}.resnets.{r}.", h, groups)
        key = f"encoder.down_blocks.{b}.downsamplers.0.conv."
        if key + "weight" in sd:                                             # Downsample2D(padding=0): pad (0,1,0,1)
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[key + "weight"], sd[key + "bias"], stride=2)
    h = _resnet2d(sd, "encoder.mid_block.resnets.0.", h, groups)
    h = _attn2d(sd, "encoder.mid_block.attentions.0.", h, groups)
    h = _resnet2d(sd, "encoder.mid_block.resnets.1.", h, groups)
    h = F.group_norm(h, groups, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6)
    h = F.conv2d(F.silu(h), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    moments = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return moments[:, :latent]                                                # mode() = mean


def surf_encoder_spec(block_out=(128, 256, 512, 512), layers_per_block=2, latent=3, in_ch=3):
    spec = {"quant_conv.weight": (2 * latent, 2 * latent, 1, 1), "quant_conv.bias": (2 * latent,)}
    spec["encoder.conv_in.weight"] = (block_out[0], in_ch, 3, 3); spec["encoder.conv_in.bias"] = (block_out[0],)

    def resnet(p, cin, cout):
        spec[p + "norm1.weight"] = (cin,); spec[p + "norm1.bias"] = (cin,)
        spec[p + "conv1.weight"] = (cout, cin, 3, 3); spec[p + "conv1.bias"] = (cout,)
        spec[p + "norm2.weight"] = (cout,); spec[p + "norm2.bias"] = (cout,)
        spec[p + "conv2.weight"] = (cout, cout, 3, 3); spec[p + "conv2.bias"] = (cout,)
        if cin != cout:
            spec[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); spec[p + "conv_shortcut.bias"] = (cout,)

    prev = block_out[0]
    for b, ch in enumerate(block_out):
        for r in range(layers_per_block):
            resnet(f"encoder.down_blocks.{b}.resnets.{r}.", prev if r == 0 else ch, ch)
        if b != len(block_out) - 1:
            spec[f"encoder.down_blocks.{b}.downsamplers.0.conv.weight"] = (ch, ch, 3, 3)
            spec[f"encoder.down_blocks.{b}.downsamplers.0.conv.bias"] = (ch,)
        prev = ch
    top = block_out[-1]
    resnet("encoder.mid_block.resnets.0.", top, top)
    a = "encoder.mid_block.attentions.0."
    spec[a + "group_norm.weight"] = (top,); spec[a + "group_norm.bias"] = (top,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        spec[a + n + ".weight"] = (top, top); spec[a + n + ".bias"] = (top,)
    resnet("encoder.mid_block.resnets.1.", top, top)
    spec["encoder.conv_norm_out.weight"] = (top,); spec["encoder.conv_norm_out.bias"] = (top,)
    spec["encoder.conv_out.weight"] = (2 * latent, top, 3, 3); spec["encoder.conv_out.bias"] = (2 * latent,)
    return spec


def downsample1d_cubic(x):
    """diffusers Downsample1d("cubic"): reflect pad 3, dense-diagonal stride-2 conv with the 8-tap kernel."""
    C = x.shape[1]
    k = torch.tensor(CUBIC, dtype=x.dtype, device=x.device)
    xp = F.pad(x, (3, 3), mode="reflect")
    w = x.new_zeros(C, C, 8)
    idx = torch.arange(C, device=x.device)
    w[idx, idx] = k
    return F.conv1d(xp, w, stride=2)


def edge_encode(sd, x, n_down=3, latent=3):
    """AutoencoderKL1DFastEncode.forward: x [G,3,32] -> mode [G,3,4] (network.py:765-783; Encoder1D 86-185)."""
    h = F.conv1d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for b in range(n_down):                                                   # DownBlock1D: down, then 3 ResConvBlocks
        h = downsample1d_cubic(h)
        for r in range(3):
            h = _resconv(sd, f"encoder.down_blocks.{b}.resnets.{r}.", h)
    for i in range(6):
        h = _resconv(sd, f"encoder.mid_block.resnets.{i}.", h)
        h = _attn1d(sd, f"encoder.mid_block.attentions.{i}.", h)
    h = F.group_norm(h, 32, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6)
    h = F.conv1d(F.silu(h), sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    moments = F.conv1d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return moments[:, :latent]


def edge_encoder_spec(block_out=(128, 256, 512), latent=3, in_ch=3):
    spec = {"quant_conv.weight": (2 * latent, 2 * latent, 1), "quant_conv.bias": (2 * latent,)}
    spec["encoder.conv_in.weight"] = (block_out[0], in_ch, 3); spec["encoder.conv_in.bias"] = (block_out[0],)

    def resconv(p, cin, mid, cout):
        if cin != cout:
            spec[p + "conv_skip.weight"] = (cout, cin, 1)
        spec[p + "conv_1.weight"] = (mid, cin, 5); spec[p + "conv_1.bias"] = (mid,)
        spec[p + "group_norm_1.weight"] = (mid,); spec[p + "group_norm_1.bias"] = (mid,)
        spec[p + "conv_2.weight"] = (cout, mid, 5); spec[p + "conv_2.bias"] = (cout,)
        spec[p + "group_norm_2.weight"] = (cout,); spec[p + "group_norm_2.bias"] = (cout,)

    prev = block_out[0]
    for b, ch in enumerate(block_out):                                       # DownBlock1D(out_channels=ch, in_channels=prev)
        spec[f"encoder.down_blocks.{b}.down.kernel"] = (8,)
        resconv(f"encoder.down_blocks.{b}.resnets.0.", prev, ch, ch)
        resconv(f"encoder.down_blocks.{b}.resnets.1.", ch, ch, ch)
        resconv(f"encoder.down_blocks.{b}.resnets.2.", ch, ch, ch)
        prev = ch
    top = block_out[-1]
    for i in range(6):
        resconv(f"encoder.mid_block.resnets.{i}.", top, top, top)
        a = f"encoder.mid_block.attentions.{i}."
        spec[a + "group_norm.weight"] = (top,); spec[a + "group_norm.bias"] = (top,)
        for n in ("query", "key", "value", "proj_attn"):
            spec[a + n + ".weight"] = (top, top); spec[a + n + ".bias"] = (top,)
    spec["encoder.conv_norm_out.weight"] = (top,); spec["encoder.conv_norm_out.bias"] = (top,)
    spec["encoder.conv_out.weight"] = (2 * latent, top, 3); spec["encoder.conv_out.bias"] = (2 * latent,)
    return spec


def seeded_state_dict(spec, seed):
    """Deterministic synthetic weights: conv/linear ~ N(0, 1/fan_in), norm gains 1 + N(0, 0.1^2), biases N(0, 0.02^2)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in spec.items():
        if key.endswith("up.kernel"):
            sd[key] = torch.tensor(CUBIC) * 2
        elif key.endswith("down.kernel"):
            sd[key] = torch.tensor(CUBIC)
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[key] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif key.endswith("weight"):
            sd[key] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            sd[key] = 0.02 * torch.randn(shape, generator=g)
    return sd
