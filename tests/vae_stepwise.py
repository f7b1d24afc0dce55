"""Step-by-step Python driver of the VAE primitives (the round-1 execution path): the same convolution-net steps the product
runs as ONE bg_vae_run program (brepgen_amd/vae.py), issued one C call at a time with torch-allocated intermediates.  Test
infrastructure only -- it cross-checks the program / the implicit-GEMM convolutions bit for bit
(tests/test_gpu_round2.py::test_vae_program_equals_step_by_step, ::test_vae_implicit_gemm_equals_materialised_im2col) and
serves tools/vae_bench.py as the A/B baseline.

    from vae_stepwise import stepwise
    y = stepwise(module, x, implicit_gemm=True)      # module: one of the four brepgen_amd VAE modules, x as module.forward takes it
"""
import ctypes
import math

import torch

from brepgen_amd import _lib
import hip_ops as ops
from brepgen_amd._lib import BG_F32, check, ptr, stream
from brepgen_amd.vae import ACT_GELU, ACT_NONE, ACT_SILU, _CODE, _pow2

IM2COL_BUDGET = 1 << 32        # bytes of im2col scratch per chunk of samples


class _Steps:
    """The primitives, written as methods over the product module `m` (weights packs, zero page, configuration)."""

    def __init__(self, m, implicit_gemm=True, im2col_budget=IM2COL_BUDGET):
        self.m, self.implicit_gemm, self.budget = m, implicit_gemm, im2col_budget

    def __getattr__(self, name):                       # everything else (packs, decoder / encoder sub-modules, ...) is the module's
        return getattr(self.m, name)

    def _stats(self, x, S, P, C, norm):
        st = torch.empty(S, norm.num_groups, 2, device=x.device, dtype=torch.float32)
        check(_lib.load().bg_groupnorm_stats(ptr(x), ptr(st), S, P, C, norm.num_groups, norm.eps, stream()),
              "bg_groupnorm_stats")
        return st

    def _conv(self, x, shape, pk, kh, kw, up=0, norm=None, act=ACT_NONE, residual=None, stride=1, pad=None):
        """conv (kh x kw) on channels-last x; pad=None: 'same' (kh//2, kw//2); pad=(py, px): zeros before only."""
        S, H, W, C = shape
        lib = _lib.load()
        Hl, Wl = H << up, W << up
        if pad is None:
            py, px = kh // 2, kw // 2
            Ho, Wo = (Hl + 2 * py - kh) // stride + 1, (Wl + 2 * px - kw) // stride + 1
        else:                                          # Downsample2D: F.pad(x, (0,1,0,1)) then stride-2 conv, no padding
            py, px = pad
            Ho, Wo = (Hl + 1 - kh) // stride + 1 if kh > 1 else Hl, (Wl + 1 - kw) // stride + 1
        rows = S * Ho * Wo
        st = self._stats(x, S, H * W, C, norm) if norm is not None else None
        g = norm.weight.detach().float().contiguous() if norm is not None else None
        b = norm.bias.detach().float().contiguous() if norm is not None else None
        implicit = (self.implicit_gemm and pk.dtype != torch.float32 and kh * kw > 1 and stride == 1 and pad is None
                    and C % 64 == 0 and _pow2(C // 64) and _pow2(Ho) and _pow2(Wo) and pk.n % 128 == 0
                    and pk.w.shape[0] == pk.n and ((rows + 127) // 128) * (pk.n // 128) >= 64 and rows < 2 ** 31)
        if implicit:
            # normalise + activate + cast ONCE (a 1x1 "im2col"), then let the GEMM's loader walk the window
            xn = torch.empty(S * H * W, C, device=x.device, dtype=pk.dtype)
            check(lib.bg_im2col(ptr(x), ptr(xn), _CODE[pk.dtype], S, H, W, C, 1, 1, 0, 1, 0, 0, H, W, ptr(st), ptr(g), ptr(b),
                                norm.num_groups if norm is not None else 1, act, None, stream()), "bg_im2col[norm+act+cast]")
            out = torch.empty(rows, pk.n, device=x.device, dtype=torch.float32)
            res = residual.contiguous() if residual is not None else None
            d = _lib.ConvDesc()
            d.x, d.S, d.H, d.W, d.C = ptr(xn), S, H, W, C
            d.kh, d.kw, d.up = kh, kw, up
            d.w, d.bias, d.N = ptr(pk.w), ptr(pk.b), pk.n
            d.out, d.ldc = ptr(out), pk.n
            d.add, d.ld_add = ptr(res), pk.n
            d.dtype, d.zero_page = _CODE[pk.dtype], ptr(self._zero_page(x.device))
            check(lib.bg_conv_gemm_fwd(ctypes.byref(d), stream()), "bg_conv_gemm_fwd")
            return out, (S, Ho, Wo, pk.n)
        a = torch.empty(rows, kh * kw * C, device=x.device, dtype=pk.dtype)
        check(lib.bg_im2col(ptr(x), ptr(a), _CODE[pk.dtype], S, H, W, C, kh, kw, up, stride, py, px, Ho, Wo,
                            ptr(st), ptr(g), ptr(b), norm.num_groups if norm is not None else 1, act, None, stream()),
              "bg_im2col")
        out = ops.linear(a, pk.w, pk.b, out_dtype=torch.float32, add=residual, add_div=1, n_valid=pk.n)
        return out, (S, Ho, Wo, pk.n)

    def _resnet(self, x, shape, P, name, r):
        S, H, W, C = shape
        h, hs = self._conv(x, shape, P[name + "c1"], 3, 3, norm=r.norm1, act=ACT_SILU)
        if name + "sc" in P:
            x, _ = self._conv(x, shape, P[name + "sc"], 1, 1)
        out, os_ = self._conv(h, hs, P[name + "c2"], 3, 3, norm=r.norm2, act=ACT_SILU, residual=x)
        return out, os_

    def _attn2d(self, x, shape, P, key, at):
        """diffusers Attention of the 2-D mid block: 1 head over H*W tokens, dim_head = C."""
        S_, H, W, C = shape
        qkv, _ = self._conv(x, shape, P[key + "qkv"], 1, 1, norm=at.group_norm)
        o = torch.empty(S_ * H * W, C, device=x.device, dtype=P[key + "proj"].dtype)
        check(_lib.load().bg_small_attn(ptr(qkv), 3 * C, ptr(o), _CODE[o.dtype], S_, H * W, C, 1, 1.0 / math.sqrt(C),
                                        stream()), "bg_small_attn")
        return ops.linear(o, P[key + "proj"].w, P[key + "proj"].b, out_dtype=torch.float32, add=x, n_valid=C)

    def _resconv(self, x, shape, P, name, r):
        # ResConvBlock: conv k5 -> GroupNorm(1) -> GELU -> conv k5 -> GroupNorm(1) -> GELU, + (1x1) skip.
        # group_norm_1 + GELU fold into the gather of conv_2; the trailing group_norm_2 + GELU cannot fold into the
        # next consumer (the residual add sits in between), so it is one 1x1 "im2col" pass with the add fused.
        h, hs = self._conv(x, shape, P[name + "c1"], 1, 5)
        h, hs = self._conv(h, hs, P[name + "c2"], 1, 5, norm=r.group_norm_1, act=ACT_GELU)
        S, H, W, C = hs
        res = x
        if name + "sk" in P:
            res, _ = self._conv(x, shape, P[name + "sk"], 1, 1)
        st = self._stats(h, S, H * W, C, r.group_norm_2)
        y = torch.empty(S * H * W, C, device=h.device, dtype=torch.float32)
        check(_lib.load().bg_im2col(ptr(h), ptr(y), BG_F32, S, H, W, C, 1, 1, 0, 1, 0, 0, H, W, ptr(st),
                                    ptr(r.group_norm_2.weight.detach().float().contiguous()),
                                    ptr(r.group_norm_2.bias.detach().float().contiguous()), 1, ACT_GELU,
                                    ptr(res.contiguous()), stream()),
              "bg_im2col[norm+gelu+residual]")
        return y, hs

    def _attn1d(self, x, shape, P, key, at):
        S, H, W, C = shape
        qkv, _ = self._conv(x, shape, P[key + "qkv"], 1, 1, norm=at.group_norm)
        nh = C // 32
        pk = P[key + "proj"]
        o = torch.empty(S * W, C, device=x.device, dtype=pk.dtype)
        check(_lib.load().bg_small_attn(ptr(qkv), 3 * C, ptr(o), _CODE[o.dtype], S,
                                        H * W, C, nh, 1.0 / math.sqrt(C // nh), stream()), "bg_small_attn")
        return ops.linear(o, pk.w, pk.b, out_dtype=torch.float32, add=x, n_valid=C)

    def _chunk(self, n, per_sample_bytes):
        return max(1, min(n, self.budget // max(1, per_sample_bytes)))


    def _decode_chunk_AutoencoderKLFastDecode(self, z_cl, dt):
        """z_cl: channels-last fp32 [S,4,4,latent] -> [S,32,32,out]."""
        P = self._pack(dt)
        d = self.decoder
        S = z_cl.shape[0]
        shape = (S, z_cl.shape[1], z_cl.shape[2], self.latent)
        x, shape = self._conv(z_cl, shape, P["pq"], 1, 1)
        x, shape = self._conv(x, shape, P["in"], 3, 3)
        x, shape = self._resnet(x, shape, P, "m0", d.mid_block.resnets[0])
        x = self._attn2d(x, shape, P, "ma", d.mid_block.attentions[0])
        x, shape = self._resnet(x, shape, P, "m1", d.mid_block.resnets[1])
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                x, shape = self._resnet(x, shape, P, f"u{bi}r{ri}", r)
            if hasattr(blk, "upsamplers"):
                x, shape = self._conv(x, shape, P[f"u{bi}up"], 3, 3, up=1)
        x, shape = self._conv(x, shape, P["out"], 3, 3, norm=d.conv_norm_out, act=ACT_SILU)
        return x.reshape(shape)

    def _decode_chunk_AutoencoderKL1DFastDecode(self, z_cl, dt):
        P = self._pack(dt)
        d = self.decoder
        S, L = z_cl.shape[0], z_cl.shape[1]
        shape = (S, 1, L, self.latent)
        x, shape = self._conv(z_cl, shape, P["pq"], 1, 1)
        x, shape = self._conv(x, shape, P["in"], 1, 3)
        for i in range(6):
            x, shape = self._resconv(x, shape, P, f"m{i}", d.mid_block.resnets[i])
            x = self._attn1d(x, shape, P, f"a{i}", d.mid_block.attentions[i])
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                x, shape = self._resconv(x, shape, P, f"u{bi}r{ri}", r)
            S_, _, L_, C = shape
            y = torch.empty(S_ * 2 * L_, C, device=x.device, dtype=torch.float32)
            check(_lib.load().bg_upsample1d_cubic(ptr(x), ptr(y), S_, L_, C, stream()), "bg_upsample1d_cubic")
            x, shape = y, (S_, 1, 2 * L_, C)
        x, shape = self._conv(x, shape, P["out"], 1, 3, norm=d.conv_norm_out, act=ACT_SILU)
        return x.reshape(shape[0], shape[2], shape[3])

    def _encode_chunk_AutoencoderKLFastEncode(self, x_cl, dt):
        P, e = self._pack(dt), self.encoder
        S = x_cl.shape[0]
        shape = (S, x_cl.shape[1], x_cl.shape[2], self.in_ch)
        x, shape = self._conv(x_cl, shape, P["in"], 3, 3)
        for bi, blk in enumerate(e.down_blocks):
            for ri, r in enumerate(blk.resnets):
                x, shape = self._resnet(x, shape, P, f"d{bi}r{ri}", r)
            if hasattr(blk, "downsamplers"):                      # Downsample2D: pad (0,1,0,1), conv 3x3 stride 2
                x, shape = self._conv(x, shape, P[f"d{bi}dn"], 3, 3, stride=2, pad=(0, 0))
        x, shape = self._resnet(x, shape, P, "m0", e.mid_block.resnets[0])
        x = self._attn2d(x, shape, P, "ma", e.mid_block.attentions[0])
        x, shape = self._resnet(x, shape, P, "m1", e.mid_block.resnets[1])
        x, shape = self._conv(x, shape, P["out"], 3, 3, norm=e.conv_norm_out, act=ACT_SILU)
        x, shape = self._conv(x, shape, P["q"], 1, 1)
        return x.reshape(shape)[..., : self.latent]               # DiagonalGaussianDistribution(moments).mode() = mean

    def _encode_chunk_AutoencoderKL1DFastEncode(self, x_cl, dt):
        P, e = self._pack(dt), self.encoder
        S, L = x_cl.shape[0], x_cl.shape[1]
        shape = (S, 1, L, self.in_ch)
        x, shape = self._conv(x_cl, shape, P["in"], 1, 3)
        for bi, blk in enumerate(e.down_blocks):
            S_, _, L_, C = shape
            y = torch.empty(S_ * (L_ // 2), C, device=x.device, dtype=torch.float32)
            check(_lib.load().bg_downsample1d_cubic(ptr(x), ptr(y), S_, L_, C, stream()), "bg_downsample1d_cubic")
            x, shape = y, (S_, 1, L_ // 2, C)
            for ri, r in enumerate(blk.resnets):
                x, shape = self._resconv(x, shape, P, f"d{bi}r{ri}", r)
        for i in range(6):
            x, shape = self._resconv(x, shape, P, f"m{i}", e.mid_block.resnets[i])
            x = self._attn1d(x, shape, P, f"a{i}", e.mid_block.attentions[i])
        x, shape = self._conv(x, shape, P["out"], 1, 3, norm=e.conv_norm_out, act=ACT_SILU)
        x, shape = self._conv(x, shape, P["q"], 1, 1)
        return x.reshape(shape[0], shape[2], shape[3])[..., : self.latent]



def stepwise(m, x, implicit_gemm=True, im2col_budget=IM2COL_BUDGET):
    """module.forward(x) through the step-by-step driver (chunked against `im2col_budget` like the round-1 path)."""
    st = _Steps(m, implicit_gemm, im2col_budget)
    dt = m._dtype()
    cls = type(m).__name__
    es = 4 if dt == torch.float32 else 2
    x = x.detach().to(torch.float32)
    two_d = cls in ("AutoencoderKLFastDecode", "AutoencoderKLFastEncode")
    x_cl = (x.permute(0, 2, 3, 1) if two_d else x.permute(0, 2, 1)).contiguous()
    n = x_cl.shape[0]
    if cls == "AutoencoderKLFastDecode":
        side = x_cl.shape[1] * 2 ** (len(m.block_out) - 1)
        worst, fn = side * side * 9 * max(m.block_out[0] * 2, m.block_out[0]) * es, st._decode_chunk_AutoencoderKLFastDecode
    elif cls == "AutoencoderKL1DFastDecode":
        length = x_cl.shape[1] * 2 ** len(m.block_out)
        worst, fn = length * 5 * m.block_out[-1] * es, st._decode_chunk_AutoencoderKL1DFastDecode
    elif cls == "AutoencoderKLFastEncode":
        worst, fn = x_cl.shape[1] * x_cl.shape[1] * 9 * m.block_out[0] * es, st._encode_chunk_AutoencoderKLFastEncode
    else:
        worst, fn = x_cl.shape[1] * 5 * m.block_out[-1] * es, st._encode_chunk_AutoencoderKL1DFastEncode
    step = st._chunk(n, worst)
    outs = [fn(x_cl[i:i + step].contiguous(), dt) for i in range(0, n, step)]
    y = torch.cat(outs) if len(outs) > 1 else outs[0]
    return (y.permute(0, 3, 1, 2) if two_d else y.permute(0, 2, 1)).contiguous()
