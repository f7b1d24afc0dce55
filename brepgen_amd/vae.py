"""Drop-in mirrors of the two VAE *decoders* BrepGen samples with, on the HIP path.

``AutoencoderKLFastDecode`` (network.py:948-1040, built at sample.py:72-84) and ``AutoencoderKL1DFastDecode``
(network.py:786-858, built at sample.py:86-99): same constructor keywords, same ``forward(z)`` -> decoded points,
and the diffusers checkpoint key layout, so ``load_state_dict(torch.load(vae.pt), strict=False)`` works as at
sample.py:83,98 (the file also holds ``encoder.*`` / ``quant_conv.*``, which are ignored).

Execution: channels-last fp32 activations; every convolution = ``bg_im2col`` (GroupNorm + SiLU/GELU and the nearest
x2 up-sampling folded into the gather) + the MFMA GEMM with bias / residual fused in its epilogue; mid-block
attention = one fused q|k|v GEMM + ``bg_small_attn`` + projection GEMM; ``Upsample1d("cubic")`` = ``bg_upsample1d_cubic``.
The nn.Module tree below only holds parameters.  Like the denoisers, bf16 operands inside autocast, exact fp32 outside.

A pass is ONE C call: each module compiles itself (once per dtype) into a flat ``bg_vae_op`` program and ``bg_vae_run``
enqueues every launch of it, chunking the batch against a caller-owned workspace.  (A step-by-step Python driver of the same
primitives -- the round-1 path -- is kept as a cross-check in tests/vae_stepwise.py, outside the product module.)
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import BG_BF16, BG_F16, BG_F32, check, ptr, stream

_CODE = {torch.bfloat16: BG_BF16, torch.float16: BG_F16, torch.float32: BG_F32}

CUBIC2 = [2 * v for v in (-0.01171875, -0.03515625, 0.11328125, 0.43359375, 0.43359375, 0.11328125, -0.03515625,
                          -0.01171875)]
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2


# --------------------------------------------------------------------------------------------------
# parameter containers (diffusers key layout)
# --------------------------------------------------------------------------------------------------
class _Resnet2D(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Attn2D(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _Up2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class _UpBlock2D(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if upsample:
            self.upsamplers = nn.ModuleList([_Up2D(cout)])


class _Mid2D(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn2D(c, groups)])
        self.resnets = nn.ModuleList([_Resnet2D(c, c, groups), _Resnet2D(c, c, groups)])


class _Decoder2D(nn.Module):
    def __init__(self, latent, out_ch, block_out, layers_per_block, groups):
        super().__init__()
        top = block_out[-1]
        self.conv_in = nn.Conv2d(latent, top, 3, padding=1)
        self.mid_block = _Mid2D(top, groups)
        rev = list(reversed(block_out))
        blocks, prev = [], rev[0]
        for i, ch in enumerate(rev):
            blocks.append(_UpBlock2D(prev, ch, layers_per_block + 1, groups, upsample=i != len(rev) - 1))
            prev = ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, block_out[0], eps=1e-6)
        self.conv_out = nn.Conv2d(block_out[0], out_ch, 3, padding=1)


class _ResConv(nn.Module):
    def __init__(self, cin, mid, cout):
        super().__init__()
        if cin != cout:
            self.conv_skip = nn.Conv1d(cin, cout, 1, bias=False)
        self.conv_1 = nn.Conv1d(cin, mid, 5, padding=2)
        self.group_norm_1 = nn.GroupNorm(1, mid)
        self.conv_2 = nn.Conv1d(mid, cout, 5, padding=2)
        self.group_norm_2 = nn.GroupNorm(1, cout)


class _Attn1D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = nn.GroupNorm(1, c)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c),
                                                            nn.Linear(c, c))


class _Cubic(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("kernel", torch.tensor(CUBIC2))


class _UpBlock1D(nn.Module):                       # network.py:30-48
    def __init__(self, cin, cout):
        super().__init__()
        self.resnets = nn.ModuleList([_ResConv(cin, cin, cin), _ResConv(cin, cin, cin), _ResConv(cin, cin, cout)])
        self.up = _Cubic()


class _Mid1D(nn.Module):                           # network.py:51-83
    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn1D(c) for _ in range(6)])
        self.resnets = nn.ModuleList([_ResConv(c, c, c) for _ in range(6)])


class _Decoder1D(nn.Module):                       # network.py:188-244
    def __init__(self, latent, out_ch, block_out, groups):
        super().__init__()
        top = block_out[-1]
        self.conv_in = nn.Conv1d(latent, top, 3, padding=1)
        self.mid_block = _Mid1D(top)
        rev = list(reversed(block_out))
        blocks, prev = [], rev[0]
        for ch in rev:
            blocks.append(_UpBlock1D(prev, ch))
            prev = ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, block_out[0], eps=1e-6)
        self.conv_out = nn.Conv1d(block_out[0], out_ch, 3, padding=1)


# --------------------------------------------------------------------------------------------------
# execution helpers
# --------------------------------------------------------------------------------------------------
class _Packed:
    """One convolution / linear as a GEMM: weight [n_pad, K] (compute dtype, or fp32 when K % 64 != 0), fp32 bias."""
    __slots__ = ("w", "b", "n", "k", "dtype")


def _pow2(v):
    return v > 0 and (v & (v - 1)) == 0


VOP_CONV, VOP_NORM_ACT_ADD, VOP_ATTN, VOP_UP1D, VOP_DOWN1D = 0, 1, 2, 3, 4
VAE_OUT = 255


class _Program:
    """A flat bg_vae_op program under construction: steps + a free-list slot allocator (slot 0 = the input)."""

    def __init__(self):
        self.steps, self.keep, self._free, self.n_slots = [], [], [], 1
        self.ops = None

    def new(self):
        if self._free:
            return self._free.pop()
        self.n_slots += 1
        return self.n_slots - 1

    def free(self, *slots):
        self._free.extend(s for s in dict.fromkeys(slots) if s > 0 and s not in self._free)

    def step(self, op, src, dst, res=-1):
        o = _lib.VaeOp()
        o.op, o.src, o.res = op, src, res
        o.dst = self.new() if dst is None else dst
        o.stride = 1
        self.steps.append(o)
        return o

    def norm(self, o, norm, act):
        g, b = norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous()
        self.keep += [g, b]
        o.gn_gamma, o.gn_beta, o.gn_groups, o.gn_eps, o.act = ptr(g), ptr(b), norm.num_groups, norm.eps, act

    def conv(self, src, pk, kh, kw, up=0, norm=None, act=ACT_NONE, res=-1, stride=1, pad_mode=0, dst=None, n_out=None):
        o = self.step(VOP_CONV, src, dst, res)
        o.kh, o.kw, o.up, o.stride, o.pad_mode = kh, kw, up, stride, pad_mode
        o.n_out, o.n_pad, o.w_dtype = pk.n if n_out is None else n_out, pk.w.shape[0], _CODE[pk.dtype]
        o.w, o.bias = ptr(pk.w), ptr(pk.b)
        if norm is not None:
            self.norm(o, norm, act)
        return o.dst

    def attn(self, src, qkv, proj, norm, heads, scale):
        o = self.step(VOP_ATTN, src, None)
        o.n_pad, o.w_dtype, o.w, o.bias = qkv.w.shape[0], _CODE[qkv.dtype], ptr(qkv.w), ptr(qkv.b)
        o.n_pad2, o.w2_dtype, o.w2, o.bias2 = proj.w.shape[0], _CODE[proj.dtype], ptr(proj.w), ptr(proj.b)
        o.heads, o.scale = heads, scale
        self.norm(o, norm, ACT_NONE)
        self.free(src)
        return o.dst

    def resnet2d(self, x, P, name, r):                 # diffusers ResnetBlock2D
        h = self.conv(x, P[name + "c1"], 3, 3, norm=r.norm1, act=ACT_SILU)
        sc = self.conv(x, P[name + "sc"], 1, 1) if name + "sc" in P else x
        out = self.conv(h, P[name + "c2"], 3, 3, norm=r.norm2, act=ACT_SILU, res=sc)
        self.free(h, x, sc)
        return out

    def resconv(self, x, P, name, r):                  # diffusers ResConvBlock (see _HipVAE._resconv)
        h1 = self.conv(x, P[name + "c1"], 1, 5)
        h2 = self.conv(h1, P[name + "c2"], 1, 5, norm=r.group_norm_1, act=ACT_GELU)
        self.free(h1)
        sk = self.conv(x, P[name + "sk"], 1, 1) if name + "sk" in P else x
        o = self.step(VOP_NORM_ACT_ADD, h2, None, sk)
        self.norm(o, r.group_norm_2, ACT_GELU)
        self.free(h2, x, sk)
        return o.dst

    def resample1d(self, x, op):
        o = self.step(op, x, None)
        self.free(x)
        return o.dst

    def finish(self):
        assert self.steps[-1].dst == VAE_OUT and self.n_slots <= 8
        self.ops = (_lib.VaeOp * len(self.steps))(*self.steps)
        return self


class _HipVAE(nn.Module):
    WS_BUDGET = 8 << 30            # bytes of bg_vae_run workspace (activation slots + scratch) per chunk of samples
    TWO_STREAMS_MIN = 2048         # samples from which a pass is cut into two concurrent halves (below: launch-bound, nothing to overlap)

    def __init__(self):
        super().__init__()
        self.compute_dtype = None
        # 16-bit modes: the 3x3 / k5 convolutions run as IMPLICIT GEMMs inside bg_vae_run: GroupNorm + activation + cast in one
        # elementwise pass, then the GEMM gathers the window itself -- the kh*kw-fold im2col matrix is never written (fp32 and
        # tiny batches take the materialised im2col path, bit-identical).  The step-by-step Python driver of the same
        # primitives that cross-checks the program lives in tests/vae_stepwise.py.
        self._packs = {}
        self._programs = {}
        self._zero = None
        self._ws = {}                  # (device, stream) -> workspace kept between passes (a fresh multi-GiB hipMalloc costs more than the pass)
        self.two_streams = True        # large batches: two halves of a pass in flight on forked streams (bit-identical; _run)
        self._side = None

    def _apply(self, fn, *a, **k):
        self._packs, self._programs, self._ws = {}, {}, {}
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._packs, self._programs = {}, {}
        return super().load_state_dict(*a, **k)

    def __setattr__(self, name, value):
        # switches of earlier versions: an nn.Module would accept the assignment silently and nothing would change
        if name in ("executor", "implicit_gemm"):
            raise AttributeError(f"{type(self).__name__}.{name} no longer exists: every pass is one bg_vae_run program; the "
                                 "step-by-step driver of the same kernels is tests/vae_stepwise.py")
        super().__setattr__(name, value)

    def release_workspace(self):
        """Drop the cached bg_vae_run workspaces (up to WS_BUDGET bytes per (device, stream) this module has decoded on)."""
        self._ws = {}

    def _zero_page(self, device):
        if self._zero is None or self._zero.device != device:
            self._zero = torch.zeros(1 << 16, dtype=torch.uint8, device=device)  # >= 2 * C bytes (one pixel of zeros) for any C <= 32768
        return self._zero

    def _run(self, x_cl, out_shape, dt):
        """bg_vae_run over the whole batch x_cl [n, (H,) W, C] -> [n, *out_shape]; chunks sized to WS_BUDGET.  Large batches are cut into
        two halves that run CONCURRENTLY -- the first on the caller's stream, the second on a forked helper stream, joined before the
        call returns (`two_streams`; each stream has its own workspace): a pass alternates MFMA-bound convolutions with HBM-bound
        GroupNorm / activation passes and tile-round tails, and two of them in flight fill each other's gaps, exactly like the sample
        groups of the denoisers (n_split).  Samples are independent and chunk boundaries do not change a sample's bits (tests), so
        the result is bit-identical."""
        if dt not in self._programs:
            self._programs[dt] = self._program(_Program(), self._pack(dt)).finish()
        n = x_cl.shape[0]
        out = torch.empty(n, *out_shape, device=x_cl.device, dtype=torch.float32)
        self._zero_page(x_cl.device)                                    # (created on the caller's stream, BEFORE the fork below orders the helper behind it)
        if self.two_streams and n >= self.TWO_STREAMS_MIN and not torch.cuda.is_current_stream_capturing():
            h = (n // 2 + 63) // 64 * 64                                # (whole 64-sample groups per half)
            cur = torch.cuda.current_stream(x_cl.device)
            side = self._side_stream(x_cl.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._run_into(x_cl[h:], out[h:], dt)
            self._run_into(x_cl[:h], out[:h], dt)
            cur.wait_stream(side)
        else:
            self._run_into(x_cl, out, dt)
        return out

    def _side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _run_into(self, x_cl, out, dt):
        pg, lib = self._programs[dt], _lib.load()
        n = x_cl.shape[0]
        h, w, c = (1, *x_cl.shape[1:]) if x_cl.dim() == 3 else x_cl.shape[1:]
        size = lambda n_, chunk: lib.bg_vae_workspace_bytes(pg.ops, len(pg.steps), pg.n_slots, h, w, c, n_, chunk)
        ref = min(n, 4096)
        per_sample = max(1, size(ref, ref) // ref)
        chunk = max(1, min(n, self.WS_BUDGET // per_sample))
        need = size(n, chunk)
        if need == 0:
            raise _lib.BrepgenHipError("bg_vae_workspace_bytes: malformed VAE program")
        key = (x_cl.device, stream())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            self._ws[key] = None                                     # release the smaller one first
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=x_cl.device)
        check(lib.bg_vae_run(pg.ops, len(pg.steps), pg.n_slots, h, w, c, ptr(x_cl), n, chunk, ptr(out),
                             ptr(self._zero_page(x_cl.device)), ptr(ws), ws.numel(), stream()), "bg_vae_run")

    def _dtype(self):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype('cuda')
            return dt if dt in (torch.bfloat16, torch.float16) else torch.bfloat16
        return torch.float32

    # ---- packing ----
    @staticmethod
    def _pack_gemm(weight2d, bias, dt, pad16=64):
        n, k = weight2d.shape
        p = _Packed()
        use = dt if (dt == torch.float32 or k % 64 == 0) else torch.float32
        w = weight2d.detach().to(torch.float32)
        pad = 1 if use == torch.float32 else pad16
        if n % pad:
            w = torch.cat([w, w.new_zeros((-n) % pad, k)])
        p.w = w.to(use).contiguous()
        b = torch.zeros(n, device=w.device) if bias is None else bias.detach().to(torch.float32)
        if b.numel() % pad:
            b = torch.cat([b, b.new_zeros((-b.numel()) % pad)])
        p.b, p.n, p.k, p.dtype = b.contiguous(), n, k, use
        return p

    def _pack_conv(self, conv, dt, pad16=64):
        """pad16: rows the 16-bit weight matrix is zero-padded to a multiple of.  128 for a NARROW windowed convolution (conv_out:
        3 output channels): the implicit GEMM then takes it as one 128-column tile that stores only the real columns, instead of
        materialising the 9-fold / 3-fold im2col matrix of the largest activation of the pass for the generic kernel."""
        w = conv.weight.detach()
        if w.dim() == 3:                               # Conv1d [Cout, Cin, k] -> [Cout, k*Cin] (tap-major)
            w2 = w.permute(0, 2, 1).reshape(w.shape[0], -1)
        else:                                          # Conv2d [Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin]
            w2 = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return self._pack_gemm(w2, conv.bias, dt, pad16)

    # ---- primitive steps on channels-last fp32 tensors [S, H, W, C] ----
    def _pack_resnet2d(self, P, name, r, dt):
        P[name + "c1"], P[name + "c2"] = self._pack_conv(r.conv1, dt), self._pack_conv(r.conv2, dt)
        if hasattr(r, "conv_shortcut"):
            P[name + "sc"] = self._pack_conv(r.conv_shortcut, dt)

    def _pack_attn2d(self, P, key, at, dt):
        P[key + "qkv"] = self._pack_gemm(torch.cat([at.to_q.weight, at.to_k.weight, at.to_v.weight]),
                                         torch.cat([at.to_q.bias, at.to_k.bias, at.to_v.bias]), dt)
        P[key + "proj"] = self._pack_gemm(at.to_out[0].weight, at.to_out[0].bias, dt)

    def _pack_resconv(self, P, name, r, dt):
        P[name + "c1"], P[name + "c2"] = self._pack_conv(r.conv_1, dt), self._pack_conv(r.conv_2, dt)
        if hasattr(r, "conv_skip"):
            P[name + "sk"] = self._pack_conv(r.conv_skip, dt)

    def _pack_attn1d(self, P, key, at, dt):
        P[key + "qkv"] = self._pack_gemm(torch.cat([at.query.weight, at.key.weight, at.value.weight]),
                                         torch.cat([at.query.bias, at.key.bias, at.value.bias]), dt)
        P[key + "proj"] = self._pack_gemm(at.proj_attn.weight, at.proj_attn.bias, dt)

class AutoencoderKLFastDecode(_HipVAE):
    """Surface-VAE decoder: z [F,3,4,4] -> points [F,3,32,32]  (network.py:948-1040)."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        if act_fn != "silu":
            raise NotImplementedError("BrepGen builds its VAEs with act_fn='silu'")
        self.block_out = tuple(block_out_channels)
        self.groups, self.latent, self.out_ch = norm_num_groups, latent_channels, out_channels
        self.decoder = _Decoder2D(latent_channels, out_channels, self.block_out, layers_per_block, norm_num_groups)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def _pack(self, dt):
        if dt in self._packs:
            return self._packs[dt]
        P = {}
        d = self.decoder
        P["pq"] = self._pack_conv(self.post_quant_conv, dt)
        P["in"] = self._pack_conv(d.conv_in, dt)
        self._pack_resnet2d(P, "m0", d.mid_block.resnets[0], dt)
        self._pack_resnet2d(P, "m1", d.mid_block.resnets[1], dt)
        self._pack_attn2d(P, "ma", d.mid_block.attentions[0], dt)
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                self._pack_resnet2d(P, f"u{bi}r{ri}", r, dt)
            if hasattr(blk, "upsamplers"):
                P[f"u{bi}up"] = self._pack_conv(blk.upsamplers[0].conv, dt)
        P["out"] = self._pack_conv(d.conv_out, dt, pad16=128)
        self._packs[dt] = P
        return P

    def _program(self, pg, P):
        d = self.decoder
        x = pg.conv(0, P["pq"], 1, 1)
        x2 = pg.conv(x, P["in"], 3, 3)
        pg.free(x)
        x = pg.resnet2d(x2, P, "m0", d.mid_block.resnets[0])
        x = pg.attn(x, P["maqkv"], P["maproj"], d.mid_block.attentions[0].group_norm, 1, 1.0 / math.sqrt(self.block_out[-1]))
        x = pg.resnet2d(x, P, "m1", d.mid_block.resnets[1])
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                x = pg.resnet2d(x, P, f"u{bi}r{ri}", r)
            if hasattr(blk, "upsamplers"):
                x2 = pg.conv(x, P[f"u{bi}up"], 3, 3, up=1)
                pg.free(x)
                x = x2
        pg.conv(x, P["out"], 3, 3, norm=d.conv_norm_out, act=ACT_SILU, dst=VAE_OUT)
        return pg

    def forward(self, z, return_dict=True, generator=None):
        if not z.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE decode runs on the MI355X only (tensor on {z.device})")
        z_cl = z.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        return self._decode_cl(z_cl).permute(0, 3, 1, 2).contiguous()

    def _decode_cl(self, z_cl):
        """Channels-last latents [F,4,4,3] -> channels-last point grids [F,32,32,3] (the layout the kernels use)."""
        dt = self._dtype()
        n = z_cl.shape[0]
        side = z_cl.shape[1] * 2 ** (len(self.block_out) - 1)
        return self._run(z_cl, (side, side, self.out_ch), dt)

    def decode_tokens(self, surfZ):
        """Token-layout latents [..., 16*3] (position-major, channel-minor: what SurfZNet denoises) -> point grids
        [..., 32, 32, 3].  Equals sample.py:289-290's `vae(z.unflatten(-1,(16,3)).flatten(0,1).permute(0,2,1)
        .unflatten(-1,(4,4))).permute(0,2,3,1).unflatten(0,(B,S))` without the two NCHW round trips: the token
        layout already is the channels-last layout."""
        if not surfZ.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE decode runs on the MI355X only (tensor on {surfZ.device})")
        lead = surfZ.shape[:-1]
        if surfZ.numel() == 0:                                       # a rank that owns no sample of a sharded batch
            return surfZ.new_zeros((*lead, *self._decode_cl_shape()), dtype=torch.float32)
        z_cl = surfZ.detach().to(torch.float32).reshape(-1, 4, 4, self.latent).contiguous()
        return self._decode_cl(z_cl).reshape(*lead, *self._decode_cl_shape())

    def _decode_cl_shape(self):
        side = 4 * 2 ** (len(self.block_out) - 1)
        return (side, side, self.out_ch)


class AutoencoderKL1DFastDecode(_HipVAE):
    """Edge-VAE decoder: z [G,3,4] -> points [G,3,32]  (network.py:786-858)."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215):
        super().__init__()
        self.block_out = tuple(block_out_channels)
        self.groups, self.latent, self.out_ch = norm_num_groups, latent_channels, out_channels
        self.decoder = _Decoder1D(latent_channels, out_channels, self.block_out, norm_num_groups)
        self.post_quant_conv = nn.Conv1d(latent_channels, latent_channels, 1)

    def _pack(self, dt):
        if dt in self._packs:
            return self._packs[dt]
        P = {}
        d = self.decoder
        P["pq"] = self._pack_conv(self.post_quant_conv, dt)
        P["in"] = self._pack_conv(d.conv_in, dt)
        for i in range(6):
            self._pack_resconv(P, f"m{i}", d.mid_block.resnets[i], dt)
            self._pack_attn1d(P, f"a{i}", d.mid_block.attentions[i], dt)
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                self._pack_resconv(P, f"u{bi}r{ri}", r, dt)
        P["out"] = self._pack_conv(d.conv_out, dt, pad16=128)
        self._packs[dt] = P
        return P

    def _program(self, pg, P):
        d = self.decoder
        x = pg.conv(0, P["pq"], 1, 1)
        x2 = pg.conv(x, P["in"], 1, 3)
        pg.free(x)
        x = x2
        c = self.block_out[-1]
        for i in range(6):
            x = pg.resconv(x, P, f"m{i}", d.mid_block.resnets[i])
            x = pg.attn(x, P[f"a{i}qkv"], P[f"a{i}proj"], d.mid_block.attentions[i].group_norm, c // 32, 1.0 / math.sqrt(32))
        for bi, blk in enumerate(d.up_blocks):
            for ri, r in enumerate(blk.resnets):
                x = pg.resconv(x, P, f"u{bi}r{ri}", r)
            x = pg.resample1d(x, VOP_UP1D)
        pg.conv(x, P["out"], 1, 3, norm=d.conv_norm_out, act=ACT_SILU, dst=VAE_OUT)
        return pg

    def forward(self, z, return_dict=True):
        if not z.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE decode runs on the MI355X only (tensor on {z.device})")
        z_cl = z.detach().to(torch.float32).permute(0, 2, 1).contiguous()          # [G, L, 3]
        return self._decode_cl(z_cl).permute(0, 2, 1).contiguous()

    def _decode_cl(self, z_cl):
        """Channels-last latents [G,4,3] -> channels-last polylines [G,32,3]."""
        dt = self._dtype()
        n = z_cl.shape[0]
        length = z_cl.shape[1] * 2 ** len(self.block_out)
        return self._run(z_cl, (length, self.out_ch), dt)

    def decode_tokens(self, edgeZ):
        """Token-layout latents [..., 4*3] (the first 12 of EdgeZNet's 18 channels) -> polylines [..., 32, 3]; equals
        sample.py:293-294's `vae(z.unflatten(-1,(4,3)).reshape(-1,4,3).permute(0,2,1)).permute(0,2,1).reshape(B,S,E,32,3)`."""
        if not edgeZ.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE decode runs on the MI355X only (tensor on {edgeZ.device})")
        lead = edgeZ.shape[:-1]
        if edgeZ.numel() == 0:
            return edgeZ.new_zeros((*lead, 4 * 2 ** len(self.block_out), self.out_ch), dtype=torch.float32)
        z_cl = edgeZ.detach().to(torch.float32).reshape(-1, 4, self.latent).contiguous()
        out = self._decode_cl(z_cl)
        return out.reshape(*lead, out.shape[1], out.shape[2])


# --------------------------------------------------------------------------------------------------
# encoders (training-time API surface of the path: trainer.py:521,925 call FastEncode under no_grad)
# --------------------------------------------------------------------------------------------------
class _Down2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)


class _DownBlock2D(nn.Module):
    def __init__(self, cin, cout, n, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet2D(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if downsample:
            self.downsamplers = nn.ModuleList([_Down2D(cout)])


class _Encoder2D(nn.Module):
    def __init__(self, in_ch, latent, block_out, layers_per_block, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, block_out[0], 3, padding=1)
        blocks, prev = [], block_out[0]
        for i, ch in enumerate(block_out):
            blocks.append(_DownBlock2D(prev, ch, layers_per_block, groups, downsample=i != len(block_out) - 1))
            prev = ch
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid2D(block_out[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, block_out[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(block_out[-1], 2 * latent, 3, padding=1)


class _CubicDown(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("kernel", torch.tensor(CUBIC2) / 2)


class _DownBlock1D(nn.Module):                     # diffusers DownBlock1D(out_channels, in_channels)
    def __init__(self, cin, cout):
        super().__init__()
        self.down = _CubicDown()
        self.resnets = nn.ModuleList([_ResConv(cin, cout, cout), _ResConv(cout, cout, cout), _ResConv(cout, cout, cout)])


class _Encoder1D(nn.Module):                       # network.py:86-185
    def __init__(self, in_ch, latent, block_out, groups):
        super().__init__()
        self.conv_in = nn.Conv1d(in_ch, block_out[0], 3, padding=1)
        blocks, prev = [], block_out[0]
        for ch in block_out:
            blocks.append(_DownBlock1D(prev, ch))
            prev = ch
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid1D(block_out[-1])
        self.conv_norm_out = nn.GroupNorm(groups, block_out[-1], eps=1e-6)
        self.conv_out = nn.Conv1d(block_out[-1], 2 * latent, 3, padding=1)


class AutoencoderKLFastEncode(_HipVAE):
    """Surface-VAE encoder: points [F,3,32,32] -> posterior mode [F,3,4,4]  (network.py:861-945)."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.block_out, self.latent, self.in_ch = tuple(block_out_channels), latent_channels, in_channels
        self.encoder = _Encoder2D(in_channels, latent_channels, self.block_out, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)

    def _pack(self, dt):
        if dt in self._packs:
            return self._packs[dt]
        P, e = {}, self.encoder
        P["in"] = self._pack_conv(e.conv_in, dt)
        for bi, blk in enumerate(e.down_blocks):
            for ri, r in enumerate(blk.resnets):
                self._pack_resnet2d(P, f"d{bi}r{ri}", r, dt)
            if hasattr(blk, "downsamplers"):
                P[f"d{bi}dn"] = self._pack_conv(blk.downsamplers[0].conv, dt)
        self._pack_resnet2d(P, "m0", e.mid_block.resnets[0], dt)
        self._pack_resnet2d(P, "m1", e.mid_block.resnets[1], dt)
        self._pack_attn2d(P, "ma", e.mid_block.attentions[0], dt)
        P["out"] = self._pack_conv(e.conv_out, dt, pad16=128)
        P["q"] = self._pack_conv(self.quant_conv, dt)
        self._packs[dt] = P
        return P

    def _program(self, pg, P):
        e = self.encoder
        x = pg.conv(0, P["in"], 3, 3)
        for bi, blk in enumerate(e.down_blocks):
            for ri, r in enumerate(blk.resnets):
                x = pg.resnet2d(x, P, f"d{bi}r{ri}", r)
            if hasattr(blk, "downsamplers"):
                x2 = pg.conv(x, P[f"d{bi}dn"], 3, 3, stride=2, pad_mode=1)
                pg.free(x)
                x = x2
        x = pg.resnet2d(x, P, "m0", e.mid_block.resnets[0])
        x = pg.attn(x, P["maqkv"], P["maproj"], e.mid_block.attentions[0].group_norm, 1, 1.0 / math.sqrt(self.block_out[-1]))
        x = pg.resnet2d(x, P, "m1", e.mid_block.resnets[1])
        x2 = pg.conv(x, P["out"], 3, 3, norm=e.conv_norm_out, act=ACT_SILU)
        pg.free(x)
        pg.conv(x2, P["q"], 1, 1, dst=VAE_OUT, n_out=self.latent)       # DiagonalGaussianDistribution(moments).mode() = mean
        return pg

    def forward(self, x, return_dict=True):
        if not x.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE encode runs on the MI355X only (tensor on {x.device})")
        x_cl = x.detach().to(torch.float32).permute(0, 2, 3, 1).contiguous()
        return self._encode_cl(x_cl).permute(0, 3, 1, 2).contiguous()

    def _encode_cl(self, x_cl):
        """Channels-last point grids [F,32,32,3] -> channels-last latent modes [F,4,4,3]."""
        dt = self._dtype()
        n, side = x_cl.shape[0], x_cl.shape[1]
        lat = side >> (len(self.block_out) - 1)
        return self._run(x_cl, (lat, lat, self.latent), dt)

    def encode_tokens(self, surfPnt):
        """Point grids [..., 32, 32, 3] (the datasets' layout) -> token-layout latents [..., 48]; equals trainer.py:519-524
        `vae(p.flatten(0,1).permute(0,3,1,2)).unflatten(0,(B,-1)).flatten(-2,-1).permute(0,1,3,2).flatten(-2,-1)` without
        the NCHW round trips."""
        if not surfPnt.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE encode runs on the MI355X only (tensor on {surfPnt.device})")
        lead = surfPnt.shape[:-3]
        x_cl = surfPnt.detach().to(torch.float32).reshape(-1, *surfPnt.shape[-3:]).contiguous()
        z = self._encode_cl(x_cl)                                  # [F, 4, 4, latent]
        return z.reshape(*lead, z.shape[1] * z.shape[2] * z.shape[3])


class AutoencoderKL1DFastEncode(_HipVAE):
    """Edge-VAE encoder: points [G,3,32] -> posterior mode [G,3,4]  (network.py:690-783)."""

    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",),
                 up_block_types=("UpDecoderBlock2D",), block_out_channels=(64,), layers_per_block=1, act_fn="silu",
                 latent_channels=4, norm_num_groups=32, sample_size=32, scaling_factor=0.18215):
        super().__init__()
        self.block_out, self.latent, self.in_ch = tuple(block_out_channels), latent_channels, in_channels
        self.encoder = _Encoder1D(in_channels, latent_channels, self.block_out, norm_num_groups)
        self.quant_conv = nn.Conv1d(2 * latent_channels, 2 * latent_channels, 1)

    def _pack(self, dt):
        if dt in self._packs:
            return self._packs[dt]
        P, e = {}, self.encoder
        P["in"] = self._pack_conv(e.conv_in, dt)
        for bi, blk in enumerate(e.down_blocks):
            for ri, r in enumerate(blk.resnets):
                self._pack_resconv(P, f"d{bi}r{ri}", r, dt)
        for i in range(6):
            self._pack_resconv(P, f"m{i}", e.mid_block.resnets[i], dt)
            self._pack_attn1d(P, f"a{i}", e.mid_block.attentions[i], dt)
        P["out"] = self._pack_conv(e.conv_out, dt, pad16=128)
        P["q"] = self._pack_conv(self.quant_conv, dt)
        self._packs[dt] = P
        return P

    def _program(self, pg, P):
        e = self.encoder
        x = pg.conv(0, P["in"], 1, 3)
        for bi, blk in enumerate(e.down_blocks):
            x = pg.resample1d(x, VOP_DOWN1D)
            for ri, r in enumerate(blk.resnets):
                x = pg.resconv(x, P, f"d{bi}r{ri}", r)
        c = self.block_out[-1]
        for i in range(6):
            x = pg.resconv(x, P, f"m{i}", e.mid_block.resnets[i])
            x = pg.attn(x, P[f"a{i}qkv"], P[f"a{i}proj"], e.mid_block.attentions[i].group_norm, c // 32, 1.0 / math.sqrt(32))
        x2 = pg.conv(x, P["out"], 1, 3, norm=e.conv_norm_out, act=ACT_SILU)
        pg.free(x)
        pg.conv(x2, P["q"], 1, 1, dst=VAE_OUT, n_out=self.latent)
        return pg

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        if not sample.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE encode runs on the MI355X only (tensor on {sample.device})")
        x_cl = sample.detach().to(torch.float32).permute(0, 2, 1).contiguous()
        return self._encode_cl(x_cl).permute(0, 2, 1).contiguous()

    def _encode_cl(self, x_cl):
        """Channels-last polylines [G,32,3] -> channels-last latent modes [G,4,3]."""
        dt = self._dtype()
        n = x_cl.shape[0]
        return self._run(x_cl, (x_cl.shape[1] >> len(self.block_out), self.latent), dt)

    def encode_tokens(self, edgePnt):
        """Polylines [..., 32, 3] -> token-layout latents [..., 12]; equals trainer.py:924-929
        `vae(p.flatten(0,1).flatten(0,1).permute(0,2,1)) ... .permute(0,1,2,4,3).flatten(-2,-1)`."""
        if not edgePnt.is_cuda:
            raise _lib.BrepgenHipError(f"brepgen_amd VAE encode runs on the MI355X only (tensor on {edgePnt.device})")
        lead = edgePnt.shape[:-2]
        x_cl = edgePnt.detach().to(torch.float32).reshape(-1, *edgePnt.shape[-2:]).contiguous()
        z = self._encode_cl(x_cl)                                  # [G, 4, latent]
        return z.reshape(*lead, z.shape[1] * z.shape[2])
