"""Test / tool infrastructure: thin tensor-level wrappers over the C ABI (one function per exported kernel entry point), so that
the parity tests and the benchmarking tools can call single kernels.  The product modules (brepgen_amd/*.py) bind the entry points
they use themselves; nothing in the package imports this file.  Every function enqueues on the current torch HIP stream and
returns torch tensors that own the output memory.  No torch math here -- only allocation and pointers.
"""
import torch

from brepgen_amd import _lib
from brepgen_amd._lib import BG_ACT_NONE, BG_ACT_RELU, BG_BF16, BG_F16, BG_F32, check, ptr, stream  # noqa: F401

_DT = {torch.float32: BG_F32, torch.bfloat16: BG_BF16, torch.float16: BG_F16}


def bg_dtype(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"brepgen_amd supports float32, bfloat16 and float16 compute, not {dt}") from None


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.BrepgenHipError("brepgen_amd kernels run on the MI355X only: tensor is on "
                                       f"{t.device}; there is no CPU fallback")


def sincos_embed(timesteps):
    _need_cuda(timesteps)
    t = timesteps.reshape(-1).to(torch.int64).contiguous()
    out = torch.empty(t.numel(), 768, device=t.device, dtype=torch.float32)
    check(_lib.load().bg_sincos_embed(ptr(t), t.numel(), ptr(out), stream()), "bg_sincos_embed")
    return out


def layernorm(x, gamma, beta, out_dtype=torch.float32, eps=1e-5, silu=False):
    _need_cuda(x, gamma, beta)
    assert x.dtype == torch.float32 and x.shape[-1] == 768
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    M = x.numel() // 768
    check(_lib.load().bg_layernorm_fwd(ptr(x), ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(y),
                                       bg_dtype(out_dtype), M, eps, int(silu), stream()), "bg_layernorm_fwd")
    return y


def linear(a, w, bias=None, out_dtype=torch.float32, act=BG_ACT_NONE, add=None, add_div=1, n_valid=None, out=None):
    """out = act(a @ w.T + bias) + add[m // add_div]; a [M,K], w [N_pad,K] (same dtype), n_valid <= N_pad."""
    _need_cuda(a, w, bias, add)
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype and a.shape[1] == w.shape[1]
    a, w = a.contiguous(), w.contiguous()
    M, K = a.shape
    n_pad = w.shape[0]
    N = n_pad if n_valid is None else n_valid
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    ld_add = add.shape[-1] if add is not None else 0
    check(_lib.load().bg_gemm_bias_act_fwd(ptr(a), K, ptr(w), ptr(bias), ptr(out), out.shape[1], M, N, n_pad, K,
                                           bg_dtype(a.dtype), bg_dtype(out.dtype), act, ptr(add), ld_add, add_div,
                                           stream()), "bg_gemm_bias_act_fwd")
    return out


def linear_ex(a, w, bias=None, *, act=BG_ACT_NONE, out_dtype=None, add=None, add_div=1, add2=None, add2_div=1,
              split_out=False, res=None, want_stats=False, stats_in=None, colsum=None, ln_eps=1e-5, inplace=False):
    """bg_gemm_ex_fwd.  Returns a dict: out (or hi), lo (split_out), stats (want_stats: [N/64, M, 2]).

    res = (hi, lo) split residual rows added to the result; stats_in/colsum = LayerNorm fold on the A rows.
    inplace (split_out + res): the result planes overwrite res, as the encoder layers run out-proj / FFN2."""
    _need_cuda(a, w, bias, add, add2, stats_in, colsum)
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype and a.shape[1] == w.shape[1]
    a, w = a.contiguous(), w.contiguous()
    M, K = a.shape
    N = w.shape[0]
    odt = a.dtype if (split_out or out_dtype is None) else out_dtype
    out = res[0] if inplace else torch.empty(M, N, device=a.device, dtype=odt)
    d = _lib.GemmDesc()
    d.a, d.lda, d.w, d.bias, d.out, d.ldc = ptr(a), K, ptr(w), ptr(bias), ptr(out), N
    d.M, d.N, d.N_pad, d.K = M, N, N, K
    d.ab_dtype, d.out_dtype, d.act = bg_dtype(a.dtype), bg_dtype(odt), act
    d.add, d.ld_add, d.add_div = ptr(add), (add.shape[-1] if add is not None else 0), add_div
    d.add2, d.ld_add2, d.add2_div = ptr(add2), (add2.shape[-1] if add2 is not None else 0), add2_div
    r = {"out": out}
    if split_out:
        r["lo"] = res[1] if inplace else torch.empty_like(out)
        d.out_lo = ptr(r["lo"])
    if res is not None:
        d.res_hi, d.res_lo, d.ld_res = ptr(res[0]), ptr(res[1]), res[0].shape[-1]
    if want_stats:
        r["stats"] = torch.zeros(N // 64, M, 2, device=a.device, dtype=torch.float32)
        d.stats_out = ptr(r["stats"])
    if stats_in is not None:
        d.stats_in, d.colsum = ptr(stats_in.contiguous()), ptr(colsum.contiguous())
    d.ln_eps = ln_eps
    r["_keep"] = (a, w, bias, add, add2, res, stats_in, colsum)
    check(_lib.load().bg_gemm_ex_fwd(d, stream()), "bg_gemm_ex_fwd")
    return r


def layernorm_split(hi, lo, gamma, beta, eps=1e-5):
    """LayerNorm(768) of x = hi + lo (two 16-bit planes) -> same 16-bit dtype."""
    _need_cuda(hi, lo, gamma, beta)
    assert hi.dtype == lo.dtype and hi.shape == lo.shape and hi.shape[-1] == 768
    hi, lo = hi.contiguous(), lo.contiguous()
    y = torch.empty_like(hi)
    check(_lib.load().bg_layernorm_split_fwd(ptr(hi), ptr(lo), ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(y),
                                             bg_dtype(hi.dtype), hi.numel() // 768, eps, stream()),
          "bg_layernorm_split_fwd")
    return y


def embed_ln_silu(x, k, w0, b0, gamma, beta, out_dtype=torch.float32, eps=1e-5):
    """SiLU(LayerNorm(x[:, :k] @ w0.T + b0)) through the fused kernel; x fp32 [rows, lda >= k] (a column-offset view
    with unit column stride is fine), w0 fp32 [768, k]."""
    from brepgen_amd.network import mfma_operand_order
    _need_cuda(x, w0, b0, gamma, beta)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and w0.shape == (768, k)
    rows, lda = x.shape[0], x.stride(0)
    w0p = mfma_operand_order(w0.to(torch.float32))
    out = torch.empty(rows, 768, device=x.device, dtype=out_dtype)
    check(_lib.load().bg_embed_ln_silu_fwd(x.data_ptr(), lda, rows, k, ptr(w0p), ptr(b0.contiguous()),
                                           ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(out),
                                           bg_dtype(out_dtype), eps, stream()), "bg_embed_ln_silu_fwd")
    return out


def embed_mlp(mlp_weights, dtype, x, k=None, add=None, add_div=1):
    """bg_embed_mlp_fwd on one packed bg_mlp_weights (e.g. net._pack(dt)[0].embed[i]); x [rows, lda]; -> fp32 [rows, n_out]."""
    import ctypes as C
    _need_cuda(x, add)
    assert x.dim() == 2 and x.stride(1) == 1
    rows, lda = x.shape[0], x.stride(0)
    out = torch.empty(rows, mlp_weights.n_out, device=x.device, dtype=torch.float32)
    nbytes = _lib.load().bg_embed_mlp_scratch_bytes(rows, bg_dtype(dtype))
    scratch = torch.empty(nbytes + 256, device=x.device, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    check(_lib.load().bg_embed_mlp_fwd(C.byref(mlp_weights), bg_dtype(dtype), x.data_ptr(), lda, rows, ptr(out), out.shape[1],
                                       ptr(add), add.shape[-1] if add is not None else 0, add_div, base, nbytes, stream()),
          "bg_embed_mlp_fwd")
    return out


def encoder_layer(layer_weights, dtype, x, key_pad, B, N):
    """bg_encoder_layer_fwd: one pre-LN encoder layer on the fp32 stream x [B*N, 768]; returns the updated copy."""
    import ctypes as C
    _need_cuda(x, key_pad)
    assert x.dtype == torch.float32 and x.shape == (B * N, 768)
    x = x.contiguous().clone()
    kp = None
    if key_pad is not None:
        kp = key_pad.contiguous()
        kp = kp.view(torch.uint8) if kp.dtype == torch.bool else kp.to(torch.uint8)
    nbytes = _lib.load().bg_encoder_layer_scratch_bytes(B, N, bg_dtype(dtype))
    scratch = torch.empty(nbytes + 256, device=x.device, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    check(_lib.load().bg_encoder_layer_fwd(C.byref(layer_weights), bg_dtype(dtype), ptr(x), ptr(kp), B, N, base, nbytes,
                                           stream()), "bg_encoder_layer_fwd")
    return x


def qkv_attention(x_hi, w_qkv, bias, colsum, stats_in, B, N, want_qkv=False, ln_eps=1e-5, key_pad=None):
    """bg_qkv_attn_fwd: LayerNorm-fold QKV + attention in one launch -> out [B*N, 768] (and the q|k|v image, want_qkv)."""
    _need_cuda(x_hi, w_qkv, bias, colsum, stats_in, key_pad)
    assert x_hi.shape == (B * N, 768) and w_qkv.shape == (2304, 768) and stats_in.shape == (12, B * N, 2)
    x_hi, w_qkv, stats_in = x_hi.contiguous(), w_qkv.contiguous(), stats_in.contiguous()
    kp = None
    if key_pad is not None:
        kp = key_pad.contiguous()
        kp = kp.view(torch.uint8) if kp.dtype == torch.bool else kp.to(torch.uint8)
    out = torch.empty(B * N, 768, device=x_hi.device, dtype=x_hi.dtype)
    dbg = torch.zeros(B * N, 2304, device=x_hi.device, dtype=x_hi.dtype) if want_qkv else None
    check(_lib.load().bg_qkv_attn_fwd(ptr(x_hi), ptr(w_qkv), ptr(bias.contiguous()), ptr(colsum.contiguous()), ptr(stats_in), ptr(kp),
                                      ptr(out), ptr(dbg), B, N, bg_dtype(x_hi.dtype), ln_eps, stream()), "bg_qkv_attn_fwd")
    return (out, dbg) if want_qkv else out


def attention(qkv, key_pad, B, N):
    """qkv [B*N, 2304] (q pre-scaled by 1/8), key_pad bool/uint8 [B,N] or None -> [B*N, 768]."""
    _need_cuda(qkv, key_pad)
    qkv = qkv.contiguous()
    assert qkv.shape == (B * N, 2304)
    kp = None
    if key_pad is not None:
        kp = key_pad.contiguous()
        kp = kp.view(torch.uint8) if kp.dtype == torch.bool else kp.to(torch.uint8)
    out = torch.empty(B * N, 768, device=qkv.device, dtype=qkv.dtype)
    check(_lib.load().bg_attn_fwd(ptr(qkv), ptr(kp), ptr(out), B, N, bg_dtype(qkv.dtype), stream()), "bg_attn_fwd")
    return out


def ln_silu_out(t0, gamma, beta, w3, b3, n_out, eps=1e-5):
    """bg_ln_silu_out_fwd: W3 . SiLU(LayerNorm(t0)) + b3 in one launch; t0 [rows, 768] 16-bit, w3 [n_out_pad, 768] same dtype."""
    _need_cuda(t0, gamma, beta, w3, b3)
    assert t0.dim() == 2 and t0.shape[1] == 768 and w3.dtype == t0.dtype and w3.shape[1] == 768
    t0, w3 = t0.contiguous(), w3.contiguous()
    out = torch.empty(t0.shape[0], n_out, device=t0.device, dtype=torch.float32)
    check(_lib.load().bg_ln_silu_out_fwd(ptr(t0), ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(w3), ptr(b3.contiguous()),
                                         ptr(out), n_out, w3.shape[0], t0.shape[0], bg_dtype(t0.dtype), eps, stream()),
          "bg_ln_silu_out_fwd")
    return out


def ffn_fused(hi, lo, stats, w1, b1, colsum1, w2, b2, m_dev=None, ln_eps=1e-5):
    """bg_ffn_fused_fwd: FFN1 (LayerNorm fold, ReLU) + FFN2 (split residual + statistics) in one launch, IN PLACE on clones of
    (hi, lo, stats).  w1 [1024, 768] / w2 [768, 1024]: the folded 16-bit matrices in their row-major form (fragment order is made
    here, by the product's own packer).  Returns (hi, lo, stats)."""
    from brepgen_amd.network import ffn_fragment_order
    _need_cuda(hi, lo, stats, w1, b1, colsum1, w2, b2)
    hi, lo, stats = hi.clone(), lo.clone(), stats.clone()
    M = hi.shape[0]
    assert hi.shape == (M, 768) and lo.shape == hi.shape and stats.shape[0] == 12 and stats.shape[2] == 2
    w1f, w2f = ffn_fragment_order(w1.contiguous(), 4), ffn_fragment_order(w2.contiguous(), 3)
    check(_lib.load().bg_ffn_fused_fwd(ptr(hi), ptr(lo), ptr(stats), ptr(w1f), ptr(b1.contiguous()), ptr(colsum1.contiguous()), ptr(w2f),
                                       ptr(b2.contiguous()), M, stats.shape[1], ptr(m_dev), bg_dtype(hi.dtype), ln_eps, stream()),
          "bg_ffn_fused_fwd")
    return hi, lo, stats

