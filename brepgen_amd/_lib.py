"""ctypes binding of libbrepgen_hip.so -- the C ABI declared in include/brepgen_hip.h.

There is NO fallback: if the library is missing the import of any compute entry point raises.  The product
path never touches ``oracle/`` or torch math for the work the HIP kernels do.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbrepgen_hip.so")

BG_F32, BG_F16, BG_BF16 = 0, 1, 2
BG_ACT_NONE, BG_ACT_RELU = 0, 1
BG_SURFPOS, BG_SURFZ, BG_EDGEPOS, BG_EDGEZ = 0, 1, 2, 3
BG_MAX_LAYERS, BG_MAX_EMBEDS = 12, 5

vp, fp, i64p, u8p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers


class MlpWeights(C.Structure):
    _fields_ = [("w0", vp), ("b0", fp), ("ln_g", fp), ("ln_b", fp), ("w3", vp), ("b3", fp),
                ("k_in", C.c_int), ("n_out", C.c_int), ("n_out_pad", C.c_int), ("w0_dtype", C.c_int),
                ("w0_mfma", fp), ("w0_colsum", fp)]


class LayerWeights(C.Structure):
    _fields_ = [("ln1_g", fp), ("ln1_b", fp), ("ln2_g", fp), ("ln2_b", fp),
                ("w_qkv", vp), ("b_qkv", fp), ("w_o", vp), ("b_o", fp),
                ("w_1", vp), ("b_1", fp), ("w_2", vp), ("b_2", fp),
                ("qkv_colsum", fp), ("w1_colsum", fp), ("w_1f", vp), ("w_2f", vp)]


class DenoiserWeights(C.Structure):
    _fields_ = [("net", C.c_int), ("dtype", C.c_int), ("n_layer", C.c_int), ("_pad", C.c_int),
                ("layers", LayerWeights * BG_MAX_LAYERS),
                ("lnf_g", fp), ("lnf_b", fp),
                ("time_embed", MlpWeights), ("fc_out", MlpWeights),
                ("embed", MlpWeights * BG_MAX_EMBEDS),
                ("class_embed", fp), ("time_table", fp), ("time_table_rows", C.c_int), ("_pad3", C.c_int)]


class DenoiserInputs(C.Structure):
    _fields_ = [("B", C.c_int), ("S", C.c_int), ("E", C.c_int), ("n_timesteps", C.c_int),
                ("x", fp), ("surf_pos", fp), ("surf_z", fp), ("edge_pos", fp), ("mask", u8p),
                ("timesteps", i64p), ("class_label", i64p), ("cond_cache", fp),
                ("cond_cache_valid", C.c_int), ("varlen", C.c_int), ("rows_hint", C.c_double),
                ("pairs_hint", C.c_double), ("n_split", C.c_int), ("_pad2", C.c_int), ("rows_plan", C.c_double * 4)]


class GemmDesc(C.Structure):          # bg_gemm_desc
    _fields_ = [("a", vp), ("lda", C.c_int), ("w", vp), ("bias", fp), ("out", vp), ("ldc", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("N_pad", C.c_int), ("K", C.c_int),
                ("ab_dtype", C.c_int), ("out_dtype", C.c_int), ("act", C.c_int),
                ("add", fp), ("ld_add", C.c_int), ("add_div", C.c_int),
                ("add2", fp), ("ld_add2", C.c_int), ("add2_div", C.c_int),
                ("out_lo", vp), ("res_hi", vp), ("res_lo", vp), ("ld_res", C.c_int),
                ("stats_out", fp), ("stats_in", fp), ("colsum", fp), ("ln_eps", C.c_float)]


BG_E_ARG, BG_E_SHAPE, BG_E_WORKSPACE, BG_E_DTYPE, BG_E_ALIGN = -1, -2, -3, -4, -5      # enum bg_err


class ConvDesc(C.Structure):          # bg_conv_desc
    _fields_ = [("x", vp), ("S", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("up", C.c_int),
                ("w", vp), ("bias", fp), ("N", C.c_int),
                ("out", fp), ("ldc", C.c_int),
                ("add", fp), ("ld_add", C.c_int),
                ("dtype", C.c_int), ("zero_page", vp)]


class VaeOp(C.Structure):             # bg_vae_op
    _fields_ = [("op", C.c_int), ("src", C.c_int), ("dst", C.c_int), ("res", C.c_int),
                ("kh", C.c_int), ("kw", C.c_int), ("up", C.c_int), ("stride", C.c_int), ("pad_mode", C.c_int),
                ("n_out", C.c_int), ("n_pad", C.c_int), ("w_dtype", C.c_int),
                ("w", vp), ("bias", fp), ("gn_gamma", fp), ("gn_beta", fp), ("gn_groups", C.c_int),
                ("gn_eps", C.c_float), ("act", C.c_int), ("heads", C.c_int), ("scale", C.c_float),
                ("n_pad2", C.c_int), ("w2_dtype", C.c_int), ("w2", vp), ("bias2", fp)]


class ProfileRow(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("launches", C.c_int), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_SIGNATURES = {
    "bg_abi_version": (C.c_int, []),
    "bg_last_error": (C.c_char_p, []),
    "bg_sincos_embed": (C.c_int, [i64p, C.c_int, fp, vp]),
    "bg_layernorm_fwd": (C.c_int, [fp, fp, fp, vp, C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    "bg_gemm_bias_act_fwd": (C.c_int, [vp, C.c_int, vp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, vp]),
    "bg_gemm_ex_fwd": (C.c_int, [C.POINTER(GemmDesc), vp]),
    "bg_conv_gemm_fwd": (C.c_int, [C.POINTER(ConvDesc), vp]),
    "bg_vae_workspace_bytes": (C.c_size_t, [C.POINTER(VaeOp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bg_vae_run": (C.c_int, [C.POINTER(VaeOp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, vp, vp,
                             C.c_size_t, vp]),
    "bg_layernorm_split_fwd": (C.c_int, [vp, vp, fp, fp, vp, C.c_int, C.c_int, C.c_float, vp]),
    "bg_embed_ln_silu_fwd": (C.c_int, [fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, vp, C.c_int, C.c_float, vp]),
    "bg_ln_silu_out_fwd": (C.c_int, [vp, fp, fp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "bg_attn_fwd": (C.c_int, [vp, u8p, vp, C.c_int, C.c_int, C.c_int, vp]),
    "bg_ffn_fused_fwd": (C.c_int, [vp, vp, fp, vp, fp, fp, vp, fp, C.c_int, C.c_int, vp, C.c_int, C.c_float, vp]),
    "bg_qkv_attn_fwd": (C.c_int, [vp, vp, fp, fp, fp, u8p, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "bg_qkv_attn_paired_fwd": (C.c_int, [vp, vp, fp, fp, fp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "bg_compact_rows_paired": (C.c_int, [u8p, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "bg_attn_varlen_fwd": (C.c_int, [vp, u8p, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "bg_compact_rows": (C.c_int, [u8p, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "bg_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bg_denoiser_fwd": (C.c_int, [C.POINTER(DenoiserWeights), C.POINTER(DenoiserInputs), fp, vp, C.c_size_t, vp]),
    "bg_embed_mlp_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "bg_embed_mlp_fwd": (C.c_int, [C.POINTER(MlpWeights), C.c_int, vp, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_int,
                                   vp, C.c_size_t, vp]),
    "bg_encoder_layer_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "bg_encoder_layer_fwd": (C.c_int, [C.POINTER(LayerWeights), C.c_int, fp, u8p, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "bg_cfg_ddpm_step": (C.c_int, [fp, fp, C.c_float, fp, fp, fp, C.c_size_t] + [C.c_float] * 6 + [vp]),
    "bg_pndm_step": (C.c_int, [fp, fp, C.c_float, fp, fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float,
                               fp, fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, fp,
                               C.c_size_t, vp]),
    "bg_groupnorm_stats": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "bg_im2col": (C.c_int, [fp, vp] + [C.c_int] * 13 + [fp, fp, fp, C.c_int, C.c_int, fp, vp]),
    "bg_downsample1d_cubic": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "bg_upsample1d_cubic": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "bg_small_attn": (C.c_int, [fp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "bg_dedup_surfaces": (C.c_int, [fp, C.c_float, fp, u8p, C.c_int, C.c_int, vp]),
    "bg_dedup_edges": (C.c_int, [fp, u8p, C.c_float, u8p, C.c_int, C.c_int, C.c_int, vp]),
    "bg_profile_begin": (C.c_int, [C.c_int]),
    "bg_profile_end": (C.c_int, [C.POINTER(ProfileRow), C.c_int]),
    "bg_tune_set": (C.c_int, [C.c_int, C.c_int]),
    "bg_allgather": (C.c_int, [vp, vp, C.c_size_t, vp, vp]),
    "bg_slot_packing_applies": (C.c_int, [C.c_int] * 6),
    "bg_gemm_p256_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bg_add_noise": (C.c_int, [fp, fp, fp, fp, fp, C.c_int, C.c_size_t, vp]),
    "bg_chamfer_offset_fit": (C.c_int, [fp, fp, vp, C.c_int, C.c_int, C.c_int] + [C.c_double] * 5 + [fp, fp, fp, vp]),
    "bg_philox_randn": (C.c_int, [fp, C.c_longlong, C.c_int, C.c_ulonglong, C.c_uint, C.c_longlong, C.c_int, vp]),
    "bg_masked_mse": (C.c_int, [fp, fp, u8p, C.c_longlong, C.c_int, C.c_int, C.c_int, vp, fp, vp]),
}
EXPORTS = tuple(_SIGNATURES)

ABI_VERSION = 6          # BG_ABI_VERSION of include/brepgen_hip.h this binding was written against

_lib = None


class BrepgenHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (once).  Raises if it has not been built: no silent CPU/torch fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BrepgenHipError(
                f"{LIB_PATH} is missing -- build it with `python -m brepgen_amd.build` "
                "(hipcc --offload-arch=gfx950); brepgen_amd has no fallback path")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so does not export the ABI
            fn.restype, fn.argtypes = res, args
        if lib.bg_abi_version() != ABI_VERSION:
            raise BrepgenHipError("libbrepgen_hip.so ABI version mismatch")
        for kv in filter(None, os.environ.get("BG_TUNE", "").split(",")):     # A/B knobs, e.g. BG_TUNE="0=10,5=1"
            k, v = kv.split("=")
            lib.bg_tune_set(int(k), int(v))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().bg_last_error().decode(errors="replace")
        raise BrepgenHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device (or host, for CPU-side tests of argument checking) address of a contiguous tensor, or None."""
    if t is None:
        return None
    assert t.is_contiguous(), "libbrepgen_hip takes dense row-major tensors"
    return t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


class profile:
    """Context manager around bg_profile_begin/end: `with profile() as p: ...; p.rows` -> list of dicts."""

    def __init__(self, max_launches=20000):
        self.max_launches, self.rows = max_launches, []

    def __enter__(self):
        check(load().bg_profile_begin(self.max_launches), "bg_profile_begin")
        return self

    def __exit__(self, *exc):
        buf = (ProfileRow * 16)()
        n = load().bg_profile_end(buf, 16)
        if n < 0:
            check(n, "bg_profile_end")
        self.rows = [{"kernel": buf[i].kernel.decode(), "launches": buf[i].launches, "total_ms": buf[i].total_ms,
                      "flops": buf[i].flops, "bytes": buf[i].bytes} for i in range(n)]
        return False
