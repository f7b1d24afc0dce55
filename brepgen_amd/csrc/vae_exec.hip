// One C call per VAE pass: bg_vae_run walks a flat program of convolution-net steps (what AutoencoderKLFastDecode /
// AutoencoderKL1DFastDecode / ...FastEncode.forward do: network.py:690-1040 over the diffusers==0.27 blocks) and
// enqueues every launch of it on one stream -- no host allocation, no synchronisation, the caller owns the workspace.
// SURVEY.md section 8(b) names these entry points bg_vae2d_decode / bg_vae1d_decode; the two (and the encoders) are the
// same interpreter over different programs, so there is one entry point and the program says which network it is.
//
// A step works on activation SLOTS (fp32 channels-last [S, H, W, C], 1-D: H = 1; slot 0 = the input chunk, slot
// BG_VAE_OUT = the output chunk).  Steps:
//   CONV          (GroupNorm [+ SiLU / GELU] ->) conv kh x kw (nearest x2 up-sampling or stride 2 folded in) + bias (+ residual)
//                 as an implicit GEMM when the persistent kernel can take it (bg_conv_gemm_fwd), else im2col + GEMM
//   NORM_ACT_ADD  GroupNorm + activation + residual add (the tail of diffusers' ResConvBlock)
//   ATTN          GroupNorm -> q|k|v projection -> softmax(q k^T) v per sample and head -> output projection + residual
//   UP1D / DOWN1D diffusers' "cubic" 1-D resamplers
// Samples are independent, so the batch is processed in chunks sized to the workspace (bg_vae_workspace_bytes).
#include "bg_common.h"

namespace bg {

struct VShape { int H = 0, W = 0, C = 0; size_t elems() const { return (size_t)H * W * C; } };

static inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// would bg_conv_gemm_fwd take this convolution for a chunk of S samples?  (mirrors its argument checks + the >= 64 tiles rule)
static bool conv_implicit_ok(const bg_vae_op& o, const VShape& in, int S) {
    if (o.w_dtype == BG_F32 || o.kh * o.kw <= 1 || o.stride != 1 || o.pad_mode != 0) return false;
    const int Ho = in.H << o.up, Wo = in.W << o.up;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    // whole 128-column tiles, or a NARROW output (conv_out: 3 channels) whose weights the module padded to one 128-row tile
    const bool wide = o.n_out % 128 == 0 && o.n_pad == o.n_out, narrow = o.n_out < 128 && o.n_pad == 128 && o.res < 0;
    if (in.C % 64 != 0 || !pow2(in.C / 64) || !pow2(Ho) || !pow2(Wo) || !(wide || narrow)) return false;
    const long long rows = (long long)S * Ho * Wo;
    return rows < (1ll << 31) && ((rows + 127) / 128) * (o.n_pad / 128) >= 64;
}

static void conv_out_shape(const bg_vae_op& o, const VShape& in, VShape& out) {
    const int Hl = in.H << o.up, Wl = in.W << o.up;
    if (o.pad_mode == 0) {
        const int py = o.kh / 2, px = o.kw / 2;
        out.H = (Hl + 2 * py - o.kh) / o.stride + 1;
        out.W = (Wl + 2 * px - o.kw) / o.stride + 1;
    } else {                                                       // Downsample2D: F.pad(x, (0,1,0,1)), then stride-2 conv
        out.H = o.kh > 1 ? (Hl + 1 - o.kh) / o.stride + 1 : Hl;
        out.W = (Wl + 1 - o.kw) / o.stride + 1;
    }
    out.C = o.n_out;
}

struct VPlan {
    VShape slot[BG_VAE_MAX_SLOTS];
    VShape out;                       // shape written to BG_VAE_OUT
    size_t slot_elems = 0;            // per sample: largest activation (fp32 elements)
    size_t scratch_bytes = 0;         // per CHUNK of S samples (depends on which convolutions go implicit)
    bool ok = false;
};

// shape inference + workspace needs of one chunk of S samples
static VPlan plan_program(const bg_vae_op* ops, int n_ops, int n_slots, int in_h, int in_w, int in_c, int S) {
    VPlan p;
    auto bad = [&](int i, const char* what) {
        set_error("bg_vae program: step %d: %s", i, what);
        return p;
    };
    if (n_slots < 1 || n_slots > BG_VAE_MAX_SLOTS) return bad(-1, "n_slots out of range");
    p.slot[0] = VShape{in_h, in_w, in_c};
    auto shape_of = [&](int id, VShape& sh) -> bool {
        if (id < 0 || id >= n_slots || p.slot[id].C == 0) return false;
        sh = p.slot[id];
        return true;
    };
    for (int i = 0; i < n_ops; ++i) {
        const bg_vae_op& o = ops[i];
        VShape in, out;
        if (!shape_of(o.src, in)) return bad(i, "src slot undefined");
        size_t scr = 0;
        const size_t px = (size_t)S * in.H * in.W;
        const size_t stats = o.gn_gamma ? al256((size_t)S * o.gn_groups * 2 * 4) : 0;
        switch (o.op) {
            case BG_VOP_CONV: {
                conv_out_shape(o, in, out);
                const size_t es = o.w_dtype == BG_F32 ? 4 : 2;
                if (conv_implicit_ok(o, in, S)) scr = stats + al256(px * in.C * es);
                else scr = stats + al256((size_t)S * out.H * out.W * o.kh * o.kw * in.C * es);
                break;
            }
            case BG_VOP_NORM_ACT_ADD:
                out = in;
                scr = stats;
                break;
            case BG_VOP_ATTN:
                out = in;
                scr = stats + al256(px * in.C * (o.w_dtype == BG_F32 ? 4 : 2)) + al256(px * 3 * in.C * 4) +
                      al256(px * in.C * (o.w2_dtype == BG_F32 ? 4 : 2));
                break;
            case BG_VOP_UP1D: out = VShape{1, in.W * 2, in.C}; break;
            case BG_VOP_DOWN1D: out = VShape{1, in.W / 2, in.C}; break;
            default: return bad(i, "unknown opcode");
        }
        if (o.res >= 0) {
            VShape r;
            if (!shape_of(o.res, r) || r.H != out.H || r.W != out.W || r.C != out.C) return bad(i, "residual slot undefined or of another shape");
        }
        if (o.dst == BG_VAE_OUT) p.out = out;
        else if (o.dst > 0 && o.dst < n_slots && o.dst != o.src && o.dst != o.res) p.slot[o.dst] = out;
        else return bad(i, "dst must be a workspace slot (or BG_VAE_OUT) other than src / res");
        if (o.dst != BG_VAE_OUT && out.elems() > p.slot_elems) p.slot_elems = out.elems();
        if (scr > p.scratch_bytes) p.scratch_bytes = scr;
    }
    if (p.out.C == 0) return bad(n_ops - 1, "no step writes BG_VAE_OUT");
    p.ok = true;
    return p;
}

static size_t chunk_bytes(const VPlan& p, int n_slots, int S) {
    return (size_t)(n_slots - 1) * al256((size_t)S * p.slot_elems * 4) + p.scratch_bytes + 256;
}

}  // namespace bg

extern "C" size_t bg_vae_workspace_bytes(const bg_vae_op* ops, int n_ops, int n_slots, int in_h, int in_w, int in_c, int n, int chunk) {
    if (!ops || n_ops <= 0 || chunk <= 0 || n <= 0) return 0;
    // the full chunks and the tail chunk can choose differently between implicit GEMM and im2col: take the larger need
    const int full = n < chunk ? n : chunk, tail = n % full;
    size_t need = 0;
    for (int S : {full, tail}) {
        if (S <= 0) continue;
        const bg::VPlan p = bg::plan_program(ops, n_ops, n_slots, in_h, in_w, in_c, S);
        if (!p.ok) return 0;
        const size_t b = bg::chunk_bytes(p, n_slots, S);
        if (b > need) need = b;
    }
    return need;
}

extern "C" int bg_vae_run(const bg_vae_op* ops, int n_ops, int n_slots, int in_h, int in_w, int in_c, const float* x, int n,
                          int chunk, float* out, const void* zero_page, void* workspace, size_t workspace_bytes,
                          bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(ops && x && out && workspace && zero_page, BG_E_ARG, "bg_vae_run: null pointer");
    BG_REQUIRE(n >= 0 && chunk > 0 && n_ops > 0, BG_E_SHAPE, "bg_vae_run: bad sizes");
    BG_REQUIRE(((uintptr_t)workspace & 255) == 0, BG_E_ALIGN, "bg_vae_run: workspace must be 256-byte aligned");
    const size_t in_elems = (size_t)in_h * in_w * in_c;
    for (int s0 = 0; s0 < n; s0 += chunk) {
        const int S = n - s0 < chunk ? n - s0 : chunk;
        const VPlan p = plan_program(ops, n_ops, n_slots, in_h, in_w, in_c, S);
        if (!p.ok) return BG_E_ARG;                               // plan_program has said which step and why
        BG_REQUIRE(chunk_bytes(p, n_slots, S) <= workspace_bytes, BG_E_WORKSPACE, "bg_vae_run: workspace %zu < %zu bytes for a chunk of %d",
                   workspace_bytes, chunk_bytes(p, n_slots, S), S);
        unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
        const size_t slot_bytes = al256((size_t)S * p.slot_elems * 4);
        unsigned char* scratch = ws + (size_t)(n_slots - 1) * slot_bytes;
        float* out_chunk = out + (size_t)s0 * p.out.elems();
        auto slot_ptr = [&](int id) -> float* {
            if (id == 0) return const_cast<float*>(x) + (size_t)s0 * in_elems;
            if (id == BG_VAE_OUT) return out_chunk;
            return reinterpret_cast<float*>(ws + (size_t)(id - 1) * slot_bytes);
        };
        VShape shp[BG_VAE_MAX_SLOTS];
        shp[0] = VShape{in_h, in_w, in_c};
        for (int i = 0; i < n_ops; ++i) {
            const bg_vae_op& o = ops[i];
            const VShape in = shp[o.src];
            const float* src = slot_ptr(o.src);
            float* dst = slot_ptr(o.dst);
            const float* res = o.res >= 0 ? slot_ptr(o.res) : nullptr;
            const int P = in.H * in.W;
            unsigned char* sc = scratch;
            float* stats = nullptr;
            int rc = 0;
            // GroupNorm(1, C) over a sample one wave can hold (the 1-D VAE's blocks): statistics + normalisation + activation in ONE pass
            // wherever the normalised tensor is produced by a 1x1 gather (everything but a materialised im2col over a window)
            const bool window_gather = o.op == BG_VOP_CONV && !conv_implicit_ok(o, in, S) && (o.kh * o.kw > 1 || o.stride != 1 || o.up != 0 || o.pad_mode != 0);
            const bool gn1 = o.gn_gamma && o.gn_groups == 1 && !window_gather && g_tune[TUNE_VAE_GN1_FUSED] != 1 &&
                             gn1_norm_act_supported(P, in.C);
            if (o.gn_gamma) {
                stats = reinterpret_cast<float*>(sc);
                sc += al256((size_t)S * o.gn_groups * 2 * 4);
                if (!gn1 && (rc = bg_groupnorm_stats(src, stats, S, P, in.C, o.gn_groups, o.gn_eps, stream))) return rc;
            }
            // act(GroupNorm(src)) (+ add) -> dst_ in `dt`: the one-pass kernel, or the 1x1 gather over the statistics above
            auto norm_act = [&](void* dst_, int dt, int act, const float* add) -> int {
                if (gn1) return gn1_norm_act(src, dst_, dt, S, P, in.C, o.gn_gamma, o.gn_beta, o.gn_eps, act, add, (hipStream_t)stream);
                return bg_im2col(src, dst_, dt, S, in.H, in.W, in.C, 1, 1, 0, 1, 0, 0, in.H, in.W, stats, o.gn_gamma, o.gn_beta,
                                 o.gn_gamma ? o.gn_groups : 1, act, add, stream);
            };
            VShape outs;
            switch (o.op) {
                case BG_VOP_CONV: {
                    conv_out_shape(o, in, outs);
                    const long long rows = (long long)S * outs.H * outs.W;
                    BG_REQUIRE(rows < (1ll << 31), BG_E_SHAPE, "bg_vae_run: chunk of %d samples has %lld output rows", S, rows);
                    if (conv_implicit_ok(o, in, S)) {
                        // normalise + activate + cast once (a 1x1 gather), then the GEMM's loader walks the window
                        if ((rc = norm_act(sc, o.w_dtype, o.act, nullptr))) return rc;
                        bg_conv_desc d{};
                        d.x = sc; d.S = S; d.H = in.H; d.W = in.W; d.C = in.C;
                        d.kh = o.kh; d.kw = o.kw; d.up = o.up;
                        d.w = o.w; d.bias = o.bias; d.N = o.n_out;
                        d.out = dst; d.ldc = o.n_out;
                        d.add = res; d.ld_add = o.n_out;
                        d.dtype = o.w_dtype; d.zero_page = zero_page;
                        if ((rc = bg_conv_gemm_fwd(&d, stream))) return rc;
                    } else {
                        const int py = o.pad_mode == 0 ? o.kh / 2 : 0, px = o.pad_mode == 0 ? o.kw / 2 : 0;
                        if (gn1) {                                              // (a 1x1 convolution behind GroupNorm(1, C))
                            if ((rc = norm_act(sc, o.w_dtype, o.act, nullptr))) return rc;
                        } else if ((rc = bg_im2col(src, sc, o.w_dtype, S, in.H, in.W, in.C, o.kh, o.kw, o.up, o.stride, py, px, outs.H, outs.W,
                                                   stats, o.gn_gamma, o.gn_beta, o.gn_gamma ? o.gn_groups : 1, o.act, nullptr, stream))) return rc;
                        const int K = o.kh * o.kw * in.C;
                        if ((rc = bg_gemm_bias_act_fwd(sc, K, o.w, o.bias, dst, o.n_out, (int)rows, o.n_out, o.n_pad, K, o.w_dtype, BG_F32,
                                                       BG_ACT_NONE, res, res ? o.n_out : 0, 1, stream))) return rc;
                    }
                    break;
                }
                case BG_VOP_NORM_ACT_ADD:
                    outs = in;
                    if ((rc = norm_act(dst, BG_F32, o.act, res))) return rc;
                    break;
                case BG_VOP_ATTN: {
                    outs = in;
                    const size_t px = (size_t)S * P;
                    void* a = sc;                                           // normalised tokens, operand dtype of the q|k|v GEMM
                    sc += al256(px * in.C * (o.w_dtype == BG_F32 ? 4 : 2));
                    float* qkv = reinterpret_cast<float*>(sc);
                    sc += al256(px * 3 * in.C * 4);
                    void* att = sc;
                    if ((rc = norm_act(a, o.w_dtype, BG_VACT_NONE, nullptr))) return rc;
                    if ((rc = bg_gemm_bias_act_fwd(a, in.C, o.w, o.bias, qkv, 3 * in.C, (int)px, 3 * in.C, o.n_pad, in.C, o.w_dtype, BG_F32,
                                                   BG_ACT_NONE, nullptr, 0, 1, stream))) return rc;
                    if ((rc = bg_small_attn(qkv, 3 * in.C, att, o.w2_dtype, S, P, in.C, o.heads, o.scale, stream))) return rc;
                    if ((rc = bg_gemm_bias_act_fwd(att, in.C, o.w2, o.bias2, dst, in.C, (int)px, in.C, o.n_pad2, in.C, o.w2_dtype, BG_F32,
                                                   BG_ACT_NONE, src, in.C, 1, stream))) return rc;
                    break;
                }
                case BG_VOP_UP1D:
                    outs = VShape{1, in.W * 2, in.C};
                    if ((rc = bg_upsample1d_cubic(src, dst, S, in.W, in.C, stream))) return rc;
                    break;
                case BG_VOP_DOWN1D:
                    outs = VShape{1, in.W / 2, in.C};
                    if ((rc = bg_downsample1d_cubic(src, dst, S, in.W, in.C, stream))) return rc;
                    break;
                default:
                    set_error("bg_vae_run: unknown step %d", o.op);
                    return BG_E_ARG;
            }
            if (o.dst != BG_VAE_OUT) shp[o.dst] = outs;
        }
    }
    return 0;
}
