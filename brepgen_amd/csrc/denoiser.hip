// Whole-net forward of the four BrepGen denoisers (network.py:1107-1126, 1176-1200, 1257-1286, 1357-1393),
// enqueued on one stream from ONE C-ABI call: batch-first [M = B*N, 768] layout end to end (the reference's
// seq-first permutes -- 38 % of its CPU time -- do not exist here), fp32 residual stream, activations in the
// compute dtype, step-invariant conditioning embeds cached across denoising steps.
#include "bg_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <mutex>

namespace bg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
}

int g_tune[TUNE_COUNT] = {0};

// ---- per-kernel event timing ------------------------------------------------------------------------------
bool g_prof_on = false;
namespace {
struct ProfRec { hipEvent_t e0, e1; int kernel; double flops, bytes; };
ProfRec* g_recs = nullptr;
int g_cap = 0, g_n = 0;
bool g_open = false;
const char* const kProfNames[PK_COUNT] = {"gemm16_persistent_kernel(128x128)", "gemm16_kernel(generic: 128x64 / 64x64 tiles)", "gemm_f32", "attn16_kernel", "attn_f32_kernel",
                                          "ln768_kernel", "ddpm_step_kernel", "pndm_step_kernel", "misc", "embed_ln_silu_kernel", "gemm16_p256_kernel(256x256)",
                                          "gemm16_split_pipe_kernel(128x128)", "gemm16_p256_kernel(256x256, split-residual launches)",
                                          "qkv_attn_kernel(256x192 + attention)", "ln_silu_out_kernel", "ffn_fused_kernel(64-row panels)"};
}  // namespace

void prof_pre(hipStream_t s) {
    g_open = false;
    if (g_n >= g_cap) return;
    if (hipEventRecord(g_recs[g_n].e0, s) == hipSuccess) g_open = true;
}

void prof_post(int kernel, double flops, double bytes, hipStream_t s) {
    if (!g_open) return;
    g_open = false;
    ProfRec& r = g_recs[g_n];
    if (hipEventRecord(r.e1, s) != hipSuccess) return;
    r.kernel = kernel; r.flops = flops; r.bytes = bytes;
    ++g_n;
}

__global__ void expand_mask_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n, int E) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i / E];
}

static inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
    size_t off_x, off_h, off_r, off_small, off_f, off_mask, off_stats, off_rows, off_slots, total;
    int M, F;
    int Mrow;           // row capacity of the token buffers: M, or 64 rows per sample where a ragged batch may run slot-packed
};

// Slot-packed variable-length execution (compact.hip: compact_rows_paired + the fused QKV / attention launch): nets whose samples
// have at most 64 tokens, 16-bit modes.  Its rows are 64-row slots of one or two samples -- up to 64 B of them, more than the
// padded batch when nothing can be paired -- so the token buffers of these nets are planned for 64 rows per sample.
static inline bool slot_packing_applies(int net, int B, int S, int E, int dtype) {
    return net == BG_SURFZ && E == 1 && S <= 64 && dtype != BG_F32 && B <= 8192;
}

static Workspace plan(int net, int B, int S, int E, int dtype) {
    Workspace w{};
    const size_t es = (dtype == BG_F32) ? 4 : 2;
    w.M = B * S * E;
    w.F = B * S;
    const bool slots = slot_packing_applies(net, B, S, E, dtype);
    // worst case of the slot-packed layout: S <= 32 -- any two samples fit one slot, at most ceil(B / 2) slots; otherwise nothing
    // may pair: one slot per sample
    const int slot_rows = 64 * (S <= 32 ? (B + 1) / 2 : B);
    w.Mrow = slots && slot_rows > w.M ? slot_rows : w.M;
    const size_t Fr = slots ? (size_t)w.Mrow : (size_t)w.F;        // (the per-face conditioning embed of SurfZNet runs on the compact rows)
    size_t o = 0;
    w.off_x = o; o += align_up((size_t)w.Mrow * 768 * 4);
    w.off_h = o; o += align_up((size_t)w.Mrow * 768 * es);
    w.off_r = o; o += align_up((size_t)w.Mrow * 2304 * es);
    w.off_small = o; o += align_up((size_t)(4 * B + B) * 768 * 4);      // sincos, t0, t1(as fp32 worst case), temb, cvec
    w.off_f = o; o += (net != BG_SURFPOS) ? align_up(Fr * 768 * 4) : 0;
    w.off_mask = o; o += (net == BG_EDGEPOS) ? align_up((size_t)w.M) : 0;
    w.off_stats = o; o += (dtype != BG_F32) ? align_up((size_t)w.Mrow * 12 * 2 * 4) : 0;   // LayerNorm-fold row partials
    w.off_rows = o; o += (net != BG_SURFPOS) ? align_up((size_t)(B + 2 + P256_RULE_ENTRIES) * 4) + align_up((size_t)w.Mrow * 4) : 0;   // var-len: offsets + GEMM partition table, row map
    w.off_slots = o; o += slots ? align_up((size_t)4 * B * 4) : 0;       // slot-packed: (n_a, n_b) per slot, first sample per slot, counts
    w.total = o;
    return w;
}

struct Ctx {
    const bg_denoiser_weights* w;
    hipStream_t s;
    int dtype;
    unsigned char* ws;
    Workspace p;
    float* X;
    void* H;
    void* R;
    // 16-bit compute dtypes with folded LayerNorms: the residual stream lives as two 16-bit planes x = XH + XL in the
    // X region (same bytes as fp32), XH doubles as the A operand of the QKV / FFN1 GEMMs, and the producers of x
    // (token embeds, out-proj, FFN2) leave per-row (sum, sum of squares) partials in `stats`.
    bool fold = false;
    void* XH = nullptr;
    void* XL = nullptr;
    float* stats = nullptr;
    // variable-length execution: valid tokens compacted into rows 0 .. *m_dev-1 (csrc/compact.hip)
    const int* m_dev = nullptr;       // device-side row count (offsets[B])
    const int* src_row = nullptr;     // compact row -> padded-layout token index
    const int* offsets = nullptr;     // per-sample first row, [B+1]
    const int* rule = nullptr;        // 256 / 128 kernel partition of the GEMM launches for this row count (compact.hip)
    const int* slot_desc = nullptr;   // slot-packed batch: (n_a, n_b) per 64-row slot (compact_rows_paired)
    int N_tok = 1;                    // tokens per sample of the padded layout
    double rows_hint = 0.0, pairs_hint = 0.0;   // host-side estimates: GEMM kernel choice + profiler accounting (brepgen_hip.h)
    double rows_plan = 0.0;                     // the row count the caller knows exactly (0 = not): launch plan only
    int concurrent = 0;               // sibling sample groups are in flight on forked streams (n_split > 1)
};

// Linear(k,768)+b -> LN -> SiLU -> Linear(768,n)+b (+adds) ; x fp32 rows (lda), or activations in compute dtype for fc_out
// `to_stream`: the result goes to the token stream X -- fp32 rows (out = c.X, and add == c.X accumulates), or, in
// fold mode, the split pair (XH, XL) with row statistics (accumulating = the stream is the addend)
// `tok`: the rows are tokens.  In a variable-length run they are the COMPACT rows (count on the device): the inputs are
// gathered through c.src_row, the broadcast addends are looked up through it (map_add: `add` is indexed by the padded
// token index / add_div; map_add2 likewise), and with `scatter` the result rows go back to the padded layout.
static int embed_mlp(Ctx& c, const bg_mlp_weights& m, const void* x, int lda, int rows, float* out, int ldc,
                     const float* add, int ld_add, int add_div, const float* add2, int ld_add2, int add2_div,
                     bool to_stream = false, bool tok = false, bool gather = false, bool map_add = false,
                     bool map_add2 = false, bool scatter = false) {
    int rc;
    const bool vl = tok && c.m_dev != nullptr;
    const int* m_dev = vl ? c.m_dev : nullptr;
    const double hint = vl ? c.rows_hint : 0.0;
    if (m.w0_mfma && m.w0_dtype == BG_F32 && embed_ln_silu_supported(m.k_in)) {
        // input embeds (k = 6 / 12 / 48): Linear + LayerNorm + SiLU in one kernel, nothing but the result written
        rc = embed_ln_silu(reinterpret_cast<const float*>(x), lda, rows, m.k_in, m.w0_mfma, m.b0, m.ln_g, m.ln_b, c.H,
                           c.dtype, 1e-5f, c.s, m_dev, (vl && gather) ? c.src_row : nullptr, hint);
        if (rc) return rc;
    } else {
        BG_REQUIRE(!(vl && gather), BG_E_ARG, "bg_denoiser_fwd: variable-length execution needs the fused input embeds (w0_mfma)");
        float* t0 = reinterpret_cast<float*>(c.R);
        GemmArgs g1{x, lda, m.w0, m.b0, t0, 768, rows, 768, 768, m.k_in, BG_F32, BG_ACT_NONE, nullptr, 0, 1};
        g1.gemv_ok = (&m == &c.w->time_embed);                    // one row per distinct timestep
        g1.m_dev = m_dev; g1.rows_hint = hint;
        rc = gemm(g1, m.w0_dtype, c.s);
        if (rc) return rc;
        rc = layernorm768(t0, m.ln_g, m.ln_b, c.H, c.dtype, rows, 1e-5f, /*silu=*/1, c.s, m_dev, hint);
        if (rc) return rc;
    }
    GemmArgs g2{c.H, 768, m.w3, m.b3, out, ldc, rows, m.n_out, m.n_out_pad, 768, BG_F32, BG_ACT_NONE, add, ld_add,
                add ? add_div : 1};
    g2.add2 = add2; g2.ld_add2 = ld_add2; g2.add2_div = add2 ? add2_div : 1;
    g2.m_dev = m_dev; g2.rows_hint = hint;
    if (vl && (map_add || map_add2 || scatter)) {
        g2.row_map = c.src_row; g2.map_add = map_add; g2.map_add2 = map_add2; g2.map_out = scatter;
    }
    if (to_stream && c.fold) {
        g2.out = c.XH; g2.out_dtype = c.dtype; g2.out_lo = c.XL; g2.stats_out = c.stats;
        if (add == c.X) {                                         // accumulate into the stream
            g2.add = nullptr; g2.ld_add = 0; g2.add_div = 1;
            g2.res_hi = c.XH; g2.res_lo = c.XL; g2.ld_res = 768;
        }
    }
    return gemm(g2, c.dtype, c.s);
}

// concurrent: sibling sample groups of the same call are in flight on forked streams (tells the GEMM launcher that a partial
// round of tiles will be filled by the other group: bg_common.h p256_rows)
static int run(const bg_denoiser_weights* w, const bg_denoiser_inputs* in, float* eps_out, void* workspace,
               size_t ws_bytes, hipStream_t s, bool concurrent = false) {
    const int net = w->net, B = in->B, S = in->S, E = (net >= BG_EDGEPOS) ? in->E : 1;
    BG_REQUIRE(net >= BG_SURFPOS && net <= BG_EDGEZ, BG_E_ARG, "bg_denoiser_fwd: bad net id %d", net);
    BG_REQUIRE(w->dtype == BG_BF16 || w->dtype == BG_F16 || w->dtype == BG_F32, BG_E_DTYPE, "bg_denoiser_fwd: compute dtype %d", w->dtype);
    BG_REQUIRE(B > 0 && S > 0 && E > 0, BG_E_SHAPE, "bg_denoiser_fwd: empty shape B=%d S=%d E=%d", B, S, E);
    BG_REQUIRE(in->n_timesteps == 1 || in->n_timesteps == B, BG_E_SHAPE, "bg_denoiser_fwd: n_timesteps must be 1 or B");
    BG_REQUIRE(in->x && in->timesteps && eps_out && workspace, BG_E_ARG, "bg_denoiser_fwd: null pointer");
    BG_REQUIRE(w->n_layer >= 0 && w->n_layer <= BG_MAX_LAYERS, BG_E_ARG, "bg_denoiser_fwd: n_layer");
    BG_REQUIRE((w->class_embed == nullptr) || in->class_label, BG_E_ARG, "bg_denoiser_fwd: class_label required (use_cf)");
    if (net != BG_SURFPOS) BG_REQUIRE(in->surf_pos, BG_E_ARG, "bg_denoiser_fwd: surf_pos missing");
    if (net >= BG_EDGEPOS) BG_REQUIRE(in->surf_z, BG_E_ARG, "bg_denoiser_fwd: surf_z missing");
    if (net == BG_EDGEZ) BG_REQUIRE(in->edge_pos, BG_E_ARG, "bg_denoiser_fwd: edge_pos missing");
    BG_REQUIRE(((uintptr_t)workspace & 255) == 0, BG_E_ALIGN, "bg_denoiser_fwd: workspace must be 256-byte aligned");

    Ctx c;
    c.w = w; c.s = s; c.dtype = w->dtype;
    c.p = plan(net, B, S, E, w->dtype);
    BG_REQUIRE(ws_bytes >= c.p.total, BG_E_WORKSPACE, "bg_denoiser_fwd: workspace %zu < %zu bytes", ws_bytes, c.p.total);
    c.ws = reinterpret_cast<unsigned char*>(workspace);
    c.X = reinterpret_cast<float*>(c.ws + c.p.off_x);
    c.H = c.ws + c.p.off_h;
    c.R = c.ws + c.p.off_r;
    const int Mpad = c.p.M, F = c.p.F, N = S * E, nt = in->n_timesteps;
    // ---- variable-length execution: compact the valid tokens (row count stays on the device) ----------------------
    const bool varlen = in->varlen != 0 && in->mask != nullptr && net != BG_SURFPOS;
    // ragged batches of short sequences: slot-packed rows + the fused QKV / attention launch (bg_tune key 13 = 1: dense packing +
    // GEMM + attention, the bit-equality baseline)
    const bool paired = varlen && slot_packing_applies(net, B, S, E, w->dtype) && w->n_layer > 0 && w->layers[0].qkv_colsum != nullptr &&
                        g_tune[TUNE_QKV_ATTN] != 1;
    const int M = paired ? c.p.Mrow : Mpad;                       // row bound of every token-wise launch
    c.fold = w->dtype != BG_F32 && w->n_layer > 0 && w->layers[0].qkv_colsum != nullptr;
    if (c.fold) {
        for (int li = 0; li < w->n_layer; ++li)
            BG_REQUIRE(w->layers[li].qkv_colsum && w->layers[li].w1_colsum, BG_E_ARG,
                       "bg_denoiser_fwd: layer %d lacks the LayerNorm-fold column sums", li);
        c.XH = c.X;
        c.XL = reinterpret_cast<unsigned char*>(c.X) + (size_t)M * 768 * 2;
        c.stats = reinterpret_cast<float*>(c.ws + c.p.off_stats);
    }
    c.N_tok = N;
    c.concurrent = concurrent ? 1 : 0;
    if (varlen) {
        int* offs = reinterpret_cast<int*>(c.ws + c.p.off_rows);
        int* srow = reinterpret_cast<int*>(c.ws + c.p.off_rows + align_up((size_t)(B + 2 + P256_RULE_ENTRIES) * 4));
        const int n_mask = (net == BG_EDGEPOS) ? S : N, rep = (net == BG_EDGEPOS) ? E : 1;
        int rcc;
        if (paired) {
            int* sd = reinterpret_cast<int*>(c.ws + c.p.off_slots);
            rcc = compact_rows_paired(in->mask, B, n_mask, offs, srow, sd, sd + 2 * B, sd + 3 * B, s, offs + B + 2);
            c.slot_desc = sd;
        } else {
            rcc = compact_rows(in->mask, B, n_mask, rep, offs, srow, s, offs + B + 2);
        }
        if (rcc) return rcc;
        c.offsets = offs; c.m_dev = offs + B; c.src_row = srow; c.rule = offs + B + 2;
        c.rows_hint = in->rows_hint > 0 ? in->rows_hint : 0.0;
        c.pairs_hint = in->pairs_hint > 0 ? in->pairs_hint : 0.0;
        c.rows_plan = in->rows_plan[0] > 0 ? in->rows_plan[0] : 0.0;
        // padded positions of the result are defined as 0 (the valid rows are scattered over this)
        const hipError_t he = hipMemsetAsync(eps_out, 0, (size_t)Mpad * w->fc_out.n_out * sizeof(float), s);
        BG_REQUIRE(he == hipSuccess, (int)he, "bg_denoiser_fwd: hipMemsetAsync failed: %s", hipGetErrorString(he));
    }
    float* small = reinterpret_cast<float*>(c.ws + c.p.off_small);
    float* sc = small;                         // [nt,768] sincos
    float* temb = small + (size_t)3 * B * 768; // [nt,768]
    float* cvec = small + (size_t)4 * B * 768; // [B,768] = time (+ class) embedding per sample
    int rc;

    // ---- time (+class) embedding -> one vector per sample --------------------------------------------------
    if (w->time_table != nullptr && w->time_table_rows > 0) {
        // the time-embedding MLP is a function of the weights and the timestep only: looked up in the table the caller precomputed
        if ((rc = cond_vector_table(w->time_table, w->time_table_rows, in->timesteps, nt, w->class_embed, in->class_label, cvec, B, s))) return rc;
    } else {
        if ((rc = sincos_embed(in->timesteps, nt, sc, s))) return rc;
        if ((rc = embed_mlp(c, w->time_embed, sc, 768, nt, temb, 768, nullptr, 0, 1, nullptr, 0, 1))) return rc;
        if ((rc = cond_vector(temb, nt, w->class_embed, in->class_label, cvec, B, s))) return rc;
    }

    // ---- token embeddings -> X [M,768] fp32 ----------------------------------------------------------------
    // step-invariant part (per face): SurfZ: p_embed(surfPos); Edge nets: surfp_embed(surfPos)+surfz_embed(surfZ)
    float* fcond = nullptr;
    // SurfZNet, variable-length, no conditioning cache: p_embed(surfPos) is computed on the compact rows (the cache, when
    // the caller provides one, stays in the padded per-face layout so that it does not depend on the mask)
    const bool fcond_compact = varlen && net == BG_SURFZ && in->cond_cache == nullptr;
    if (net != BG_SURFPOS) {
        fcond = in->cond_cache ? in->cond_cache : reinterpret_cast<float*>(c.ws + c.p.off_f);
        if (!(in->cond_cache && in->cond_cache_valid)) {
            if (net == BG_SURFZ && fcond_compact) {
                // faces == tokens: without a cache to fill, only the valid faces need their conditioning embed
                if ((rc = embed_mlp(c, w->embed[1], in->surf_pos, 6, paired ? M : F, fcond, 768, nullptr, 0, 1, nullptr, 0, 1, false, true, true))) return rc;
            } else if (net == BG_SURFZ) {
                if ((rc = embed_mlp(c, w->embed[1], in->surf_pos, 6, F, fcond, 768, nullptr, 0, 1, nullptr, 0, 1))) return rc;
            } else {
                if ((rc = embed_mlp(c, w->embed[0], in->surf_pos, 6, F, fcond, 768, nullptr, 0, 1, nullptr, 0, 1))) return rc;
                if ((rc = embed_mlp(c, w->embed[1], in->surf_z, 48, F, fcond, 768, fcond, 768, 1, nullptr, 0, 1))) return rc;
            }
        }
    }
    switch (net) {
        case BG_SURFPOS:   // tokens = p_embed(x) + c
            rc = embed_mlp(c, w->embed[0], in->x, 6, M, c.X, 768, cvec, 768, N, nullptr, 0, 1, true);
            break;
        // (variable-length: rows are the compact tokens; x is gathered, cvec [B] and the per-face conditioning
        //  fcond [B*S] -- kept in the padded layout, so the conditioning cache is unaffected -- are looked up through the
        //  row map: sample = token / N, face = token / E)
        case BG_SURFZ:     // tokens = z_embed(x) + p_embed(surfPos) + c
            rc = embed_mlp(c, w->embed[0], in->x, 48, M, c.X, 768, cvec, 768, N, fcond, 768, 1, true, true, true, true, !fcond_compact);
            break;
        case BG_EDGEPOS:   // tokens = edgep_embed(x) + surf[m/E] + c
            rc = embed_mlp(c, w->embed[2], in->x, 6, M, c.X, 768, cvec, 768, N, fcond, 768, E, true, true, true, true, true);
            break;
        default:           // EdgeZ: edgez_embed(x[:, :12]) + vertp_fc(x[:, 12:]) + edgep_embed(edgePos) + surf[m/E] + c
            rc = embed_mlp(c, w->embed[3], in->x, 18, M, c.X, 768, cvec, 768, N, fcond, 768, E, true, true, true, true, true);
            if (!rc) rc = embed_mlp(c, w->embed[4], in->x + 12, 18, M, c.X, 768, c.X, 768, 1, nullptr, 0, 1, true, true, true);
            if (!rc) rc = embed_mlp(c, w->embed[2], in->edge_pos, 6, M, c.X, 768, c.X, 768, 1, nullptr, 0, 1, true, true, true);
            break;
    }
    if (rc) return rc;

    // ---- key-padding mask [B,N] ----------------------------------------------------------------------------
    const uint8_t* key_pad = in->mask;
    if (varlen) key_pad = nullptr;                                // every compact row is a valid key
    if (net == BG_EDGEPOS && in->mask && !varlen) {
        uint8_t* mexp = c.ws + c.p.off_mask;
        const size_t n = (size_t)M;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(expand_mask_kernel, dim3(grid), dim3(256), 0, s, in->mask, mexp, n, E);
        if ((rc = launch_status("expand_mask"))) return rc;
        key_pad = mexp;
    }

    // ---- 12 pre-LN encoder layers ---------------------------------------------------------------------------
    BG_REQUIRE(!paired || c.fold, BG_E_ARG, "bg_denoiser_fwd: slot-packed execution needs the LayerNorm-fold operands");
    const bool fused_qkv = c.fold && !varlen && g_tune[TUNE_QKV_ATTN] != 1 &&
                           qkv_attn_eligible(B, N, c.dtype, c.stats, w->layers[0].qkv_colsum, w->layers[0].b_qkv) &&
                           (g_tune[TUNE_QKV_ATTN] == 2 || qkv_attn_worthwhile(B, N));
    for (int li = 0; c.fold && li < w->n_layer; ++li) {
        // x = XH + XL.  LN1 / LN2 are folded: QKV and FFN1 read the raw 16-bit rows XH and normalise in their epilogue.
        const bg_layer_weights& L = w->layers[li];
        GemmArgs qkv{c.XH, 768, L.w_qkv, L.b_qkv, c.R, 2304, M, 2304, 2304, 768, c.dtype, BG_ACT_NONE, nullptr, 0, 1};
        qkv.stats_in = c.stats; qkv.colsum = L.qkv_colsum;
        qkv.m_dev = c.m_dev; qkv.rule_table = c.rule; qkv.rows_hint = c.rows_hint; qkv.rows_plan = c.rows_plan; qkv.concurrent = c.concurrent;
        if (paired) {
            // ragged batch, slot-packed: one or two whole samples per 64-row slot (qkv_attn.hip, PAIR)
            if ((rc = qkv_attention_paired(c.XH, L.w_qkv, L.b_qkv, L.qkv_colsum, c.stats, c.H, nullptr, c.m_dev, c.slot_desc,
                                           B < (M + 63) / 64 ? B : (M + 63) / 64, M, c.dtype,
                                           1e-5f, s, c.rows_hint, c.pairs_hint))) return rc;
        } else if (fused_qkv) {
            // short, equally long sequences (SurfPosNet; SurfZNet executed densely): q|k|v never leave the CU (qkv_attn.hip; bit-identical)
            if ((rc = qkv_attention(c.XH, L.w_qkv, L.b_qkv, L.qkv_colsum, c.stats, c.H, key_pad, B, N, c.dtype, 1e-5f, s))) return rc;
        } else {
            if ((rc = gemm(qkv, c.dtype, s))) return rc;
            if ((rc = attention(c.R, key_pad, c.H, B, N, c.dtype, s, c.offsets, c.pairs_hint, c.rows_hint))) return rc;
        }
        GemmArgs op{c.H, 768, L.w_o, L.b_o, c.XH, 768, M, 768, 768, 768, c.dtype, BG_ACT_NONE, nullptr, 0, 1};
        op.out_lo = c.XL; op.res_hi = c.XH; op.res_lo = c.XL; op.ld_res = 768; op.stats_out = c.stats;
        op.m_dev = c.m_dev; op.rule_table = c.rule; op.rows_hint = c.rows_hint; op.rows_plan = c.rows_plan; op.concurrent = c.concurrent;
        if ((rc = gemm(op, c.dtype, s))) return rc;
        if (L.w_1f && L.w_2f && g_tune[TUNE_FFN_FUSED] != 1) {
            // FFN1 + ReLU + FFN2 + residual as one launch: the [M, 1024] hidden tensor stays in the CU's LDS (ffn_fused.hip; bit-identical)
            FfnArgs ff{c.XH, c.XL, c.stats, L.w_1f, L.b_1, L.w1_colsum, L.w_2f, L.b_2, M, M, c.m_dev, 1e-5f};      // (statistics stride = the launch's row bound, as the GEMMs')
            BG_REQUIRE(ffn_fused_eligible(ff, c.dtype), BG_E_ARG, "bg_denoiser_fwd: w_1f / w_2f given, but the fused FFN launch does not apply");
            if ((rc = ffn_fused(ff, c.dtype, s, c.rows_hint))) return rc;
            continue;
        }
        GemmArgs f1{c.XH, 768, L.w_1, L.b_1, c.R, 1024, M, 1024, 1024, 768, c.dtype, BG_ACT_RELU, nullptr, 0, 1};
        f1.stats_in = c.stats; f1.colsum = L.w1_colsum;
        f1.m_dev = c.m_dev; f1.rule_table = c.rule; f1.rows_hint = c.rows_hint; f1.rows_plan = c.rows_plan; f1.concurrent = c.concurrent;
        if ((rc = gemm(f1, c.dtype, s))) return rc;
        GemmArgs f2{c.R, 1024, L.w_2, L.b_2, c.XH, 768, M, 768, 768, 1024, c.dtype, BG_ACT_NONE, nullptr, 0, 1};
        f2.out_lo = c.XL; f2.res_hi = c.XH; f2.res_lo = c.XL; f2.ld_res = 768; f2.stats_out = c.stats;
        f2.m_dev = c.m_dev; f2.rule_table = c.rule; f2.rows_hint = c.rows_hint; f2.rows_plan = c.rows_plan; f2.concurrent = c.concurrent;
        if ((rc = gemm(f2, c.dtype, s))) return rc;
    }
    for (int li = 0; !c.fold && li < w->n_layer; ++li) {
        const bg_layer_weights& L = w->layers[li];
        auto vl = [&](GemmArgs& g) { g.m_dev = c.m_dev; g.rule_table = c.rule; g.rows_hint = c.rows_hint; g.rows_plan = c.rows_plan; g.concurrent = c.concurrent; };
        if ((rc = layernorm768(c.X, L.ln1_g, L.ln1_b, c.H, c.dtype, M, 1e-5f, 0, s, c.m_dev, c.rows_hint))) return rc;
        GemmArgs qkv{c.H, 768, L.w_qkv, L.b_qkv, c.R, 2304, M, 2304, 2304, 768, c.dtype, BG_ACT_NONE, nullptr, 0, 1};
        vl(qkv);
        if ((rc = gemm(qkv, c.dtype, s))) return rc;
        if ((rc = attention(c.R, key_pad, c.H, B, N, c.dtype, s, c.offsets, c.pairs_hint, c.rows_hint))) return rc;
        GemmArgs op{c.H, 768, L.w_o, L.b_o, c.X, 768, M, 768, 768, 768, BG_F32, BG_ACT_NONE, c.X, 768, 1};
        vl(op);
        if ((rc = gemm(op, c.dtype, s))) return rc;
        if ((rc = layernorm768(c.X, L.ln2_g, L.ln2_b, c.H, c.dtype, M, 1e-5f, 0, s, c.m_dev, c.rows_hint))) return rc;
        GemmArgs f1{c.H, 768, L.w_1, L.b_1, c.R, 1024, M, 1024, 1024, 768, c.dtype, BG_ACT_RELU, nullptr, 0, 1};
        vl(f1);
        if ((rc = gemm(f1, c.dtype, s))) return rc;
        GemmArgs f2{c.R, 1024, L.w_2, L.b_2, c.X, 768, M, 768, 768, 1024, BG_F32, BG_ACT_NONE, c.X, 768, 1};
        vl(f2);
        if ((rc = gemm(f2, c.dtype, s))) return rc;
    }

    // ---- final LayerNorm + fc_out ---------------------------------------------------------------------------
    const bg_mlp_weights& mo = w->fc_out;
    if (mo.w0_colsum != nullptr) {
        BG_REQUIRE(c.fold && mo.w0_dtype == c.dtype && ln_silu_out_supported(mo.n_out, mo.n_out_pad), BG_E_ARG,
                   "bg_denoiser_fwd: fc_out carries the folded final LayerNorm (w0_colsum): needs the 16-bit LayerNorm-fold layers and n_out <= 48");
        // 16-bit modes: the final LayerNorm is folded into fc_out.0 (the epilogue of QKV / FFN1: raw XH rows in, the statistics the
        // last FFN2 left behind), and LayerNorm + SiLU + Linear(768, n_out) are one launch (out_tail.hip) -- two launches, and the
        // [M, 768] intermediate crosses HBM once, in 16 bits
        GemmArgs g0{c.XH, 768, mo.w0, mo.b0, c.H, 768, M, 768, 768, 768, c.dtype, BG_ACT_NONE, nullptr, 0, 1};
        g0.stats_in = c.stats; g0.colsum = mo.w0_colsum;
        g0.m_dev = c.m_dev; g0.rule_table = c.rule; g0.rows_hint = c.rows_hint; g0.rows_plan = c.rows_plan; g0.concurrent = c.concurrent;
        if ((rc = gemm(g0, c.dtype, s))) return rc;
        // (variable-length: the compact result rows are scattered into the zero-filled padded eps_out)
        return ln_silu_out(c.H, mo.ln_g, mo.ln_b, mo.w3, mo.b3, eps_out, mo.n_out, mo.n_out_pad, M, c.dtype, 1e-5f, s, c.m_dev,
                           varlen ? c.src_row : nullptr, c.rows_hint);
    }
    // fc_out.0 reads the final-LN output from H and writes its fp32 result to R; the LN+SiLU then overwrites H.
    {
        void* hf = c.H;
        if (c.fold) rc = layernorm768_split(c.XH, c.XL, w->lnf_g, w->lnf_b, hf, c.dtype, M, 1e-5f, s, c.m_dev, c.rows_hint);
        else rc = layernorm768(c.X, w->lnf_g, w->lnf_b, hf, c.dtype, M, 1e-5f, 0, s, c.m_dev, c.rows_hint);
        if (rc) return rc;
        // (variable-length: the compact result rows are scattered into the zero-filled padded eps_out)
        rc = embed_mlp(c, mo, hf, 768, M, eps_out, mo.n_out, nullptr, 0, 1, nullptr, 0, 1, false, true, false, false, false, true);
    }
    return rc;
}

}  // namespace bg

extern "C" int bg_profile_begin(int max_launches) {
    using namespace bg;
    BG_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), BG_E_ARG, "bg_profile_begin: bad max_launches");
    BG_REQUIRE(g_recs == nullptr, BG_E_ARG, "bg_profile_begin: already profiling");
    g_recs = new ProfRec[max_launches];
    for (int i = 0; i < max_launches; ++i) {
        if (hipEventCreate(&g_recs[i].e0) != hipSuccess || hipEventCreate(&g_recs[i].e1) != hipSuccess) {
            set_error("bg_profile_begin: hipEventCreate failed");
            return BG_E_ARG;
        }
    }
    g_cap = max_launches; g_n = 0; g_prof_on = true;
    return 0;
}

extern "C" int bg_profile_end(bg_profile_row* rows, int max_rows) {
    using namespace bg;
    BG_REQUIRE(g_recs != nullptr, BG_E_ARG, "bg_profile_end: not profiling");
    g_prof_on = false;
    (void)hipDeviceSynchronize();            // measurement aid only -- never on the product path
    bg_profile_row agg[PK_COUNT];
    for (int k = 0; k < PK_COUNT; ++k) agg[k] = bg_profile_row{kProfNames[k], 0, 0.0, 0.0, 0.0};
    for (int i = 0; i < g_n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_recs[i].e0, g_recs[i].e1) != hipSuccess) continue;
        bg_profile_row& a = agg[g_recs[i].kernel];
        a.launches += 1; a.total_ms += ms; a.flops += g_recs[i].flops; a.bytes += g_recs[i].bytes;
    }
    for (int i = 0; i < g_cap; ++i) { (void)hipEventDestroy(g_recs[i].e0); (void)hipEventDestroy(g_recs[i].e1); }
    delete[] g_recs;
    g_recs = nullptr; g_cap = 0; g_n = 0;
    int n = 0;
    for (int k = 0; k < PK_COUNT && n < max_rows; ++k)
        if (agg[k].launches > 0 && rows) rows[n++] = agg[k];
    return n;
}

extern "C" int bg_tune_set(int key, int value) {
    BG_REQUIRE(key >= 0 && key < bg::TUNE_COUNT, BG_E_ARG, "bg_tune_set: unknown key %d", key);
    bg::g_tune[key] = value;
    return 0;
}

extern "C" int bg_gemm_p256_rows(int rows, int n_cols, int split_residual, int concurrent) {
    BG_REQUIRE(rows >= 0 && n_cols > 0 && (n_cols & 255) == 0, BG_E_ARG, "bg_gemm_p256_rows: rows >= 0 and n_cols a positive multiple of 256 expected");
    return bg::p256_rows(rows, n_cols >> 8, split_residual != 0, concurrent != 0);
}

extern "C" int bg_abi_version(void) { return BG_ABI_VERSION; }
extern "C" const char* bg_last_error(void) { return bg::g_err; }

namespace bg {
constexpr int MAX_SPLIT = 4;
// contiguous sample groups of an n-way split (sizes differ by at most one)
static inline void split_range(int B, int n, int k, int& lo, int& hi) {
    const int base = B / n, rem = B % n;
    lo = k * base + (k < rem ? k : rem);
    hi = lo + base + (k < rem ? 1 : 0);
}
static size_t plan_total_split(int net, int B, int S, int E, int dtype, int n) {
    size_t t = 0;
    for (int k = 0; k < n; ++k) {
        int lo, hi;
        split_range(B, n, k, lo, hi);
        if (hi > lo) t += align_up(plan(net, hi - lo, S, E, dtype).total);
    }
    return t;
}
// helper streams + fork / join events of the split mode: created once PER DEVICE (a process that drives nets on several devices
// must not enqueue a sub-batch on another device's stream), used under a mutex (enqueue only: microseconds)
struct SplitState {
    hipStream_t aux[MAX_SPLIT - 1];
    hipEvent_t fork, join[MAX_SPLIT - 1];
    bool tried = false, ok = false;
};
constexpr int MAX_SPLIT_DEVICES = 64;
static SplitState g_split_dev[MAX_SPLIT_DEVICES];
static std::mutex g_split_mutex;
// (called with g_split_mutex held) -> the state of the CURRENT device, or nullptr
static SplitState* split_state() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_SPLIT_DEVICES) return nullptr;
    SplitState& st = g_split_dev[dev];
    if (!st.tried) {
        st.tried = true;
        bool ok = hipEventCreateWithFlags(&st.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < MAX_SPLIT - 1 && ok; ++i)
            ok = hipStreamCreateWithFlags(&st.aux[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&st.join[i], hipEventDisableTiming) == hipSuccess;
        st.ok = ok;
    }
    return st.ok ? &st : nullptr;
}
}  // namespace bg

extern "C" int bg_slot_packing_applies(int net, int B, int S, int E, int dtype, int fold) {
    // the predicate bg_denoiser_fwd itself evaluates for a variable-length call (`paired`), for hosts that want to compute rows_plan
    if (net < BG_EDGEPOS) E = 1;
    return (bg::slot_packing_applies(net, B, S, E, dtype) && fold != 0 && bg::g_tune[bg::TUNE_QKV_ATTN] != 1) ? 1 : 0;
}

extern "C" size_t bg_workspace_bytes(int net, int B, int S, int E, int dtype) {
    if (B <= 0 || S <= 0) return 0;
    if (net < BG_EDGEPOS) E = 1;
    if (E <= 0) return 0;
    size_t t = bg::plan(net, B, S, E, dtype).total;
    for (int n = 2; n <= bg::MAX_SPLIT; ++n) {
        const size_t ts = bg::plan_total_split(net, B, S, E, dtype, n);
        t = ts > t ? ts : t;
    }
    return t;
}

extern "C" int bg_denoiser_fwd(const bg_denoiser_weights* w, const bg_denoiser_inputs* in, float* eps_out,
                               void* workspace, size_t workspace_bytes, bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(w && in, BG_E_ARG, "bg_denoiser_fwd: null descriptor");
    hipStream_t s = (hipStream_t)stream;
    BG_REQUIRE(in->n_split >= 0, BG_E_ARG, "bg_denoiser_fwd: n_split must be >= 0 (got %d)", in->n_split);
    const int ns = in->n_split > MAX_SPLIT ? MAX_SPLIT : in->n_split;
    if (ns < 2 || in->B < ns) return run(w, in, eps_out, workspace, workspace_bytes, s);

    // ---- n-way split over sample groups, one stream each, fork / join by events on the caller's stream ----
    const int net = w->net;
    BG_REQUIRE(net >= BG_SURFPOS && net <= BG_EDGEZ, BG_E_ARG, "bg_denoiser_fwd: bad net id %d", net);
    BG_REQUIRE(in->B > 0 && in->S > 0 && eps_out && workspace, BG_E_ARG, "bg_denoiser_fwd: null pointer / empty shape");
    const int S = in->S, E = (net >= BG_EDGEPOS) ? in->E : 1;
    BG_REQUIRE(E > 0, BG_E_SHAPE, "bg_denoiser_fwd: empty shape");
    BG_REQUIRE(((uintptr_t)workspace & 255) == 0, BG_E_ALIGN, "bg_denoiser_fwd: workspace must be 256-byte aligned");
    BG_REQUIRE(workspace_bytes >= plan_total_split(net, in->B, S, E, w->dtype, ns), BG_E_WORKSPACE,
               "bg_denoiser_fwd: workspace too small for n_split = %d", ns);
    static const int kInCols[4] = {6, 48, 6, 18};                 // channels of x per net (SurfPos, SurfZ, EdgePos, EdgeZ)
    const size_t tok = (size_t)S * E;
    const size_t mask_per_sample = (net == BG_EDGEZ) ? tok : (size_t)S;
    std::lock_guard<std::mutex> lock(g_split_mutex);
    SplitState* sp = split_state();
    BG_REQUIRE(sp != nullptr, BG_E_ARG, "bg_denoiser_fwd: could not create the helper streams of the split mode on this device");
    SplitState& g_split = *sp;
    hipError_t he = hipEventRecord(g_split.fork, s);
    BG_REQUIRE(he == hipSuccess, (int)he, "bg_denoiser_fwd: hipEventRecord failed: %s", hipGetErrorString(he));
    unsigned char* wsp = reinterpret_cast<unsigned char*>(workspace);
    int rc = 0;
    for (int k = 0; k < ns; ++k) {
        int lo, hi;
        split_range(in->B, ns, k, lo, hi);
        bg_denoiser_inputs sub = *in;
        sub.B = hi - lo;
        sub.n_split = 1;
        sub.x = in->x + (size_t)lo * tok * kInCols[net];
        if (in->surf_pos) sub.surf_pos = in->surf_pos + (size_t)lo * S * 6;
        if (in->surf_z) sub.surf_z = in->surf_z + (size_t)lo * S * 48;
        if (in->edge_pos) sub.edge_pos = in->edge_pos + (size_t)lo * tok * 6;
        if (in->mask) sub.mask = in->mask + (size_t)lo * mask_per_sample;
        if (in->n_timesteps == in->B) { sub.timesteps = in->timesteps + lo; sub.n_timesteps = sub.B; }
        if (in->class_label) sub.class_label = in->class_label + lo;
        if (in->cond_cache) sub.cond_cache = in->cond_cache + (size_t)lo * S * 768;
        sub.rows_hint = in->rows_hint * sub.B / in->B;            // estimates: proportional share
        sub.pairs_hint = in->pairs_hint * sub.B / in->B;
        sub.rows_plan[0] = in->rows_plan[k];                      // (exact per group, or 0)
        const size_t bytes = align_up(plan(net, sub.B, S, E, w->dtype).total);
        hipStream_t sk = k == 0 ? s : g_split.aux[k - 1];
        if (k > 0 && (he = hipStreamWaitEvent(sk, g_split.fork, 0)) != hipSuccess) { rc = (int)he; break; }
        rc = run(w, &sub, eps_out + (size_t)lo * tok * w->fc_out.n_out, wsp, bytes, sk, /*concurrent=*/true);
        if (rc) break;
        wsp += bytes;
    }
    // join every helper stream back into the caller's stream -- also after an error, so nothing is left dangling
    for (int k = 1; k < ns; ++k) {
        if (hipEventRecord(g_split.join[k - 1], g_split.aux[k - 1]) == hipSuccess)
            (void)hipStreamWaitEvent(s, g_split.join[k - 1], 0);
    }
    return rc;
}

// ---- stand-alone pieces of the whole-net call (same code paths, caller-owned scratch) ---------------------------------
extern "C" size_t bg_embed_mlp_scratch_bytes(int rows, int dtype) {
    if (rows <= 0) return 0;
    const size_t es = (dtype == BG_F32) ? 4 : 2;
    return bg::align_up((size_t)rows * 768 * es) + bg::align_up((size_t)rows * 768 * 4);
}

extern "C" int bg_embed_mlp_fwd(const bg_mlp_weights* m, int dtype, const void* x, int lda, int rows, float* out, int ldc,
                                const float* add, int ld_add, int add_div, void* scratch, size_t scratch_bytes,
                                bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(m && x && out && scratch, BG_E_ARG, "bg_embed_mlp_fwd: null pointer");
    BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16 || dtype == BG_F16, BG_E_DTYPE, "bg_embed_mlp_fwd: dtype %d", dtype);
    BG_REQUIRE(rows > 0 && lda >= m->k_in && ldc >= m->n_out, BG_E_SHAPE, "bg_embed_mlp_fwd: bad shape");
    BG_REQUIRE(scratch_bytes >= bg_embed_mlp_scratch_bytes(rows, dtype), BG_E_WORKSPACE, "bg_embed_mlp_fwd: scratch too small");
    BG_REQUIRE(((uintptr_t)scratch & 255) == 0, BG_E_ALIGN, "bg_embed_mlp_fwd: scratch must be 256-byte aligned");
    bg_denoiser_weights none{};
    Ctx c;
    c.w = &none; c.s = (hipStream_t)stream; c.dtype = dtype; c.ws = reinterpret_cast<unsigned char*>(scratch);
    c.X = nullptr;
    c.H = c.ws;
    c.R = c.ws + align_up((size_t)rows * 768 * ((dtype == BG_F32) ? 4 : 2));
    return embed_mlp(c, *m, x, lda, rows, out, ldc, add, ld_add, add ? add_div : 1, nullptr, 0, 1);
}

extern "C" size_t bg_encoder_layer_scratch_bytes(int B, int N, int dtype) {
    if (B <= 0 || N <= 0) return 0;
    const size_t es = (dtype == BG_F32) ? 4 : 2, M = (size_t)B * N;
    return bg::align_up(M * 768 * es) + bg::align_up(M * 2304 * es);
}

// One pre-LN encoder layer (nn.TransformerEncoderLayer(norm_first=True), network.py:1076-1078) on an fp32 residual
// stream x [B*N, 768], in place -- the unfolded formulation: LayerNorm kernels + unfolded weights (qkv_colsum == NULL).
extern "C" int bg_encoder_layer_fwd(const bg_layer_weights* L, int dtype, float* x, const uint8_t* key_pad, int B, int N,
                                    void* scratch, size_t scratch_bytes, bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(L && x && scratch, BG_E_ARG, "bg_encoder_layer_fwd: null pointer");
    BG_REQUIRE(dtype == BG_F32 || dtype == BG_BF16 || dtype == BG_F16, BG_E_DTYPE, "bg_encoder_layer_fwd: dtype %d", dtype);
    BG_REQUIRE(B > 0 && N > 0, BG_E_SHAPE, "bg_encoder_layer_fwd: empty shape");
    BG_REQUIRE(L->qkv_colsum == nullptr && L->w1_colsum == nullptr && L->ln1_g && L->ln2_g, BG_E_ARG,
               "bg_encoder_layer_fwd: takes unfolded weights (the LayerNorm-folded layers run inside bg_denoiser_fwd)");
    BG_REQUIRE(scratch_bytes >= bg_encoder_layer_scratch_bytes(B, N, dtype), BG_E_WORKSPACE, "bg_encoder_layer_fwd: scratch too small");
    BG_REQUIRE(((uintptr_t)scratch & 255) == 0, BG_E_ALIGN, "bg_encoder_layer_fwd: scratch must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int M = B * N;
    unsigned char* ws = reinterpret_cast<unsigned char*>(scratch);
    void* H = ws;
    void* R = ws + align_up((size_t)M * 768 * ((dtype == BG_F32) ? 4 : 2));
    int rc;
    if ((rc = layernorm768(x, L->ln1_g, L->ln1_b, H, dtype, M, 1e-5f, 0, s))) return rc;
    GemmArgs qkv{H, 768, L->w_qkv, L->b_qkv, R, 2304, M, 2304, 2304, 768, dtype, BG_ACT_NONE, nullptr, 0, 1};
    if ((rc = gemm(qkv, dtype, s))) return rc;
    if ((rc = attention(R, key_pad, H, B, N, dtype, s))) return rc;
    GemmArgs op{H, 768, L->w_o, L->b_o, x, 768, M, 768, 768, 768, BG_F32, BG_ACT_NONE, x, 768, 1};
    if ((rc = gemm(op, dtype, s))) return rc;
    if ((rc = layernorm768(x, L->ln2_g, L->ln2_b, H, dtype, M, 1e-5f, 0, s))) return rc;
    GemmArgs f1{H, 768, L->w_1, L->b_1, R, 1024, M, 1024, 1024, 768, dtype, BG_ACT_RELU, nullptr, 0, 1};
    if ((rc = gemm(f1, dtype, s))) return rc;
    GemmArgs f2{R, 1024, L->w_2, L->b_2, x, 768, M, 768, 768, 1024, BG_F32, BG_ACT_NONE, x, 768, 1};
    return gemm(f2, dtype, s);
}

