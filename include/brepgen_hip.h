/*
 * brepgen_hip.h -- C ABI of libbrepgen_hip.so (gfx950 / MI355X only).
 *
 * The reference (samxuxiang/BrepGen) has no FFI / plugin layer: its hot path is
 * Python calling torch ops.  This ABI is therefore the boundary a maintainer
 * would bind *under* the reference's Python call surface; every entry point
 * names the reference code it replaces (file:line in /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. the PyTorch
 *     caching allocator), including the workspace; the library never allocates,
 *     never synchronises, keeps no global state; everything is enqueued on the
 *     `stream` argument (pass torch.cuda.current_stream().cuda_stream);
 *   - tensors are dense row-major, batch-first: tokens [B, N, C] == rows [M=B*N, C];
 *   - return 0 on success, a positive hipError_t from the launch, or a negative
 *     BG_E_* argument error; bg_last_error() gives the thread-local message;
 *   - nothing throws across the boundary.
 */
#ifndef BREPGEN_HIP_H
#define BREPGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_ABI_VERSION 6

typedef void* bg_stream_t;              /* hipStream_t */

enum bg_dtype { BG_F32 = 0, BG_F16 = 1, BG_BF16 = 2 };
enum bg_act { BG_ACT_NONE = 0, BG_ACT_RELU = 1 };
enum bg_net { BG_SURFPOS = 0, BG_SURFZ = 1, BG_EDGEPOS = 2, BG_EDGEZ = 3 };
enum bg_err { BG_E_ARG = -1, BG_E_SHAPE = -2, BG_E_WORKSPACE = -3, BG_E_DTYPE = -4, BG_E_ALIGN = -5 };

#define BG_D_MODEL 768
#define BG_N_HEAD 12
#define BG_D_HEAD 64
#define BG_D_FF 1024
#define BG_MAX_LAYERS 12
#define BG_MAX_EMBEDS 5

int bg_abi_version(void);
const char* bg_last_error(void);

/* ---- elementwise / normalisation ------------------------------------------------------ */

/* network.py:1043-1063 sincos_embedding: out[i, 0:384]=cos(t_i*f), out[i,384:768]=sin(t_i*f). */
int bg_sincos_embed(const int64_t* timesteps, int n, float* out /*[n,768]*/, bg_stream_t stream);

/* torch.nn.LayerNorm(768, eps) as used at network.py:1076-1099 (optionally followed by SiLU, the
 * `LayerNorm -> SiLU` pair inside every embed MLP).  x fp32 [M,768]; y fp32 or bf16 [M,768]. */
int bg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype,
                     int M, float eps, int fuse_silu, bg_stream_t stream);

/* nn.Linear (+ReLU) (+residual / broadcast add) -- every addmm of network.py:1076-1099.
 *   out[m,n] = act(sum_k a[m,k]*w[n,k] + bias[n]) + (add ? add[(m/add_div)*ld_add + n] : 0)
 * a: [M,K] dtype ab_dtype (BG_F32 or BG_BF16), row stride lda; w: [N_pad,K] same dtype (nn.Linear
 * layout, row stride K); out: fp32 or bf16, row stride ldc, only columns < N are written.
 * BG_BF16: K % 64 == 0, N_pad % 64 == 0 (weights zero-padded by the packer).  BG_F32: any shape.
 * `add` may alias `out` (in-place residual, add_div = 1). */
int bg_gemm_bias_act_fwd(const void* a, int lda, const void* w, const float* bias, void* out, int ldc,
                         int M, int N, int N_pad, int K, int ab_dtype, int out_dtype, int act,
                         const float* add, int ld_add, int add_div, bg_stream_t stream);

/* The same GEMM with every epilogue option of the 16-bit kernels, as one descriptor.  Beyond bg_gemm_bias_act_fwd:
 *   add2            second fp32 addend, row (m / add2_div)
 *   out_lo          split output: the fp32 result v is stored as hi = T(v) -> out and lo = T(v - hi) -> out_lo
 *                   (two 16-bit planes, row stride ldc); hi is directly the next GEMM's A operand
 *   res_hi, res_lo  split residual addend rows [M,N] (row stride ld_res); may alias out / out_lo (in place)
 *   stats_out       [N_pad/64][M][2] fp32 (part-major): per 64-column group (sum, sum of squares) of v
 *   stats_in, colsum  LayerNorm fold: a holds raw (un-normalised) 16-bit rows, w = T(gamma * W), bias = b + W beta,
 *                   colsum[n] = sum_k w[n,k]; the epilogue computes act(rstd_m * acc - mean_m * rstd_m * colsum[n] +
 *                   bias[n]) with mean / rstd summed from stats_in [K/64][M][2] (written by a producer's stats_out).
 *                   (Launches large enough for the 256 x 256 persistent kernel take it when M is even and stats_in is
 *                   16-byte aligned -- it fetches the partials with its LDS-DMA stream, two rows per element; otherwise the
 *                   128 x 128 kernel runs.  Same bits either way.)
 * The new options need ab_dtype BG_BF16 | BG_F16, N == N_pad and ldc % 8 == 0. */
typedef struct {
    const void* a; int lda;
    const void* w; const float* bias;
    void* out; int ldc;
    int M, N, N_pad, K;
    int ab_dtype, out_dtype, act;
    const float* add; int ld_add, add_div;
    const float* add2; int ld_add2, add2_div;
    void* out_lo;
    const void* res_hi; const void* res_lo; int ld_res;
    float* stats_out;
    const float* stats_in; const float* colsum; float ln_eps;
} bg_gemm_desc;
int bg_gemm_ex_fwd(const bg_gemm_desc* d, bg_stream_t stream);

/* Convolution as an IMPLICIT GEMM (the 3x3 / k5 convolutions of the VAE decoders: ResnetBlock2D.conv1/conv2, Upsample2D.conv,
 * ResConvBlock.conv_1/conv_2 -- network.py:30-83, 188-299, 948-1040 via diffusers): the MFMA GEMM's activation loader
 * gathers the window straight from the channels-last tensor, so the kh*kw-fold im2col matrix is never written.
 *   x     16-bit channels-last activations [S, H, W, C] (1-D: H = 1), ALREADY normalised / activated (bg_im2col with a 1x1
 *         window does GroupNorm + SiLU/GELU + the cast in one pass); C % 64 == 0, C / 64 a power of two
 *   'same' convolution, stride 1, window kh x kw (odd), on the nearest-upsampled grid (H << up, W << up) = (Ho, Wo), both
 *         powers of two; zero padding outside (padded taps read zero_page: one pixel = 2 * C bytes of zeros on the device)
 *   w     [N, kh*kw*C] 16-bit, tap-major (tap = ky*kw + kx, then channel), N % 128 == 0;  bias fp32 [N] or NULL
 *         -- or a NARROW output, N < 128 (the decoders' conv_out, network.py:786-858 / 948-1040: 3 channels): w [128, kh*kw*C] and
 *         bias [128] zero-padded by the caller, no residual; one 128-column tile per row panel, only the N real columns stored
 *   out   fp32 [S*Ho*Wo, N] (row stride ldc >= N) = conv + bias (+ add: fp32 residual rows, row stride ld_add)
 * Needs at least 64 output tiles of 128 x 128 (it runs on the persistent kernel); smaller problems: im2col + GEMM. */
typedef struct {
    const void* x; int S, H, W, C;
    int kh, kw, up;
    const void* w; const float* bias; int N;
    float* out; int ldc;
    const float* add; int ld_add;
    int dtype;                 /* BG_BF16 | BG_F16: operand dtype of x and w */
    const void* zero_page;
} bg_conv_desc;
int bg_conv_gemm_fwd(const bg_conv_desc* d, bg_stream_t stream);

/* LayerNorm(768) of rows given as the split pair x = hi + lo (two 16-bit planes of dtype `dtype`) -> y (same dtype):
 * the denoisers' final net.norm when the residual stream is kept split. */
int bg_layernorm_split_fwd(const void* hi, const void* lo, const float* gamma, const float* beta, void* y, int dtype,
                           int M, float eps, bg_stream_t stream);

/* First half of an input-embedding MLP, fused: out = SiLU(LayerNorm(x W0^T + b0)), x fp32 [rows, lda] (k in {6,12,48}
 * leading columns used), exact-fp32 matrix-core arithmetic, out [rows,768] fp32 / bf16 / fp16 (sub-keys .0 .1 + SiLU
 * of p_embed, z_embed, surfp_embed, surfz_embed, edgep_embed, edgez_embed, vertp_fc: network.py:1080-1085 etc.).
 * w0_mfma: see bg_mlp_weights. */
int bg_embed_ln_silu_fwd(const float* x, int lda, int rows, int k, const float* w0_mfma, const float* b0,
                         const float* ln_g, const float* ln_b, void* out, int out_dtype, float eps, bg_stream_t stream);

/* F.scaled_dot_product_attention inside nn.MultiheadAttention (network.py:1076-1078 via
 * torch/nn/modules/transformer.py slow path): qkv packed [B*N, 2304] (q|k|v, head h at columns
 * 64h..64h+63 of each third; q already multiplied by 1/8), key_pad uint8 [B,N] (1 = padded key, -inf)
 * or NULL; out [B*N,768].  dtype BG_BF16 or BG_F32.  A sample whose keys are all padded yields 0
 * (the reference yields NaN there; the pipeline never produces such a sample, sample.py:163,261). */
int bg_attn_fwd(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype,
                bg_stream_t stream);
/* The same over a COMPACTED batch (variable-length execution): offsets int32 [B+1] on the device, sample b owns rows
 * offsets[b] .. offsets[b+1]-1 of qkv / out (at most N of them, every one a valid key; key_pad must then be NULL).
 * offsets == NULL is bg_attn_fwd. */
int bg_attn_varlen_fwd(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype,
                       const int* offsets, bg_stream_t stream);
/* QKV projection with the LayerNorm fold + self-attention in ONE launch (csrc/qkv_attn.hip): sequences of an even length
 * N <= 64, all B of them equally long; key_pad (may be NULL) as bg_attn_fwd's.  x_hi [B*N, 768] raw 16-bit rows, w_qkv [2304, 768] /
 * bias / colsum as bg_gemm_ex_fwd's fold operands, stats_in [12][B*N][2] the row statistics partials of x; out [B*N, 768] = what
 * bg_gemm_ex_fwd followed by bg_attn_fwd produce, bit for bit.  qkv_dbg (tests; may be NULL): [B*N, 2304] receives the q|k|v of
 * that GEMM. */
int bg_qkv_attn_fwd(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in,
                    const uint8_t* key_pad, void* out, void* qkv_dbg, int B, int N, int dtype, float ln_eps, bg_stream_t stream);
/* FFN1 (LayerNorm fold, ReLU) + FFN2 (split residual in place + row statistics) of one encoder layer in ONE launch
 * (csrc/ffn_fused.hip; network.py:1076-1078, dim_feedforward = 1024): the residual planes x_hi / x_lo [M, 768] (16-bit, dtype) are
 * updated in place, stats [12][m_stride][2] holds the row-statistics partials of x on entry and those of the new x on return;
 * w1_frag / w2_frag: bg_layer_weights.w_1f / w_2f; b1 / colsum1: the fold's bias and column sums [1024]; b2 [768].  m_dev (may be
 * NULL): device-side row count <= M.  Bit for bit what bg_gemm_ex_fwd (fold, ReLU) followed by bg_gemm_ex_fwd (split residual +
 * statistics) produce. */
int bg_ffn_fused_fwd(void* x_hi, void* x_lo, float* stats, const void* w1_frag, const float* b1, const float* colsum1,
                     const void* w2_frag, const float* b2, int M, int m_stride, const int* m_dev, int dtype, float ln_eps,
                     bg_stream_t stream);
/* The same launch on a SLOT-PACKED ragged batch (bg_compact_rows_paired): *m_dev rows (device-side, a multiple of 64) in 64-row
 * slots of one or two whole samples, slot_desc[2 k] / [2 k + 1] their lengths, at most slot_bound slots; m_stats = row stride of
 * stats_in.  out rows = what bg_gemm_ex_fwd + bg_attn_varlen_fwd produce for the same samples on the dense packing, bit for bit.
 * Invariants the kernel TRUSTS (bg_compact_rows_paired guarantees them; nothing on the device re-checks): *m_dev <= 64 * slot_bound
 * and a multiple of 64; slot_desc[2 k] >= 1, slot_desc[2 k] + slot_desc[2 k + 1] <= 64 for every slot below *m_dev / 64;
 * slot_bound <= 43 690 (32-bit row offsets; rejected above). */
int bg_qkv_attn_paired_fwd(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in,
                           void* out, void* qkv_dbg, const int* m_dev, const int* slot_desc, int slot_bound, int m_stats,
                           int dtype, float ln_eps, bg_stream_t stream);
/* The tail of the denoisers' output MLP in one launch (csrc/out_tail.hip): out[m, :n_out] = W3 . SiLU(LayerNorm(t0[m, :])) + b3 with
 * t0 [rows, 768] 16-bit rows of `dtype`, w3 [n_out_pad, 768] of the same dtype (the first 16 * ceil(n_out / 16) rows are read),
 * b3 fp32 [n_out_pad], 1 <= n_out <= 48, out fp32 [rows, n_out]. */
int bg_ln_silu_out_fwd(const void* t0, const float* ln_g, const float* ln_b, const void* w3, const float* b3, float* out,
                       int n_out, int n_out_pad, int rows, int dtype, float eps, bg_stream_t stream);
/* Slot-packed compaction for samples of at most 64 tokens (mask [B, n_mask], 1 = padded): every 64-row slot holds one or two whole
 * samples (shortest joins longest while the sum fits), the rows behind them are clones of the slot's first row.  offsets [B + 1]:
 * first row of every sample, offsets[B] = 64 * slots (stays on the device); src_row [64 B]; slot_desc [2 B]; slot_a [B]: the first
 * sample of every slot; counts [B]: scratch. */
int bg_compact_rows_paired(const uint8_t* mask, int B, int n_mask, int* offsets, int* src_row, int* slot_desc, int* slot_a,
                           int* counts, bg_stream_t stream);
/* Valid-token compaction behind the variable-length execution (csrc/compact.hip): mask uint8 [B, n_mask] (1 = padded),
 * each entry covering `rep` consecutive tokens (EdgePosNet: one entry per face, rep = E).  Writes offsets int32 [B+1]
 * (offsets[B] = number of valid tokens; it stays on the device) and src_row int32 [B*n_mask*rep]: the padded-layout index of
 * every compact row, in order (entries past offsets[B] are not written). */
int bg_compact_rows(const uint8_t* mask, int B, int n_mask, int rep, int* offsets, int* src_row, bg_stream_t stream);

/* ---- whole denoiser -------------------------------------------------------------------- */

typedef struct {            /* Linear(k_in,768) -> LayerNorm -> SiLU -> Linear(768,n_out)   (sub-keys .0 .1 .3) */
    const void* w0;         /* [768,k_in]; fp32 for the input/time embeds (tiny K or M: exact fp32 MFMA),
                               compute dtype for fc_out (K = 768, M = all tokens) */
    const float* b0;
    const float* ln_g;
    const float* ln_b;
    const void* w3;         /* compute dtype [n_out_pad,768] */
    const float* b3;        /* fp32 [n_out_pad] */
    int k_in, n_out, n_out_pad, w0_dtype;
    const float* w0_mfma;   /* optional, k_in in {6,12,48} with fp32 w0: W0 in MFMA operand order,
                               [24][k_in/2][64] with element (ct, kk, lane) = w0[ct*32 + (lane & 31)][2*kk + (lane >> 5)]:
                               selects the fused Linear + LayerNorm + SiLU kernel (bg_embed_ln_silu_fwd) */
    const float* w0_colsum; /* fc_out only (16-bit modes with the LayerNorm fold; NULL elsewhere): the encoder's final LayerNorm
                               (net.norm) is folded into fc_out.0 -- w0 = T(net.norm.weight * W0), b0 = b0 + W0 net.norm.bias,
                               w0_colsum[n] = sum_k float(w0[n,k]) -- and the rest of the MLP (LayerNorm + SiLU + Linear(768, n_out))
                               runs as one launch (bg_ln_silu_out_fwd) */
} bg_mlp_weights;

typedef struct {            /* one nn.TransformerEncoderLayer (norm_first) */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    const void* w_qkv;      /* [2304,768], q rows pre-scaled by 1/8 */
    const float* b_qkv;     /* q part pre-scaled */
    const void* w_o;        /* [768,768]  */
    const float* b_o;
    const void* w_1;        /* [1024,768] */
    const float* b_1;
    const void* w_2;        /* [768,1024] */
    const float* b_2;
    /* LayerNorm fold (16-bit compute dtypes; both NULL = unfolded: the two LayerNorms run as kernels).  When set,
     * w_qkv = T(norm1.weight * W_in) and b_qkv = b_in + W_in norm1.bias (q parts pre-scaled), likewise w_1 / b_1 with
     * norm2, and *_colsum[n] = sum_k float(w[n,k]) of the ROUNDED folded weights; the GEMM then takes the raw 16-bit
     * residual rows and applies mean / rstd per row in its epilogue. */
    const float* qkv_colsum; /* fp32 [2304] */
    const float* w1_colsum;  /* fp32 [1024] */
    /* optional (ABI 6; both NULL = FFN1 and FFN2 stay two launches): the SAME folded w_1 / w_2 once more, in MFMA fragment order for
     * bg_ffn_fused_fwd (one launch per layer for FFN1 + ReLU + FFN2 + residual, the hidden tensor never leaves the CU):
     *   w_1f[w][s][j][lane][e] = w_1[(4 w + j) * 32 + (lane & 31)][16 s + 8 (lane >> 5) + e]   w < 8, s < 48, j < 4, lane < 64, e < 8
     *   w_2f[w][s][j][lane][e] = w_2[(3 w + j) * 32 + (lane & 31)][16 s + 8 (lane >> 5) + e]   w < 8, s < 64, j < 3 */
    const void* w_1f;
    const void* w_2f;
} bg_layer_weights;

typedef struct {
    int net;                /* enum bg_net */
    int dtype;              /* GEMM/attention operand dtype: BG_BF16 or BG_F32 */
    int n_layer;            /* 12 */
    int _pad;
    bg_layer_weights layers[BG_MAX_LAYERS];
    const float *lnf_g, *lnf_b;            /* net.norm */
    bg_mlp_weights time_embed;
    bg_mlp_weights fc_out;
    /* SurfPos: {p_embed}; SurfZ: {z_embed,p_embed}; EdgePos: {surfp,surfz,edgep};
     * EdgeZ: {surfp,surfz,edgep,edgez,vertp} */
    bg_mlp_weights embed[BG_MAX_EMBEDS];
    const float* class_embed;              /* fp32 [11,768] or NULL (use_cf=False) */
    /* optional: the time-embedding MLP (sincos -> time_embed) evaluated once for t = 0 .. time_table_rows - 1, fp32
     * [time_table_rows, 768] -- a function of the weights only; with it an evaluation looks its timestep(s) up instead of running
     * the five small launches of that MLP (a timestep outside the table yields NaN).  NULL: computed per call. */
    const float* time_table;
    int time_table_rows;
    int _pad3;
} bg_denoiser_weights;

typedef struct {
    int B, S, E;            /* E = 1 for the surface nets; tokens per sample N = S*E */
    int n_timesteps;        /* 1 (sampling) or B (training-style per-sample t) */
    const float* x;         /* noisy input: SurfPos [B,S,6] | SurfZ [B,S,48] | EdgePos [B,S,E,6] | EdgeZ [B,S,E,18] */
    const float* surf_pos;  /* [B,S,6]   (SurfZ, EdgePos, EdgeZ) */
    const float* surf_z;    /* [B,S,48]  (EdgePos, EdgeZ) */
    const float* edge_pos;  /* [B,S,E,6] (EdgeZ) */
    const uint8_t* mask;    /* 1 = padded. SurfZ/EdgePos: [B,S]; EdgeZ: [B,S,E]; SurfPos: NULL */
    const int64_t* timesteps;
    const int64_t* class_label;  /* [B] or NULL */
    float* cond_cache;      /* optional [B*S,768] fp32: step-invariant conditioning embeds */
    int cond_cache_valid;   /* 1: reuse cond_cache, 0: (re)compute and store if cond_cache != NULL */
    /* Variable-length execution (needs mask): the valid tokens of every sample are packed into consecutive rows, all
     * GEMM / LayerNorm work runs on sum(valid) rows, attention runs per sample over its own rows, eps_out is the padded
     * layout with 0 at padded positions.  Valid positions are unchanged up to rounding (a padded token is never a key
     * of a valid query: network.py:1196, 1283, 1390); the reference computes -- and discards, sample.py:284, 307-314 --
     * values at padded positions.  The row count stays on the device: no host synchronisation. */
    int varlen;
    /* Host-side ESTIMATE of the number of valid tokens / of the sum over samples of valid^2 (variable-length execution; 0 =
     * unknown).  The row count that the kernels use is always the one counted on the device; the estimate only (a) lets the GEMM
     * launcher choose between "256 x 256 + 128 x 128 kernels, split rule evaluated on the device" and "128 x 128 kernel alone"
     * -- either choice is correct for ANY actual count, and the results are bit-identical -- and (b) is what the opt-in profiler
     * (bg_profile_*) books as executed FLOPs / bytes.  With 0 the launcher plans for the bound B * S * E and the profiler books
     * the padded sizes. */
    double rows_hint, pairs_hint;
    /* Software pipelining over independent sub-batches: n_split in 2..4 cuts the batch into that many contiguous
     * groups of samples and runs their forwards concurrently -- the first on `stream`, the others on helper streams
     * forked from it and joined back into it by events before the call returns (HIP-graph capturable).  Every op of the
     * path is per-sample and every kernel bit-stable across batch sizes, so the result is identical; what it buys is
     * occupancy: the tile-round tails and the memory-bound epilogues of one group hide under the K loops of another.
     * 0 / 1 = off; negative values are rejected.  The helper streams and events are created lazily, once per DEVICE (a process
     * that drives several devices gets one set each), and used under a mutex (enqueue only). */
    int n_split;
    int _pad2;
    /* Optional (variable-length execution; 0 = not known): the row count the kernels of sample group k WILL see -- the valid tokens of
     * the group, or 64 x its slots where the batch runs slot-packed -- when the caller knows it exactly ([0] alone without n_split).
     * Like rows_hint it only chooses between launch plans that are all correct for any actual count (the device still counts); what an
     * exact count buys is launches that are not made: a GEMM whose rows all belong to the 256 x 256 kernel is ONE launch instead of that
     * kernel + a 128 x 128 tail kernel that finds nothing to do (4-5 us each, twice per encoder layer on the face LDM's ragged loop). */
    double rows_plan[4];
} bg_denoiser_inputs;

/* 1 when a variable-length call of bg_denoiser_fwd with these shapes runs SLOT-PACKED (64-row slots of one or two samples, the fused QKV +
 * attention launch): the library's own predicate (net, shapes, 16-bit dtype, LayerNorm-fold weights given = `fold`, bg_tune key 13),
 * exported so that a host that fills bg_denoiser_inputs.rows_plan counts the rows of the layout the library will actually use.
 * rows_plan only ever chooses between launch plans that are all correct for any row count: it must never be used to skip work. */
int bg_slot_packing_applies(int net, int B, int S, int E, int dtype, int fold);

/* bytes of scratch bg_denoiser_fwd needs for these shapes (sized so that any n_split <= 4 fits) */
size_t bg_workspace_bytes(int net, int B, int S, int E, int dtype);

/* {SurfPosNet,SurfZNet,EdgePosNet,EdgeZNet}.forward -- network.py:1107-1126,1176-1200,1257-1286,
 * 1357-1393.  eps_out fp32, shaped like in->x. */
int bg_denoiser_fwd(const bg_denoiser_weights* w, const bg_denoiser_inputs* in, float* eps_out,
                    void* workspace, size_t workspace_bytes, bg_stream_t stream);

/* Pieces of the whole-net call on their own (same kernels, caller-owned 256-byte-aligned scratch):
 *   bg_embed_mlp_fwd      Linear(k,768) -> LayerNorm -> SiLU -> Linear(768,n) (+ add[m / add_div]) of one bg_mlp_weights
 *                         (network.py:1080-1099); x fp32 rows (input embeds) or `dtype` rows (fc_out); out fp32.
 *   bg_encoder_layer_fwd  one nn.TransformerEncoderLayer(768, 12, norm_first=True, dim_feedforward=1024) (network.py:
 *                         1076-1078) on an fp32 residual stream x [B*N,768], in place; key_pad uint8 [B,N] or NULL.  Takes
 *                         UNFOLDED layer weights (qkv_colsum == w1_colsum == NULL): the LayerNorm-folded, split-residual
 *                         formulation of the 16-bit modes keeps state between layers and lives inside bg_denoiser_fwd. */
size_t bg_embed_mlp_scratch_bytes(int rows, int dtype);
int bg_embed_mlp_fwd(const bg_mlp_weights* m, int dtype, const void* x, int lda, int rows, float* out, int ldc,
                     const float* add, int ld_add, int add_div, void* scratch, size_t scratch_bytes, bg_stream_t stream);
size_t bg_encoder_layer_scratch_bytes(int B, int N, int dtype);
int bg_encoder_layer_fwd(const bg_layer_weights* L, int dtype, float* x, const uint8_t* key_pad, int B, int N,
                         void* scratch, size_t scratch_bytes, bg_stream_t stream);

/* ---- scheduler steps (diffusers==0.27 arithmetic, called at sample.py:137,153,202,222,236,282) */

/* DDPMScheduler.step fused with the classifier-free-guidance combine (sample.py:132-134):
 *   eps = eps_c*(1+w) - eps_u*w   (eps_u NULL -> eps = eps_c)
 *   x0  = clamp((x - sqrt_beta_prod*eps) / sqrt_alpha_prod, +-clip)   (clip <= 0: no clamp)
 *   out = x0_coeff*x0 + xt_coeff*x + sigma*noise                      (noise NULL or sigma==0: skipped)
 * scalars are computed on the host in fp32 (brepgen_amd.schedulers).  n = element count. */
int bg_cfg_ddpm_step(const float* eps_c, const float* eps_u, float guidance_w, const float* x,
                     const float* noise, float* out, size_t n, float sqrt_alpha_prod,
                     float sqrt_beta_prod, float x0_coeff, float xt_coeff, float sigma, float clip,
                     bg_stream_t stream);

/* PNDMScheduler.step (PRK + PLMS) as one fused pass.
 *   e      = eps_c*(1+w) - eps_u*w
 *   if store_e:  e_store = e                       (PRK phase 0 / every PLMS step: the `ets` history)
 *   comb   = c_e*e + c_acc*acc + sum_i c_hist[i]*hist[i]
 *   if acc_mode==1: acc_out = acc_scale_old*acc + acc_scale_e*e    (PRK running cur_model_output)
 *   out    = sample_coeff*x - eps_coeff*comb
 */
int bg_pndm_step(const float* eps_c, const float* eps_u, float guidance_w, const float* x,
                 float* e_store, const float* acc, float* acc_out, float acc_scale_old,
                 float acc_scale_e, float c_e, float c_acc, const float* hist0, const float* hist1,
                 const float* hist2, float c_h0, float c_h1, float c_h2, float sample_coeff,
                 float eps_coeff, float* out, size_t n, bg_stream_t stream);

/* ---- VAE steps (AutoencoderKLFastDecode / AutoencoderKL1DFastDecode / ...FastEncode, network.py:690-1040; blocks from
 * diffusers==0.27).  Channels-last fp32 activations [S, H, W, C] (1-D: H = 1).  A convolution is either bg_im2col with a
 * 1x1 window (GroupNorm + activation + cast) followed by the implicit GEMM bg_conv_gemm_fwd, or bg_im2col over the whole
 * window followed by bg_gemm_bias_act_fwd, on weights reshaped to [C_out, kh*kw*C_in]; bg_vae_run (below) strings a whole
 * pass together. ------------------------------------------------------------------------------------------------- */

/* nn.GroupNorm statistics: stats[s, g] = (mean, 1/sqrt(var + eps)) over the P positions x C/G channels of group g. */
int bg_groupnorm_stats(const float* x, float* stats /*[S,G,2]*/, int S, int P, int C, int G, float eps,
                       bg_stream_t stream);

/* im2col of a kh x kw convolution with `stride`, over the logical input grid (Hin << up, Win << up), producing
 * an Ho x Wo output grid; `pad_y` / `pad_x` zeros precede the first row / column, whatever the window needs beyond
 * the last row / column is zero as well (covers pad=1 convs, and Downsample2D's F.pad(0,1,0,1) + stride 2).  With
 *   - optional nearest x2 up-sampling of the source (Upsample2D) folded in (up = 1),
 *   - optional GroupNorm (stats from bg_groupnorm_stats, gamma, beta) and activation (0 none, 1 SiLU, 2 GELU-erf)
 *     applied on the fly (ResnetBlock2D / ResConvBlock / conv_norm_out),
 *   - optional residual `add` [S*Ho*Wo, C] added after the activation (1x1 window only: the tail of ResConvBlock).
 * out: [S*Ho*Wo, kh*kw*C] fp32 / bf16 / fp16, tap-major / channel-minor. */
int bg_im2col(const float* x, void* out, int out_dtype, int S, int Hin, int Win, int C, int kh, int kw, int up,
              int stride, int pad_y, int pad_x, int Ho, int Wo, const float* stats, const float* gamma,
              const float* beta, int G, int act, const float* add, bg_stream_t stream);

/* diffusers Upsample1d("cubic") (network.py:43): x [S,L,C] -> y [S,2L,C], reflect pad + 8-tap transposed conv. */
int bg_upsample1d_cubic(const float* x, float* y, int S, int L, int C, bg_stream_t stream);

/* diffusers Downsample1d("cubic") of the edge encoder (network.py:86-185 via get_down_block): [S,L,C] -> [S,L/2,C]. */
int bg_downsample1d_cubic(const float* x, float* y, int S, int L, int C, bg_stream_t stream);

/* Self-attention of the VAE mid blocks (diffusers Attention with 1 head over 16 tokens; SelfAttention1d with
 * C/32 heads over 4 tokens): qkv fp32 [S*T, ld] with q|k|v at columns 0, C, 2C; out [S*T, C] fp32 or bf16. */
int bg_small_attn(const float* qkv, int ld, void* out, int out_dtype, int S, int T, int C, int nh, float scale,
                  bg_stream_t stream);

/* ---- a whole VAE pass as ONE call (bg_vae2d_decode / bg_vae1d_decode; the encoders too).
 * Replaces AutoencoderKLFastDecode / AutoencoderKL1DFastDecode / ...FastEncode.forward (network.py:690-1040): the four
 * networks are the same few steps in different orders, so there is one interpreter and the caller hands it the
 * network as a flat program (brepgen_amd/vae.py builds it once per module and dtype from the state dict).  Every
 * launch of the pass is enqueued on `stream` by this call; nothing is allocated and nothing synchronises.
 *
 * Activations live in SLOTS: fp32 channels-last [S, H, W, C] (1-D: H = 1).  Slot 0 is the caller's input, slot
 * BG_VAE_OUT the caller's output, slots 1 .. n_slots-1 are carved from the workspace (all the same size: the largest
 * activation of the program).  A step reads `src` (+ optional residual `res`) and writes `dst` (dst != src, res).
 *   BG_VOP_CONV          [GroupNorm (+act) ->] conv kh x kw [on the nearest-x2 up-sampled grid: up = 1] [stride 2] + bias
 *                        [+ res].  pad_mode 0: "same" zero padding (kh/2, kw/2); 1: Downsample2D's F.pad(0,1,0,1), no other.
 *                        w: [n_pad, kh*kw*C_in] tap-major / channel-minor, dtype w_dtype; columns < n_out are written.
 *                        Runs as the implicit GEMM bg_conv_gemm_fwd when that kernel's shape rules hold, else
 *                        bg_im2col + bg_gemm_bias_act_fwd -- the same choice for the same shapes, so results do not
 *                        depend on how the batch is chunked beyond which of the two a chunk size selects.
 *   BG_VOP_NORM_ACT_ADD  dst = act(GroupNorm(src)) + res                     (the tail of diffusers' ResConvBlock)
 *   BG_VOP_ATTN          dst = src + proj(softmax(q k^T * scale) v), q|k|v = GroupNorm(src) w^T + bias ([3C, C] fused),
 *                        per sample over its H*W tokens with `heads` heads; w2 / bias2 = the output projection
 *   BG_VOP_UP1D / DOWN1D bg_upsample1d_cubic / bg_downsample1d_cubic
 * The batch is processed in chunks of `chunk` samples; bg_vae_workspace_bytes(n, chunk) is the workspace that takes. */
enum bg_vae_opcode { BG_VOP_CONV = 0, BG_VOP_NORM_ACT_ADD = 1, BG_VOP_ATTN = 2, BG_VOP_UP1D = 3, BG_VOP_DOWN1D = 4 };
enum bg_vae_act { BG_VACT_NONE = 0, BG_VACT_SILU = 1, BG_VACT_GELU = 2 };
#define BG_VAE_MAX_SLOTS 8
#define BG_VAE_OUT 255
typedef struct {
    int op;                    /* bg_vae_opcode */
    int src, dst, res;         /* slots; res = -1: none */
    int kh, kw, up, stride, pad_mode;
    int n_out, n_pad, w_dtype; /* CONV: output channels, rows of w, BG_F32 | BG_BF16 | BG_F16.  ATTN: n_pad / w_dtype of the q|k|v weights */
    const void* w;
    const float* bias;
    const float* gn_gamma;     /* NULL: no GroupNorm in front */
    const float* gn_beta;
    int gn_groups;
    float gn_eps;
    int act;                   /* bg_vae_act, applied after the GroupNorm */
    int heads;                 /* ATTN */
    float scale;
    int n_pad2, w2_dtype;
    const void* w2;
    const float* bias2;
} bg_vae_op;
size_t bg_vae_workspace_bytes(const bg_vae_op* ops, int n_ops, int n_slots, int in_h, int in_w, int in_c, int n, int chunk);
/* x: [n, in_h, in_w, in_c] fp32; out: [n, Ho, Wo, C_out] fp32 (the shape of the step that writes BG_VAE_OUT);
 * zero_page: >= 2 * max C bytes of zeros (see bg_conv_desc); workspace 256-byte aligned. */
int bg_vae_run(const bg_vae_op* ops, int n_ops, int n_slots, int in_h, int in_w, int in_c, const float* x, int n, int chunk,
               float* out, const void* zero_page, void* workspace, size_t workspace_bytes, bg_stream_t stream);

/* ---- bbox de-duplication between the cascade stages, on the device (the reference does it on the host in numpy:
 * sample.py:159-183 faces, 242-261 edges).  Same greedy order-dependent algorithm in float32, incl. the
 * corner-swapped match and (faces) np.round(x, 4); decisions are bit-identical to the numpy code. */
int bg_dedup_surfaces(const float* surf_pos /*[B,S,6]*/, float threshold, float* pos_out /*[B,S,6] kept, 0-padded*/,
                      uint8_t* mask_out /*[B,S] 1 = padding*/, int B, int S, bg_stream_t stream);
int bg_dedup_edges(const float* edge_pos /*[B,S,E,6]*/, const uint8_t* surf_mask /*[B,S]*/, float threshold,
                   uint8_t* edge_mask /*[B,S,E] 1 = padded face or duplicate edge*/, int B, int S, int E,
                   bg_stream_t stream);

/* ---- measurement aid (bench.py's roofline leg): hipEvent pairs around every kernel launch ------------
 * bg_profile_begin allocates up to max_launches event pairs and switches recording on (this is the one
 * place the library owns state; it is off by default and costs nothing when off).  bg_profile_end
 * synchronises, aggregates per kernel and frees the events.  flops / bytes are the ALGORITHMIC counts the
 * launcher derives from the shapes (DESIGN.md "roofline accounting"). */
typedef struct {
    const char* kernel;     /* static string */
    int launches;
    double total_ms;        /* sum of launch durations */
    double flops;           /* sum of algorithmic FLOPs */
    double bytes;           /* sum of algorithmic HBM bytes */
} bg_profile_row;
int bg_profile_begin(int max_launches);
int bg_profile_end(bg_profile_row* rows, int max_rows);   /* returns the number of rows written (<0: error) */

/* ---- the path's one collective (batch sharding + ONE all-gather of the finished latents) ----
 * recv[r * bytes_per_rank ...] = rank r's `send` (bytes_per_rank bytes each; device pointers; every rank passes the same size -- the
 * Python host pads the last shard, brepgen_amd/sampling.py: gather_latents), enqueued on `stream`: a thin ncclAllGather (RCCL over
 * xGMI) on a communicator the CALLER created with ncclCommInitRank (`nccl_comm` = the ncclComm_t).  For hosts without
 * torch.distributed; the Python host uses all_gather_into_tensor on the "nccl" (= RCCL) backend, which is the same collective.
 * RCCL is dlopen'ed at the first call; returns 0, BG_E_ARG, or 1000 + ncclResult_t. */
int bg_allgather(const void* send, void* recv, size_t bytes_per_rank, void* nccl_comm, bg_stream_t stream);

/* Kernel-selection knobs.  Every choice computes bit-identical results: the keys exist so that the parity tests can run the same
 * GEMM on each kernel that may serve it (0 always = the library's own choice; process-global, not for production use).
 *   key  8  split-residual launches on the 256 x 256 kernel: start delay of the second phase group, x 1024 cycles (< 0: none)
 *   key 10  256 x 256 persistent kernel: 1 = alone wherever eligible, 2 = never
 *   key 12  split-residual GEMMs of the encoder layers: 1 = the 128 x 128 persistent kernel instead of the pipelined one
 *   key 13  QKV + attention of short, equally long sequences: 1 = as two launches (GEMM, attention) instead of the fused kernel,
 *           2 = fused wherever eligible (the library's own choice leaves a nearly empty second round of tiles to the two launches)
 *   key 14  tile walk of that fused kernel: 1 = plain (every XCD runs all 12 heads), 2 = XCD-pinned head halves wherever the grid
 *           allows (the library's own choice: from two rounds of tiles on)
 *   key 15  small launches: tile-count threshold of the 64 x 64-tile path (< 0: off)
 *   key 16  FFN1 / FFN2 of layers that carry w_1f / w_2f: 1 = as two GEMM launches instead of the fused launch (bg_ffn_fused_fwd)
 *   keys 17 / 18  measurement aid: low / high 32 bits of a device address that receives s_memtime stamps of the fused FFN launch's
 *           phases (tools/ffn_fused_bench.py stamps); 0 / 0 = off
 *   key 19  bg_vae_run, GroupNorm(1, C) over samples of 2048 / 4096 values (the 1-D VAE's blocks): 1 = statistics pass + gather as two
 *           launches (bg_groupnorm_stats, bg_im2col) instead of the one-pass kernel
 *   key 20  long-sequence attention: 1 = each XCD walks a contiguous eighth of the (sample, head) units (rounds 2-5) instead of the
 *           units going round-robin over the XCDs (ragged batches: the eighths' shares of sum n_b^2 differ, the launch lasts as long
 *           as the heaviest) */
int bg_tune_set(int key, int value);

/* How a 16-bit GEMM launch of `rows` x `n_cols` (n_cols a multiple of 256) is partitioned between the 256 x 256
 * persistent kernel (rows [0, return value)) and the 128 x 128 one (the rest): the rule both kernels evaluate on the
 * device-side row count and the launcher evaluates on the host ("tile-round quantisation").
 * split_residual: the out-proj / FFN2 epilogue; concurrent: launches of sibling sample groups are in flight
 * (all-or-nothing).  No device work; usable without a GPU.  <0: error code. */
int bg_gemm_p256_rows(int rows, int n_cols, int split_residual, int concurrent);

/* DDPMScheduler.add_noise (training-time forward diffusion, trainer.py:348,399,515,...):
 *   out[b,:] = sqrt_alpha_prod[b] * x0[b,:] + sqrt_one_minus_alpha_prod[b] * noise[b,:]
 * the two per-sample scalar vectors are device fp32 [B] (gathered from alphas_cumprod by the host shim). */
int bg_add_noise(const float* x0, const float* noise, const float* sqrt_alpha_prod,
                 const float* sqrt_one_minus_alpha_prod, float* out, int B, size_t per_sample,
                 bg_stream_t stream);

/* Ancestral noise of the DDPM loops (the reference draws it on the device with the global RNG, sample.py:144-153 ->
 * diffusers randn_tensor(device=...)): out[b, e] ~ N(0,1) for samples b = 0 .. n_samples-1 of `per_sample` elements,
 * Philox4x32-10 with key = seed, counter = (e / 4, first_sample + b, draw_id, tag) and Box-Muller on the four words.
 * The value of an element depends only on (seed, draw_id, GLOBAL sample index, e): a rank that owns samples
 * [lo, hi) of a sharded batch passes first_sample = lo and reproduces exactly those rows of the single-GPU draw.
 * raw_bits != 0 writes the four 32-bit words themselves (bit pattern in the float slots) -- used by the parity test. */
int bg_philox_randn(float* out, long long n_samples, int per_sample, unsigned long long seed, unsigned draw_id,
                    long long first_sample, int raw_bits, bg_stream_t stream);

/* Device loop of joint_optimize (utils.py:746-772): per-face 3-D offsets fitted with `iters` AdamW steps (torch.optim.AdamW
 * arithmetic; the reference uses lr 1e-3, betas (0.95, 0.999), weight_decay 1e-6, eps 1e-8, 200 iterations) on
 *   L = mean_f sum_{e in edges_f} min_{s in surf_f} |e - (s + off_f)|^2        (chamferdist ChamferDistance(reverse=True)).
 * surf [F,P,3] (P <= 4096), edge_pts [edge_off[F],3] with face f owning rows edge_off[f] .. edge_off[f+1]-1 (int32, device).
 * offsets_out [F,3] and surf_out [F,P,3] (optional) are those of the LAST evaluated iteration (what the reference returns,
 * utils.py:770); loss_out [F] (optional) the per-face Chamfer sums of that iteration.  One launch, deterministic. */
int bg_chamfer_offset_fit(const float* surf, const float* edge_pts, const int* edge_off, int F, int P, int iters, double lr,
                          double beta1, double beta2, double weight_decay, double eps, float* offsets_out, float* surf_out,
                          float* loss_out, bg_stream_t stream);

/* Masked MSE of the trainers' loss / validation forward (trainer.py:354, 538, 597, 950-952; `loss_fn(pred[~mask],
 * noise[~mask])`): pred / target fp32 [rows, ld], row_mask uint8 [rows] (1 = padded, skipped) or NULL, columns
 * [col0, col0+ncols).  scratch: 1024 doubles.  out3 (device): {mean over valid elements, sum over valid rows of the
 * per-row mean (the reduction of test_val, trainer.py:597), number of valid rows}.  Deterministic (no atomics). */
int bg_masked_mse(const float* pred, const float* target, const uint8_t* row_mask, long long rows, int ld, int col0,
                  int ncols, double* scratch, float* out3, bg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
