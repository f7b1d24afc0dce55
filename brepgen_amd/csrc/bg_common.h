// Shared device/host helpers for libbrepgen_hip.so (gfx950 only -- wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/brepgen_hip.h"

namespace bg {

// ---- error plumbing (thread-local message behind bg_last_error) ---------------------------------
void set_error(const char* fmt, ...);
int launch_status(const char* what);   // hipGetLastError -> 0 / positive hipError_t (+message)

#define BG_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            ::bg::set_error(__VA_ARGS__);      \
            return (code);                     \
        }                                      \
    } while (0)

// ---- optional per-kernel timing (bg_profile_begin / bg_profile_end; off by default, zero cost when off) ----
enum ProfKernel { PK_GEMM_BF16_128 = 0,  /* persistent 128x128 kernel (gemm_bf16_p_kernel) */ PK_GEMM_BF16_64, PK_GEMM_F32, PK_ATTN_BF16, PK_ATTN_F32, PK_LAYERNORM,
                  PK_DDPM_STEP, PK_PNDM_STEP, PK_MISC, PK_EMBED, PK_GEMM_P256, PK_GEMM_SPLIT, PK_GEMM_P256_SPLIT, PK_QKV_ATTN, PK_OUT_TAIL, PK_FFN_FUSED, PK_COUNT };
extern bool g_prof_on;
void prof_pre(hipStream_t s);
void prof_post(int kernel, double flops, double bytes, hipStream_t s);
struct ProfScope {      // records a hipEvent pair around one launch when profiling is enabled
    int k; double f, b; hipStream_t s;
    ProfScope(int kernel, double flops, double bytes, hipStream_t st) : k(kernel), f(flops), b(bytes), s(st) {
        if (g_prof_on) prof_pre(s);
    }
    ~ProfScope() { if (g_prof_on) prof_post(k, f, b, s); }
};

// ---- tuning knobs (bg_tune_set; defaults are the shipped configuration) ----
// (8: phase-group delay of split-residual launches on the 256 x 256 kernel; 10: 256-kernel mode; 12: split-residual kernel choice;
//  13: 1 = QKV and attention as two launches even where the fused kernel (qkv_attn.hip) applies; 14: tile walk of that kernel (1 = plain, 2 = XCD-pinned head halves, 0 = by size); 15: small-launch threshold; 16: 1 = FFN1 and FFN2 as two launches even where w_1f / w_2f are given -- each backs a bit-equality test, see gemm_16bit.hip launch16)
enum TuneKey { TUNE_GEMM_STAGGER = 8, TUNE_P256_MODE = 10, TUNE_SPLIT_PIPE = 12, TUNE_QKV_ATTN = 13, TUNE_QKV_WALK = 14, TUNE_SMALL_TILES = 15, TUNE_FFN_FUSED = 16, TUNE_DEBUG_PTR_LO = 17, TUNE_DEBUG_PTR_HI = 18, TUNE_VAE_GN1_FUSED = 19, TUNE_ATTN_WALK = 20, TUNE_COUNT = 21 };
extern int g_tune[TUNE_COUNT];

// ---- vector types -----------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ bf16x4 to_bf16x4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (__bf16)a; r[1] = (__bf16)b; r[2] = (__bf16)c; r[3] = (__bf16)d;   // RNE (v_cvt_pk_bf16_f32)
    return r;
}

// 4 floats -> 4 packed 16-bit values of dtype `dt` (BG_BF16 | BG_F16), as a 64-bit payload
__device__ __forceinline__ uint2 pack4_16(float a, float b, float c, float d, int dt) {
    union { bf16x4 b; half4_t h; uint2 u; } r;
    if (dt == BG_F16) {
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));      // no fma + conversion fusion (v_fma_mixlo_f16: one rounding): see gemm16.h
        r.h[0] = (_Float16)a; r.h[1] = (_Float16)b; r.h[2] = (_Float16)c; r.h[3] = (_Float16)d;
    }
    else r.b = to_bf16x4(a, b, c, d);
    return r.u;
}

// fp32 -> 16-bit with the value made opaque first when the target is fp16: hipcc otherwise fuses a preceding multiply / fma with the
// conversion into v_fma_mixlo_f16 (ONE rounding instead of two) in some kernels and not in others, which breaks the bit-identity of
// kernels that compute the same thing (gemm16.h Elem<true>::pack4; seen in round 6 when the build flags changed)
template <typename T> __device__ __forceinline__ T cvt16(float x) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, __bf16)) asm volatile("" : "+v"(x));
    return (T)x;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each) for results that are rounded to 16 bits next: the IEEE division of silu_f is
// about ten VALU instructions per element, and the kernels that use this issue 768 of them per token (embed.hip, out_tail.hip)
__device__ __forceinline__ float silu_rcp(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// XCD-aware bijective remap of a 1-D block id (MI355X: block b runs on XCD b % 8, each XCD has a private
// 4 MiB L2).  Consecutive *logical* ids land on the same XCD back-to-back, so tiles that share an operand
// panel hit that XCD's L2 instead of 8 different ones.  Placement is a speed assumption only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// ---- launchers implemented in the kernel files (host side, all asynchronous on `s`) -----------
struct GemmArgs {
    const void* a; int lda;
    const void* w;            // [N_pad, K] row-major
    const float* bias;        // [N_pad] or null
    void* out; int ldc;
    int M, N, N_pad, K;
    int out_dtype;            // BG_F32, or the 16-bit operand dtype
    int act;                  // bg_act
    const float* add; int ld_add; int add_div;   // optional fp32 addend, row (m / add_div)
    const float* add2 = nullptr; int ld_add2 = 0; int add2_div = 1;   // optional second addend
    // ---- 16-bit operands only: split residual stream + LayerNorm fold ----
    // Split output: the fp32 result v is stored as two 16-bit planes hi = T(v) (-> out, ldc) and lo = T(v - hi)
    // (-> out_lo, ldc); hi is directly the next GEMM's A operand, hi + lo carries ~16 mantissa bits of v.
    void* out_lo = nullptr;
    const void* res_hi = nullptr; const void* res_lo = nullptr; int ld_res = 0;   // split residual addend rows [M, N]
    float* stats_out = nullptr;        // [N_pad/64][M][2] (part-major): per 64-column group (sum, sum of squares) of v
    // LayerNorm fold on the consumer: a = 16-bit rows x (NOT normalised), w = T(gamma * W), bias = b + W beta,
    // colsum[n] = sum_k w[n,k];  out = act(rstd_m * acc - mean_m * rstd_m * colsum[n] + bias[n]) with the row
    // statistics summed from stats_in [K/64][M][2].
    const float* stats_in = nullptr; const float* colsum = nullptr; float ln_eps = 1e-5f;
    // fp32 operands, M <= 8: allow the wave-per-column GEMV (tree reduction over K instead of the MFMA's k-ordered fma
    // chain -- last-bit different, so only callers whose M never depends on the batch composition set it: the
    // time-embedding MLP)
    int gemv_ok = 0;
    // ---- variable-length (compacted) batches: the number of rows actually present is only known on the device ----
    // m_dev != null: *m_dev (<= M) rows exist; M stays the host-side upper bound that sizes the grid and the part-major
    // statistics stride.  row_map[r] = index of compact row r in the PADDED token layout: addend rows (map_add / map_add2:
    // add[(row_map[r] / add_div)]) and output rows (map_out: out[row_map[r]]) are looked up through it.
    const int* m_dev = nullptr;
    const int* row_map = nullptr;
    int map_add = 0, map_add2 = 0, map_out = 0;
    double rows_hint = 0.0;           // host-side estimate of *m_dev (0 = unknown): kernel choice in launch16 + profiler accounting
    double rows_plan = 0.0;           // *m_dev as the caller knows it EXACTLY (0 = not known): kernel choice only (brepgen_hip.h: rows_plan)
    // ---- implicit-GEMM convolution (16-bit operands, persistent kernel only; csrc/gemm_16bit.hip) ----
    // cv_C > 0: `a` is a channels-last activation tensor [S, H, W, C] (already normalised / activated), row m of the GEMM is
    // output pixel (s, oy, ox) of the 'same' stride-1 convolution on the nearest-upsampled grid (H << up, W << up), and
    // K = kh * kw * C runs tap-major over the window; padded taps read `cv_zero` (one pixel: 2 * C bytes of zeros).
    int cv_C = 0, cv_H = 0, cv_W = 0, cv_kh = 1, cv_kw = 1, cv_up = 0;
    int cv_wo_log2 = 0, cv_ho_log2 = 0, cv_spt_log2 = 0;      // log2 of the output grid and of C / 64 (K-steps per tap)
    const void* cv_zero = nullptr;
    // ---- 256 + 128 hybrid (set by the launcher only): rows [0, p256_rows(*m_dev, N_pad / 256)) belong to the 256 x 256 kernel,
    // the rest to the 128 x 128 kernel -- both evaluate the same rule on the device-side row count ----
    int hybrid = 0;                   // 1: the split above; 2: all-or-nothing (concurrent sample groups, see p256_rows)
    // the number of rows the 256 x 256 kernel owns in a hybrid launch: known on the host (rows256_host) or, with a device-side row
    // count, read from the table compact_rows left behind (rule_table: set by the caller of gemm(); rows256_dev: the launch's entry)
    const int* rule_table = nullptr;
    const int* rows256_dev = nullptr;
    int rows256_host = 0;
    int p256_stagger = 0;             // split-residual launches of the 256 x 256 kernel: start delay of the second phase group, x 1024 cycles
    int concurrent = 0;               // caller's hint: other launches of the same kind are in flight on sibling streams
};

// Tile-round quantisation: one 256 x 256 tile per CU and round, so a launch whose tile count is a little
// above a multiple of 256 would spend a whole round on a few tiles.  The 256 kernel therefore takes the whole row panels that
// fill complete rounds (or everything when the last round is at least P256_TAIL_MIN tiles full); the 128 x 128 kernel -- four
// times finer, two workgroups per CU -- takes the remaining rows.  Evaluated on the host when the row count is known there, and
// otherwise ONCE on the device, by the compaction kernel that produces the row count (compact.hip: a table of P256_RULE_ENTRIES
// answers, one per (column tiles, split epilogue, all-or-nothing)); the GEMM kernels only read their entry.
constexpr int P256_TAIL_MIN = 160;
constexpr int P256_RULE_ENTRIES = 12;          // rule table: index = (column-tile class {3, 4, 9}) * 4 + split * 2 + all_or_nothing
__host__ __device__ inline int p256_rule_index(int nt_n256, bool split, bool all_or_nothing) {
    const int c = nt_n256 == 3 ? 0 : (nt_n256 == 4 ? 1 : (nt_n256 == 9 ? 2 : -1));      // 768 / 1024 / 2304 columns
    return c < 0 ? -1 : c * 4 + (split ? 2 : 0) + (all_or_nothing ? 1 : 0);
}
constexpr int P256_SPLIT_STAGGER = 24;         // start delay of the second phase group of a split-residual launch, x 1024 cycles ...
constexpr int P256_SPLIT_STAGGER_ROUNDS = 5;   // ... from this many rounds on (+1-3 % there, a loss below: profiles/r04/gemm_split_bench_sweep.log)
constexpr int P256_SPLIT_MIN_HALF_ROUNDS = 5;  // split-residual launches: the 256 x 256 kernel from 2.5 rounds of tiles on
// split-residual launches: the rule on whole rounds of Gs = (256 / nt_n256) * nt_n256 tiles (see p256_rows)
__host__ __device__ inline int p256_split_rows(int panels, int nt_n256, bool all_or_nothing) {
    const int q = 256 / nt_n256;                                  // row panels per round
    const int Gs = q * nt_n256, T = panels * nt_n256;
    if (T <= Gs) return T >= 200 ? panels << 8 : 0;             // one well-filled round (the compacted face batch: 204 tiles, 40 vs 45 us)
    if (2 * T < P256_SPLIT_MIN_HALF_ROUNDS * Gs) return 0;
    if (all_or_nothing) return panels << 8;
    const int R = panels / q;                                     // full rounds (T / Gs == panels / q)
    const int rem = T - R * Gs;
    if (rem == 0 || rem >= 200) return panels << 8;
    return (R * q) << 8;
}
__host__ __device__ inline int p256_rows(int rows, int nt_n256, bool split = false, bool all_or_nothing = false) {
    const int panels = (rows + 255) >> 8, T = panels * nt_n256;
    if (all_or_nothing) {
        // Several sample groups in flight on forked streams (n_split > 1): a partial round of one group's launch is filled by the
        // other group's, and a tail kernel only adds a dependent launch to each chain -- the 256 kernel runs alone or not at all
        // (measured, profiles/r03/face_ldm_legs_ab_p256.log: leg B 4.17 vs 4.41 ms with the tail kernels)
        if (split) return p256_split_rows(panels, nt_n256, true);
        // (non-split threshold: 300 vs 400 tiles measured on the compacted face batch, 2 x 306 QKV tiles in flight: -1.4 % per step,
        //  profiles/r03/face_ldm_legs_ab_fold_in_kernel.log)
        return (T >= 300 || (T >= 200 && T <= 256)) ? panels << 8 : 0;
    }
    if (split) {
        // split-residual launches (out-proj / FFN2; 8 B of residual traffic per output element next to 2 K FLOP): what bounds them
        // is the bytes a CU has to pull through its vector-memory path -- operands AND residual -- and a 256 x 256 tile needs the
        // fewest per FLOP; but its epilogue is a serial tail of every tile (nothing of the next tile's K loop fits beside 128
        // accumulator registers), so with few tiles per CU the round quantisation eats the advantage.  Measured
        // (profiles/r04/gemm_split_bench_sweep.log): from ~2.5 rounds of tiles on the 256 kernel wins (M = 61 440: 141 vs 163 us for
        // FFN2; M = 138 752: 311 vs 357 us), below that the pipelined 128 x 128 kernel (gemm_split.hip) does (M = 30 720: 74 vs
        // 85-89 us); the rows beyond the last full round go to the 128 kernel unless that round is well filled.  A round here is
        // Gs = the largest multiple of the column tiles <= 256 (gemm_p256.hip: phase groups keep whole row panels).
        return p256_split_rows(panels, nt_n256, false);
    }
    const int R = T >> 8, rem = T - (R << 8);
    if (rem == 0 || rem >= P256_TAIL_MIN) return panels << 8;
    return ((R << 8) / nt_n256) << 8;
}

// 4 x 16-bit (bf16 | fp16) payload <-> floats
template <bool F16> __device__ __forceinline__ void unpack4_16(uint2 u, float (&f)[4]) {
    if (F16) {
        union { uint2 u; half4_t h; } c; c.u = u;
        f[0] = (float)c.h[0]; f[1] = (float)c.h[1]; f[2] = (float)c.h[2]; f[3] = (float)c.h[3];
    } else {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    }
}
// v -> (hi, lo) planes: hi = T(v), lo = T(v - float(hi)); (v - hi) is exact in fp32
template <bool F16> __device__ __forceinline__ void split4_16(const float (&v)[4], uint2& hi, uint2& lo) {
    hi = pack4_16(v[0], v[1], v[2], v[3], F16 ? BG_F16 : BG_BF16);
    float h[4];
    unpack4_16<F16>(hi, h);
    lo = pack4_16(v[0] - h[0], v[1] - h[1], v[2] - h[2], v[3] - h[3], F16 ? BG_F16 : BG_BF16);
}
// butterfly sums inside aligned groups of 8 / 16 lanes (pure DPP, every lane ends with the group total; the
// association order is fixed, so the persistent and the generic GEMM produce bit-identical row statistics)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);     // row_half_mirror: lane i <- lane 7 - i of its 8-lane half row
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
    v = group8_sum(v);
    v += dpp_mov<0x140>(v);     // row_mirror: lane i <- lane 15 - i of its 16-lane row
    return v;
}
// Row statistics from the per-64-column partials a producer GEMM wrote.  The 16 (zero-padded) partials are summed
// in the butterfly order of group16_sum, written out explicitly, so a lane that owns a whole row (persistent kernel)
// and 16 lanes that share one (generic kernel) get bit-identical sums.
__device__ __forceinline__ float tree16(const float (&p)[16]) {
    const float q0 = (p[0] + p[1]) + (p[2] + p[3]), q1 = (p[4] + p[5]) + (p[6] + p[7]);
    const float q2 = (p[8] + p[9]) + (p[10] + p[11]), q3 = (p[12] + p[13]) + (p[14] + p[15]);
    return (q0 + q1) + (q2 + q3);
}
// (S, Q) = (sum, sum of squares) over K elements  ->  (rstd, -mean * rstd)
__device__ __forceinline__ float2 ln_fold_coeffs(float S, float Q, int K, float eps) {
    const float inv = 1.0f / (float)K;
    const float mean = S * inv;
    float var = Q * inv - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    return make_float2(rstd, -(mean * rstd));
}
// LayerNorm fold applied to one accumulator: rstd * acc + (-mean * rstd * colsum + bias'), as two explicit fmas (the
// build runs with -ffp-contract=off, so every kernel that calls this rounds identically)
__device__ __forceinline__ float ln_fold_apply(float acc, float rstd, float nmr, float colsum, float bias) {
    return __builtin_fmaf(acc, rstd, __builtin_fmaf(nmr, colsum, bias));
}
int gemm_f32(const GemmArgs& g, hipStream_t s);
int gemm_16bit(const GemmArgs& g, int ab_dtype, hipStream_t s);   // ab_dtype: BG_BF16 | BG_F16
int gemm(const GemmArgs& g, int ab_dtype, hipStream_t s);

int layernorm768(const float* x, const float* g, const float* b, void* y, int y_dtype, int M, float eps,
                 int silu, hipStream_t s, const int* m_dev = nullptr, double rows_hint = 0.0);
// same, input rows given as the split pair x = hi + lo (16-bit planes of dtype y_dtype)
int layernorm768_split(const void* hi, const void* lo, const float* g, const float* b, void* y, int y_dtype, int M,
                       float eps, hipStream_t s, const int* m_dev = nullptr, double rows_hint = 0.0);
// h = SiLU(LayerNorm(x W0^T + b0)) for k in {6, 12, 48}; w0p = W0 in MFMA operand order (embed.hip)
bool embed_ln_silu_supported(int k);
// vae.hip: GroupNorm(1 group) + activation (+ add) + cast of small samples in one pass (bit-identical to bg_groupnorm_stats + bg_im2col 1x1)
bool gn1_norm_act_supported(int P, int C);
int gn1_norm_act(const float* x, void* out, int out_dtype, int S, int P, int C, const float* gamma, const float* beta, float eps,
                 int act, const float* add, hipStream_t s);
// (m_dev / src_row: compacted batches -- *m_dev rows exist, row r reads x[src_row[r]])
int embed_ln_silu(const float* x, int lda, int rows, int k, const float* w0p, const float* b0, const float* g,
                  const float* b, void* out, int out_dtype, float eps, hipStream_t s, const int* m_dev = nullptr,
                  const int* src_row = nullptr, double rows_hint = 0.0);
// offsets != null: compacted batch -- sample b owns qkv / out rows offsets[b] .. offsets[b+1]-1 (<= N of them, all valid
// keys; key_pad is ignored); otherwise sample b owns rows b*N .. b*N+N-1 and key_pad marks the padded keys.
int attention(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype, hipStream_t s,
              const int* offsets = nullptr, double pairs_hint = 0.0, double rows_hint = 0.0);
// QKV (LayerNorm fold) + attention of short, equally long sequences (key_pad: optional [B, N] mask) in one launch (qkv_attn.hip)
bool qkv_attn_eligible(int B, int N, int dtype, const void* stats_in, const void* colsum, const void* bias);
bool qkv_attn_worthwhile(int B, int N);      // enough (sample group, head) tiles to fill the chip
int qkv_attention(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in, void* out,
                  const uint8_t* key_pad, int B, int N, int dtype, float ln_eps, hipStream_t s);

int qkv_attention_paired(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in, void* out,
                         void* qkv_dbg, const int* m_dev, const int* slot_desc, int slot_bound, int m_stats, int dtype, float ln_eps,
                         hipStream_t s, double rows_hint, double pairs_hint);
// slot-packed compaction (rep == 1, n_mask <= 64): 64-row slots of one or two whole samples; offsets [B + 1] (offsets[B] = rows),
// src_row [64 B], slot_desc [2 B] (lengths of the samples of every slot), slot_a [B], counts [B] (scratch)
int compact_rows_paired(const uint8_t* mask, int B, int n_mask, int* offsets, int* src_row, int* slot_desc, int* slot_a, int* counts,
                        hipStream_t s, int* rule = nullptr);
// valid-token compaction of a padded batch (csrc/compact.hip): mask [B, n_mask] uint8 (1 = padded), each mask entry
// covering `rep` consecutive tokens (EdgePosNet: rep = E).  offsets [B+1] (offsets[B] = *m_dev = number of valid tokens),
// src_row [B * n_mask * rep]: padded-layout index of every compact row, in order.
int compact_rows(const uint8_t* mask, int B, int n_mask, int rep, int* offsets, int* src_row, hipStream_t s, int* rule = nullptr);
int sincos_embed(const int64_t* t, int n, float* out, hipStream_t s);
// c[b,:] = temb[(nt==1?0:b),:] + (class_embed ? class_embed[label[b],:] : 0)
int cond_vector(const float* temb, int nt, const float* class_embed, const int64_t* label, float* c, int B,
                hipStream_t s);
// same with temb[b] = table[timesteps[(nt == 1 ? 0 : b)]] looked up in a precomputed [table_rows, 768] time-embedding table
// (a timestep outside the table gives NaN rows: fails loudly in the result)
int cond_vector_table(const float* table, int table_rows, const int64_t* timesteps, int nt, const float* class_embed,
                      const int64_t* label, float* c, int B, hipStream_t s);
// eps = W3 . SiLU(LayerNorm(t0)) + b3 in one launch (out_tail.hip): t0 [M, 768] 16-bit, out fp32 [*, n_out]
bool ln_silu_out_supported(int n_out, int n_out_pad);
int ln_silu_out(const void* t0, const float* gam, const float* bet, const void* w3, const float* b3, float* out, int n_out, int n_out_pad,
                int M, int dtype, float eps, hipStream_t s, const int* m_dev = nullptr, const int* row_map = nullptr, double rows_hint = 0.0);
// FFN1 (LayerNorm fold, ReLU) + FFN2 (split residual in place + row statistics) of an encoder layer in one launch (ffn_fused.hip)
struct FfnArgs {
    void* xh; void* xl;                 // residual planes [M, 768], updated in place
    float* stats;                       // [12][m_stride][2]: LayerNorm-2 partials in, the new rows' partials out
    const void* w1f; const float* b1; const float* colsum1;     // W1' in fragment order, b1' = b1 + W1 beta, column sums of W1'
    const void* w2f; const float* b2;
    int M, m_stride;
    const int* m_dev;
    float ln_eps;
    long long* stamps = nullptr;        // measurement aid (bg_tune keys 17 / 18 = a device address): s_memtime of workgroup 0's phases
};
bool ffn_fused_eligible(const FfnArgs& g, int dtype);
int ffn_fused(const FfnArgs& g, int dtype, hipStream_t s, double rows_hint);
// out = f32 -> bf16 cast (n elements, n % 4 == 0)
int cast_f32_bf16(const float* in, void* out, size_t n, hipStream_t s);

}  // namespace bg
