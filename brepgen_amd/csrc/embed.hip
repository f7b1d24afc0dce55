// First half of every input-embedding MLP of the denoisers, fused:  h = SiLU(LayerNorm(x W0^T + b0))
// (sub-keys .0 .1 and the SiLU of p_embed / z_embed / surfp / surfz / edgep / edgez / vertp, network.py:1080-1085,
// 1142-1153, 1216-1234, 1302-1332; x = raw bboxes / latents with k = 6, 12 or 48 columns).
//
// Unfused this is an exact-fp32 GEMM that writes [rows,768] fp32 (94 MB at BASELINE configs[1]) and a LayerNorm
// kernel that reads it back.  Here a 512-thread block owns 32 rows: each of its 8 waves computes 32 x 96 outputs on the
// f32-input matrix core (v_mfma_f32_32x32x2_f32: an fma chain over k in ascending order, i.e. the numerics of the
// fp32 GEMM it replaces -- the raw inputs are never rounded to 16 bits), keeps them in registers, the row
// statistics are reduced two-pass (mean, then centred variance, like torch.nn.LayerNorm) across the 32 column lanes
// by DPP / swizzle and across the 8 waves through 2 KiB of LDS, and only the 16-bit (or fp32) result is written:
// HBM-bound on its output, 2 bytes per element.
//   W0 is repacked on the host into MFMA operand order, w0p[ct][kk][lane] = W0[ct*32 + (lane & 31)][2*kk + (lane >> 5)],
//   so every B-operand load is one coalesced 256-byte line (bg_mlp_weights.w0_mfma).
#include "bg_common.h"

namespace bg {

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 32 lanes that share (lane >> 5); every lane of the half ends with the total
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_f<0xB1>(v);                                          // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);                                          // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);                                         // row_half_mirror
    v += dpp_f<0x140>(v);                                         // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // lane ^ 16
    return v;
}

template <int OUT, int K2>      // OUT: BG_F32 | BG_BF16 | BG_F16;  K2 = k / 2 in {3, 6, 24}
__global__ __launch_bounds__(512) void embed_ln_silu_kernel(const float* __restrict__ x, int lda, int rows,
                                                            const float* __restrict__ w0p, const float* __restrict__ b0,
                                                            const float* __restrict__ gam, const float* __restrict__ bet,
                                                            void* __restrict__ out, float eps,
                                                            const int* __restrict__ m_dev, const int* __restrict__ src_row) {
    constexpr int NW = 8, TPW = 3;                                // waves per block, 32-column tiles per wave
    __shared__ float red[2][NW][32];
    __shared__ __attribute__((aligned(16))) unsigned tile[OUT == BG_F32 ? 1 : 32 * 384];      // 32 rows x 768 x 2 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, c = lane & 31;
    const int row0 = blockIdx.x * 32;
    if (m_dev) rows = *m_dev;                                     // compacted batch: the row count lives on the device
    if (row0 >= rows) return;                                     // uniform per workgroup, before any barrier

    // A operand: lane supplies x[row0 + c][2*kk + h] (compacted batch: row r of the output reads x[src_row[r]])
    int arow = row0 + c;
    arow = arow < rows ? arow : rows - 1;
    if (src_row) arow = src_row[arow];
    const float* xr = x + (size_t)arow * lda + h;
    float a[K2];
#pragma unroll
    for (int kk = 0; kk < K2; ++kk) a[kk] = xr[2 * kk];

    // per-column vectors first: their L2 round trip overlaps the MFMA chain instead of following it
    float bias[TPW], gcol[TPW], bcol[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int col = (wave * TPW + t) * 32 + c;
        bias[t] = b0[col];
        gcol[t] = gam[col];
        bcol[t] = bet[col];
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* wp = w0p + (size_t)(wave * TPW + t) * K2 * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < K2; ++kk)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], wp[kk * 64], acc[t], 0, 0, 0);
    }

    // C layout: column = ct*32 + c, row = (r & 3) + 8 * (r >> 2) + 4 * h
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += bias[t];

    // ---- LayerNorm statistics, two-pass in registers ----
    float mean[16], rstd[16];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float part[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float d = pass == 0 ? acc[t][r] : (acc[t][r] - mean[r]);
                s += pass == 0 ? d : d * d;
            }
            part[r] = half_wave_sum(s);
        }
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[pass][wave][(r & 3) + 8 * (r >> 2) + 4 * h] = part[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            float tot = red[pass][0][row];
#pragma unroll
            for (int w = 1; w < NW; ++w) tot += red[pass][w][row];
            if (pass == 0) mean[r] = tot * (1.0f / 768.0f);
            else rstd[r] = 1.0f / sqrtf(tot * (1.0f / 768.0f) + eps);
        }
    }

    // ---- normalise, SiLU, store ----
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float y = (acc[t][r] - mean[r]) * rstd[r] * gcol[t] + bcol[t];
            acc[t][r] = OUT == BG_F32 ? silu_f(y) : silu_rcp(y);   // (fp32 output = the exact-fp32 mode: IEEE division; 16-bit: rounded next)
        }

    if (OUT == BG_F32) {
        float* o = reinterpret_cast<float*>(out);
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < rows) o[(size_t)row * 768 + (wave * TPW + t) * 32 + c] = acc[t][r];   // 128-byte runs per half wave
            }
        return;
    }
    // 16-bit: lane pairs swap one value (DPP) so each lane owns two neighbouring columns of one row, the block's
    // 32 x 768 slab goes through LDS and leaves as 16-byte stores of whole 1536-byte rows
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) {
            const float va = acc[t][2 * rp], vb = acc[t][2 * rp + 1];
            const float send = (lane & 1) ? va : vb;
            const float recv = dpp_f<0xB1>(send);
            const float lo = (lane & 1) ? recv : va, hi = (lane & 1) ? vb : recv;
            const int r = 2 * rp + (lane & 1);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            union { __bf16 b[2]; _Float16 f[2]; unsigned u; } pk;
            if (OUT == BG_F16) { pk.f[0] = (_Float16)lo; pk.f[1] = (_Float16)hi; }
            else { pk.b[0] = (__bf16)lo; pk.b[1] = (__bf16)hi; }
            tile[row * 384 + (wave * TPW + t) * 16 + (c >> 1)] = pk.u;
        }
    __syncthreads();
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int chunk = it * 512 + threadIdx.x;                 // 3072 16-byte chunks: 96 per row
        const int row = chunk / 96, cc = chunk % 96;
        const uint4 v = *reinterpret_cast<const uint4*>(&tile[row * 384 + cc * 4]);
        if (row0 + row < rows) *reinterpret_cast<uint4*>(o + (size_t)(row0 + row) * 768 + cc * 8) = v;
    }
}

template <int OUT>
static void launch_embed(int k2, dim3 grid, hipStream_t s, const float* x, int lda, int rows, const float* w0p,
                         const float* b0, const float* g, const float* b, void* out, float eps, const int* m_dev,
                         const int* src_row) {
    if (k2 == 3) hipLaunchKernelGGL((embed_ln_silu_kernel<OUT, 3>), grid, dim3(512), 0, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
    else if (k2 == 6) hipLaunchKernelGGL((embed_ln_silu_kernel<OUT, 6>), grid, dim3(512), 0, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
    else hipLaunchKernelGGL((embed_ln_silu_kernel<OUT, 24>), grid, dim3(512), 0, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
}

bool embed_ln_silu_supported(int k) { return k == 6 || k == 12 || k == 48; }

int embed_ln_silu(const float* x, int lda, int rows, int k, const float* w0p, const float* b0, const float* g,
                  const float* b, void* out, int out_dtype, float eps, hipStream_t s, const int* m_dev, const int* src_row, double rows_hint) {
    if (rows <= 0) return 0;
    if (!embed_ln_silu_supported(k) || lda < k) {
        set_error("embed_ln_silu: k must be 6, 12 or 48 with lda >= k (k=%d lda=%d)", k, lda);
        return BG_E_SHAPE;
    }
    const dim3 grid((rows + 31) / 32);
    const double prows = rows_hint > 0 ? rows_hint : (double)rows;
    ProfScope prof(PK_EMBED, 2.0 * prows * 768.0 * k, prows * (4.0 * k + 768.0 * (out_dtype == BG_F32 ? 4.0 : 2.0)), s);
    if (out_dtype == BG_BF16) launch_embed<BG_BF16>(k / 2, grid, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
    else if (out_dtype == BG_F16) launch_embed<BG_F16>(k / 2, grid, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
    else if (out_dtype == BG_F32) launch_embed<BG_F32>(k / 2, grid, s, x, lda, rows, w0p, b0, g, b, out, eps, m_dev, src_row);
    else {
        set_error("embed_ln_silu: unsupported output dtype %d", out_dtype);
        return BG_E_DTYPE;
    }
    return launch_status("embed_ln_silu");
}

}  // namespace bg

extern "C" int bg_embed_ln_silu_fwd(const float* x, int lda, int rows, int k, const float* w0_mfma, const float* b0,
                                    const float* ln_g, const float* ln_b, void* out, int out_dtype, float eps,
                                    bg_stream_t stream) {
    BG_REQUIRE(x && w0_mfma && b0 && ln_g && ln_b && out, BG_E_ARG, "bg_embed_ln_silu_fwd: null pointer");
    return bg::embed_ln_silu(x, lda, rows, k, w0_mfma, b0, ln_g, ln_b, out, out_dtype, eps, (hipStream_t)stream);
}
