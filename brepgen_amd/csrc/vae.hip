// Elementwise / gather kernels of the surface / edge VAEs (network.py:690-1040 + the diffusers==0.27 blocks).
//
// Layout: channels-last activations, fp32: 2-D [F, H, W, C], 1-D [G, L, C] (== 2-D with H = 1).  A convolution is
//   16-bit modes, 3x3 / k5, stride 1:  norm_act_kernel (GroupNorm + activation + cast, 1x1 window) -> implicit GEMM
//                                      (gemm_16bit.hip, CONV instantiation: the GEMM's loader walks the window)
//   everything else:                   im2col_kernel over the window -> MFMA GEMM (gemm_16bit.hip / gemm_f32.hip)
// with bias + residual fused in the GEMM epilogue either way:
//   * the GroupNorm + SiLU / GELU that precedes each conv in ResnetBlock2D / ResConvBlock is applied INSIDE the
//     im2col gather (per-(sample, group) mean / rstd from gn_stats_kernel), so the normalised tensor never exists;
//   * the nearest-neighbour x2 up-sampling of Upsample2D is folded into the gather as well (source index >> 1);
//   * zero padding is applied after norm + activation, as the reference pads the activated tensor.
// Everything here is HBM-bound (judged in GB/s); the FLOPs are in the GEMMs.
#include "bg_common.h"
#include <math.h>

namespace bg {

// ---- GroupNorm statistics: one wave per (sample, group), two passes (mean, centred variance) ----------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ stats, int S, int P,
                                                       int C, int G, float eps) {
    const int lane = threadIdx.x & 63;
    const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= S * G) return;
    const int s = id / G, g = id % G;
    const int cpg = C / G, q4 = cpg >> 2;                 // float4 per position
    const int n4 = P * q4;
    const float* base = x + (size_t)s * P * C + g * cpg;
    float sum = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C + c4 * 4);
        sum += (v.x + v.y) + (v.z + v.w);
    }
    const float inv_n = 1.0f / (float)(P * cpg);
    const float mean = wave_sum(sum) * inv_n;
    float var = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const int p = i / q4, c4 = i - p * q4;
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * C + c4 * 4);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        var += (a * a + b * b) + (c * c + d * d);
    }
    var = wave_sum(var) * inv_n;
    if (lane == 0) {
        stats[(size_t)id * 2 + 0] = mean;
        stats[(size_t)id * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
}

// ---- GroupNorm statistics, many groups per row (G > 1; the 2-D VAE: 32 groups of 4..16 channels): one workgroup per
// sample streams its rows whole -- every wave reads 1 KiB of consecutive channels -- instead of one wave per group
// picking 16-byte pieces out of every row.  256 % (C/4) == 0, so a thread always meets the same 4 channels, hence one
// group.  ONE pass: per-thread sums shifted by the thread's first value (no cancellation against the mean), combined
// across the threads of a group with the pairwise formula M2 = sum M2_t + sum n_t (mean_t - mean)^2.
__global__ __launch_bounds__(256) void gn_stats_rows_kernel(const float* __restrict__ x, float* __restrict__ stats, int P, int C,
                                                            int G, float eps) {
    __shared__ float sh_n[256], sh_mean[256], sh_m2[256];
    const int tid = threadIdx.x, s = blockIdx.x;
    const int c4n = C >> 2, n4 = P * c4n;
    const float4* __restrict__ base = reinterpret_cast<const float4*>(x + (size_t)s * P * C);
    float n = 0.f, s1 = 0.f, s2 = 0.f, pivot = 0.f;
    if (tid < n4) pivot = base[tid].x;
    int i = tid;
    for (; i + 768 < n4; i += 1024) {                      // 4 independent 16-byte loads in flight per thread
        const float4 v0 = base[i], v1 = base[i + 256], v2 = base[i + 512], v3 = base[i + 768];
        const float4 vs[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = vs[k].x - pivot, b = vs[k].y - pivot, c = vs[k].z - pivot, d = vs[k].w - pivot;
            s1 += (a + b) + (c + d);
            s2 += (a * a + b * b) + (c * c + d * d);
        }
        n += 16.f;
    }
    for (; i < n4; i += 256) {
        const float4 v = base[i];
        const float a = v.x - pivot, b = v.y - pivot, c = v.z - pivot, d = v.w - pivot;
        s1 += (a + b) + (c + d);
        s2 += (a * a + b * b) + (c * c + d * d);
        n += 4.f;
    }
    const float mean_t = n > 0.f ? pivot + s1 / n : 0.f;
    sh_n[tid] = n;
    sh_mean[tid] = mean_t;
    sh_m2[tid] = n > 0.f ? s2 - s1 * s1 / n : 0.f;
    __syncthreads();
    if (tid < G) {
        // the threads of group g: c4 = t % c4n in [g q4, (g+1) q4), q4 = (C/G)/4
        const int q4 = (C / G) >> 2, reps = 256 / c4n;
        float nt = 0.f, sm = 0.f;
        for (int r = 0; r < reps; ++r)
            for (int j = 0; j < q4; ++j) {
                const int t = r * c4n + tid * q4 + j;
                nt += sh_n[t];
                sm += sh_n[t] * sh_mean[t];
            }
        const float mean = sm / nt;
        float m2 = 0.f;
        for (int r = 0; r < reps; ++r)
            for (int j = 0; j < q4; ++j) {
                const int t = r * c4n + tid * q4 + j;
                const float d = sh_mean[t] - mean;
                m2 += sh_m2[t] + sh_n[t] * d * d;
            }
        stats[((size_t)s * G + tid) * 2 + 0] = mean;
        stats[((size_t)s * G + tid) * 2 + 1] = 1.0f / sqrtf(m2 / nt + eps);
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return silu_f(v);
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));    // nn.GELU() (erf form)
    return v;
}

struct Im2colArgs {
    const float* x; void* out; int out_dtype;
    int S, Hin, Win, C, kh, kw, up;
    int stride, pad_y, pad_x, Ho, Wo;   // output grid and the padding BEFORE the first row / column (after: zero fill)
    const float* stats; const float* gamma; const float* beta; int G; int act;
    const float* add;        // optional residual, added after norm + activation (1x1 / fp32-out use only)
};

// one thread = 4 consecutive channels of one (row, tap).  IDX = unsigned whenever the element count allows it: the index
// arithmetic is seven divisions per 16 bytes moved, and 64-bit ones cost more issue slots than the HBM stream leaves.
template <typename IDX>
__global__ __launch_bounds__(256) void im2col_kernel(Im2colArgs a) {
    const int H = a.Hin << a.up, W = a.Win << a.up;       // logical input grid
    const IDX c4n = (IDX)(a.C >> 2), taps = (IDX)(a.kh * a.kw), Wo = (IDX)a.Wo, Ho = (IDX)a.Ho;
    const IDX total = (IDX)a.S * Ho * Wo * taps * c4n;
    const int cpg = a.stats ? a.C / a.G : 1;
    for (IDX i = blockIdx.x * (IDX)blockDim.x + threadIdx.x; i < total; i += (IDX)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        IDX r = i / c4n;
        const int tap = (int)(r % taps);
        r /= taps;                                            // row = (s, oy, ox)
        const int ox = (int)(r % Wo);
        const IDX r2 = r / Wo;
        const int oy = (int)(r2 % Ho), s = (int)(r2 / Ho);
        const int iy = oy * a.stride + tap / a.kw - a.pad_y, ix = ox * a.stride + tap % a.kw - a.pad_x;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const int c = c4 * 4;
            v = *reinterpret_cast<const float4*>(a.x + (((size_t)s * a.Hin + (iy >> a.up)) * a.Win + (ix >> a.up)) * a.C + c);
            if (a.stats) {
                const int g = c / cpg;                        // cpg % 4 == 0: the 4 channels share a group
                const float mean = a.stats[((size_t)s * a.G + g) * 2], rstd = a.stats[((size_t)s * a.G + g) * 2 + 1];
                const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c);
                const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
                v.x = (v.x - mean) * rstd * ga.x + be.x;
                v.y = (v.y - mean) * rstd * ga.y + be.y;
                v.z = (v.z - mean) * rstd * ga.z + be.z;
                v.w = (v.w - mean) * rstd * ga.w + be.w;
            }
            v.x = act_apply(v.x, a.act); v.y = act_apply(v.y, a.act);
            v.z = act_apply(v.z, a.act); v.w = act_apply(v.w, a.act);
        }
        const size_t o = (size_t)i * 4;                       // == ((row * taps + tap) * C + c4 * 4)
        if (a.add) {
            const float4 ad = *reinterpret_cast<const float4*>(a.add + o);
            v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
        }
        if (a.out_dtype != BG_F32) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.out) + o) = pack4_16(v.x, v.y, v.z, v.w, a.out_dtype);
        else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + o) = v;
    }
}

// The 1x1 window on the same grid (no gather at all): out = act(GroupNorm(x)) (+ add), cast.  This is the pass in front
// of every implicit-GEMM convolution, the tail of ResConvBlock and the pre-norm of the mid-block attention -- the
// general kernel above spends more issue slots on its 64-bit index arithmetic than the HBM stream leaves room for.
// Flat 32-bit index over the float4 elements; same per-element arithmetic as im2col_kernel (bit-identical results).
__global__ __launch_bounds__(256) void norm_act_kernel(Im2colArgs a, unsigned total4, unsigned n4s, unsigned c4n) {
    const unsigned cpg4 = a.stats ? (unsigned)(a.C / a.G) >> 2 : 1u;
    const bool pow2 = (c4n & (c4n - 1)) == 0;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
    const unsigned i0 = blockIdx.x * 1024u + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = i0 + k * 256u;
        if (i < total4) v[k] = x4[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = i0 + k * 256u;
        if (i >= total4) continue;
        float4 w = v[k];
        const unsigned c4 = pow2 ? (i & (c4n - 1)) : (i % c4n);
        if (a.stats) {
            const unsigned smp = i / n4s;
            const float2 st = *reinterpret_cast<const float2*>(a.stats + ((size_t)smp * a.G + c4 / cpg4) * 2);
            const float mean = st.x, rstd = st.y;
            const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c4 * 4);
            const float4 be = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
            w.x = (w.x - mean) * rstd * ga.x + be.x;
            w.y = (w.y - mean) * rstd * ga.y + be.y;
            w.z = (w.z - mean) * rstd * ga.z + be.z;
            w.w = (w.w - mean) * rstd * ga.w + be.w;
        }
        w.x = act_apply(w.x, a.act); w.y = act_apply(w.y, a.act);
        w.z = act_apply(w.z, a.act); w.w = act_apply(w.w, a.act);
        if (a.add) {
            const float4 ad = reinterpret_cast<const float4*>(a.add)[i];
            w.x += ad.x; w.y += ad.y; w.z += ad.z; w.w += ad.w;
        }
        if (a.out_dtype != BG_F32) reinterpret_cast<uint2*>(a.out)[i] = pack4_16(w.x, w.y, w.z, w.w, a.out_dtype);
        else reinterpret_cast<float4*>(a.out)[i] = w;
    }
}

// ---- GroupNorm(1 group) + activation (+ add) + cast of SMALL samples in ONE pass: the 1-D VAE's blocks normalise whole samples of
// 2048 / 4096 values (diffusers' ResConvBlock / SelfAttention1d: GroupNorm(1, C) over [L, C], L = 4 .. 32).  One wave per sample
// holds it in registers (NV float4 per lane), so the tensor is read from HBM once instead of three times (two statistics passes of
// gn_stats_kernel + norm_act_kernel).  The arithmetic -- lane-strided partial sums, wave butterfly, two-pass variance, then
// norm_act_kernel's per-element expression -- is that of the two kernels, operation for operation: results are bit-identical to
// bg_groupnorm_stats + bg_im2col(1x1) (tests: program == step by step, where the step-by-step driver calls the two).
template <int NV, int C4N>           // float4 per lane (n4 = 64 NV per sample); float4 per position (C / 4)
__global__ __launch_bounds__(256) void gn1_norm_act_kernel(const float* __restrict__ x, void* __restrict__ out, int out_dtype, int S,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           int act, const float* __restrict__ add) {
    const int lane = threadIdx.x & 63;
    const int id = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= S) return;
    constexpr int N4 = NV * 64;
    const size_t e0 = (size_t)id * N4 + lane;
    const float4* __restrict__ base = reinterpret_cast<const float4*>(x) + e0;
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = base[k * 64];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float inv_n = 1.0f / (float)(N4 * 4);
    const float mean = wave_sum(sum) * inv_n;
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
        var += (a * a + b * b) + (c * c + d * d);
    }
    var = wave_sum(var) * inv_n;
    const float rstd = 1.0f / sqrtf(var + eps);
    // channels of element lane + 64 k: c4 = (lane + 64 k) % C4N -- one (C4N <= 64) or two (C4N = 128) distinct sets per lane
    constexpr int NSET = C4N > 64 ? C4N / 64 : 1;
    float4 ga[NSET], be[NSET];
#pragma unroll
    for (int q = 0; q < NSET; ++q) {
        const int c4 = (lane + 64 * q) & (C4N - 1);
        ga[q] = reinterpret_cast<const float4*>(gamma)[c4];
        be[q] = reinterpret_cast<const float4*>(beta)[c4];
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float4 g4 = ga[k % NSET], b4 = be[k % NSET];
        float4 w = v[k];
        w.x = (w.x - mean) * rstd * g4.x + b4.x;
        w.y = (w.y - mean) * rstd * g4.y + b4.y;
        w.z = (w.z - mean) * rstd * g4.z + b4.z;
        w.w = (w.w - mean) * rstd * g4.w + b4.w;
        w.x = act_apply(w.x, act); w.y = act_apply(w.y, act);
        w.z = act_apply(w.z, act); w.w = act_apply(w.w, act);
        if (add) {
            const float4 ad = reinterpret_cast<const float4*>(add)[e0 + k * 64];
            w.x += ad.x; w.y += ad.y; w.z += ad.z; w.w += ad.w;
        }
        if (out_dtype != BG_F32) reinterpret_cast<uint2*>(out)[e0 + k * 64] = pack4_16(w.x, w.y, w.z, w.w, out_dtype);
        else reinterpret_cast<float4*>(out)[e0 + k * 64] = w;
    }
}

bool gn1_norm_act_supported(int P, int C) {
    const int n4 = P * (C / 4);
    return C % 4 == 0 && (n4 == 512 || n4 == 1024) && (C == 128 || C == 256 || C == 512);
}

// GroupNorm(1, C) + activation (+ add) + cast of [S, P, C] fp32 samples; the caller has checked gn1_norm_act_supported(P, C)
int gn1_norm_act(const float* x, void* out, int out_dtype, int S, int P, int C, const float* gamma, const float* beta, float eps,
                 int act, const float* add, hipStream_t s) {
    const int nv = P * (C / 4) / 64, c4n = C / 4;
    ProfScope prof(PK_MISC, 0.0, (double)S * P * C * (4.0 + (out_dtype == BG_F32 ? 4.0 : 2.0) + (add ? 4.0 : 0.0)), s);
    const dim3 grid((S + 3) / 4), block(256);
#define BG_GN1(NV_, C4N_) hipLaunchKernelGGL((gn1_norm_act_kernel<NV_, C4N_>), grid, block, 0, s, x, out, out_dtype, S, gamma, beta, eps, act, add)
    if (nv == 8 && c4n == 32) BG_GN1(8, 32);
    else if (nv == 8 && c4n == 64) BG_GN1(8, 64);
    else if (nv == 8 && c4n == 128) BG_GN1(8, 128);
    else if (nv == 16 && c4n == 32) BG_GN1(16, 32);
    else if (nv == 16 && c4n == 64) BG_GN1(16, 64);
    else if (nv == 16 && c4n == 128) BG_GN1(16, 128);
    else {
        set_error("gn1_norm_act: unsupported shape P=%d C=%d", P, C);
        return BG_E_SHAPE;
    }
#undef BG_GN1
    return launch_status("gn1_norm_act");
}

// any C (the 3-channel latent inputs of post_quant_conv / conv_in), no norm, fp32 out
__global__ __launch_bounds__(256) void im2col_scalar_kernel(Im2colArgs a) {
    const int H = a.Hin << a.up, W = a.Win << a.up, taps = a.kh * a.kw;
    const size_t total = (size_t)a.S * a.Ho * a.Wo * taps * a.C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % a.C);
        size_t r = i / a.C;
        const int tap = (int)(r % taps);
        r /= taps;
        const int ox = (int)(r % a.Wo);
        const size_t r2 = r / a.Wo;
        const int oy = (int)(r2 % a.Ho), s = (int)(r2 / a.Ho);
        const int iy = oy * a.stride + tap / a.kw - a.pad_y, ix = ox * a.stride + tap % a.kw - a.pad_x;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
            v = act_apply(a.x[(((size_t)s * a.Hin + (iy >> a.up)) * a.Win + (ix >> a.up)) * a.C + c], a.act);
        if (a.out_dtype == BG_BF16) reinterpret_cast<__bf16*>(a.out)[i] = (__bf16)v;
        else if (a.out_dtype == BG_F16) reinterpret_cast<_Float16*>(a.out)[i] = (_Float16)v;
        else reinterpret_cast<float*>(a.out)[i] = v;
    }
}

// ---- Upsample1d("cubic"): reflect-pad 2, stride-2 transposed conv with the 8-tap kernel x2, padding 7 ----------
// written as the equivalent 4-tap depthwise gather: y[o] = sum_i hp[i] * w[o + 7 - 2i], hp[i] = x[reflect(i - 2)]
__constant__ float kCubic2[8] = {-0.0234375f, -0.0703125f, 0.2265625f, 0.8671875f, 0.8671875f, 0.2265625f, -0.0703125f, -0.0234375f};

__global__ __launch_bounds__(256) void upsample1d_cubic_kernel(const float* __restrict__ x, float* __restrict__ y, int S, int L,
                                                               int C) {
    const size_t total = (size_t)S * 2 * L * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int o = (int)(r % (2 * L)), s = (int)(r / (2 * L));
        float acc = 0.f;
        const int i0 = (o + 1) >> 1, i1 = (o + 7) >> 1;
        for (int ii = i0; ii <= i1; ++ii) {
            const int kk = o + 7 - 2 * ii;
            if (kk < 0 || kk > 7 || ii > L + 3) continue;
            int j = ii - 2;
            j = j < 0 ? -j : (j >= L ? 2 * (L - 1) - j : j);
            acc = fmaf(x[((size_t)s * L + j) * C + c], kCubic2[kk], acc);
        }
        y[i] = acc;
    }
}

// the same for 4 consecutive channels per thread, 32-bit index arithmetic (C % 4 == 0 and < 2^32 elements: every call of
// the edge decoder); per channel the same fmaf chain in the same order as the scalar kernel
__global__ __launch_bounds__(256) void upsample1d_cubic4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned total4,
                                                                unsigned L, unsigned c4n) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const unsigned c4 = i % c4n, r = i / c4n;
    const unsigned o = r % (2 * L), s = r / (2 * L);
    const float4* __restrict__ xs = reinterpret_cast<const float4*>(x) + (size_t)s * L * c4n + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int i0 = (int)(o + 1) >> 1, i1 = (int)(o + 7) >> 1;
    for (int ii = i0; ii <= i1; ++ii) {
        const int kk = (int)o + 7 - 2 * ii;
        if (kk < 0 || kk > 7 || ii > (int)L + 3) continue;
        int j = ii - 2;
        j = j < 0 ? -j : (j >= (int)L ? 2 * ((int)L - 1) - j : j);
        const float4 v = xs[(size_t)j * c4n];
        const float w = kCubic2[kk];
        acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y);
        acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
    }
    reinterpret_cast<float4*>(y)[i] = acc;
}

// ---- Downsample1d("cubic") of the edge encoder (diffusers DownBlock1D): reflect-pad 3, depthwise stride-2 conv with
// the 8-tap cubic kernel: y[o] = sum_k hp[2o + k] * w[k], hp[i] = x[reflect(i - 3)] ----
__global__ __launch_bounds__(256) void downsample1d_cubic_kernel(const float* __restrict__ x, float* __restrict__ y, int S,
                                                                 int L, int C) {
    const int Lo = L >> 1;
    const size_t total = (size_t)S * Lo * C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t r = i / C;
        const int o = (int)(r % Lo), s = (int)(r / Lo);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int j = 2 * o + k - 3;
            j = j < 0 ? -j : (j >= L ? 2 * (L - 1) - j : j);
            acc = fmaf(x[((size_t)s * L + j) * C + c], 0.5f * kCubic2[k], acc);
        }
        y[i] = acc;
    }
}

// ---- tiny self-attention of the VAE mid blocks: T tokens (16 for the 4x4 surface latent, 4 for the edge latent),
// nh heads; one workgroup per sample; fp32 math.  qkv: [S*T, ld] with q | k | v at column offsets 0, C, 2C. ----
__global__ __launch_bounds__(256) void small_attn_kernel(const float* __restrict__ qkv, int ld, void* __restrict__ out,
                                                         int out_dtype, int T, int C, int nh, float scale) {
    extern __shared__ float sc[];                           // [nh][T][T]
    const int s = blockIdx.x, d = C / nh, n_sc = nh * T * T;
    const float* base = qkv + (size_t)s * T * ld;
    for (int e = threadIdx.x; e < n_sc; e += blockDim.x) {
        const int j = e % T, i = (e / T) % T, hh = e / (T * T);
        const float* q = base + (size_t)i * ld + hh * d;
        const float* k = base + (size_t)j * ld + C + hh * d;
        float acc = 0.f;
        for (int t = 0; t < d; ++t) acc = fmaf(q[t], k[t], acc);
        sc[e] = acc * scale;
    }
    __syncthreads();
    for (int row = threadIdx.x; row < nh * T; row += blockDim.x) {
        float* p = sc + (size_t)row * T;
        float m = -INFINITY;
        for (int j = 0; j < T; ++j) m = fmaxf(m, p[j]);
        float l = 0.f;
        for (int j = 0; j < T; ++j) { p[j] = expf(p[j] - m); l += p[j]; }
        const float inv = 1.0f / l;
        for (int j = 0; j < T; ++j) p[j] *= inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * C; e += blockDim.x) {
        const int col = e % C, i = e / C, hh = col / d;
        const float* p = sc + ((size_t)hh * T + i) * T;
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(p[j], base[(size_t)j * ld + 2 * C + col], acc);
        const size_t o = ((size_t)s * T + i) * C + col;
        if (out_dtype == BG_BF16) reinterpret_cast<__bf16*>(out)[o] = (__bf16)acc;
        else if (out_dtype == BG_F16) reinterpret_cast<_Float16*>(out)[o] = (_Float16)acc;
        else reinterpret_cast<float*>(out)[o] = acc;
    }
}

// the same with the sample's q|k|v rows staged in LDS by coalesced 16-byte loads first (the edge VAE: 4 tokens x 1536
// floats = 24 KiB per sample; the strided scalar reads of the kernel above ran at a quarter of the HBM rate).  Same
// fmaf chains in the same order: bit-identical results.
// LDS image: column c of a row lives at c + 4 * (c / d) (4 floats of padding per head-sized chunk), rows `wp` floats apart with
// wp % 64 == 16: in the score phase the 64 lanes of a wave read q (and k) of 4 heads x 4 rows at the SAME offset t -- unpadded
// (row stride 1536, head stride 32) those 16 addresses fall on two of the 64 banks, an 8-way conflict on every one of the 2 x 32
// reads of a dot product, and the launch ran at 2.5 TB/s bound by the LDS pipe; padded they fall on 16 different banks.
__device__ __forceinline__ int sa_col(int c, int d) { return c + 4 * (c / d); }
__host__ __device__ inline int sa_row_stride(int C, int d) {
    int wp = 3 * C + 4 * (3 * C / d);
    while ((wp & 63) != 16) wp += 4;
    return wp;
}

__global__ __launch_bounds__(256) void small_attn_lds_kernel(const float* __restrict__ qkv, int ld, void* __restrict__ out,
                                                             int out_dtype, int T, int C, int nh, float scale) {
    extern __shared__ float sm[];                           // [T][wp] padded rows | [nh][T][T] scores
    const int s = blockIdx.x, d = C / nh, n_sc = nh * T * T, w4 = (3 * C) >> 2;
    const int wp = sa_row_stride(C, d);
    float* sc = sm + (size_t)T * wp;
    const float* base = qkv + (size_t)s * T * ld;
    for (int e = threadIdx.x; e < T * w4; e += blockDim.x) {
        const int row = e / w4, c4 = e - row * w4;
        *reinterpret_cast<float4*>(sm + row * wp + sa_col(c4 * 4, d)) = *reinterpret_cast<const float4*>(base + (size_t)row * ld + c4 * 4);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n_sc; e += blockDim.x) {
        const int j = e % T, i = (e / T) % T, hh = e / (T * T);
        const float* q = sm + i * wp + sa_col(hh * d, d);
        const float* k = sm + j * wp + sa_col(C + hh * d, d);
        float acc = 0.f;
        for (int t = 0; t < d; ++t) acc = fmaf(q[t], k[t], acc);
        sc[e] = acc * scale;
    }
    __syncthreads();
    for (int row = threadIdx.x; row < nh * T; row += blockDim.x) {
        float* p = sc + (size_t)row * T;
        float m = -INFINITY;
        for (int j = 0; j < T; ++j) m = fmaxf(m, p[j]);
        float l = 0.f;
        for (int j = 0; j < T; ++j) { p[j] = expf(p[j] - m); l += p[j]; }
        const float inv = 1.0f / l;
        for (int j = 0; j < T; ++j) p[j] *= inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T * C; e += blockDim.x) {
        const int col = e % C, i = e / C, hh = col / d;
        const float* p = sc + ((size_t)hh * T + i) * T;
        const int vc = sa_col(2 * C + col, d);
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc = fmaf(p[j], sm[j * wp + vc], acc);
        const size_t o = ((size_t)s * T + i) * C + col;
        if (out_dtype == BG_BF16) reinterpret_cast<__bf16*>(out)[o] = (__bf16)acc;
        else if (out_dtype == BG_F16) reinterpret_cast<_Float16*>(out)[o] = (_Float16)acc;
        else reinterpret_cast<float*>(out)[o] = acc;
    }
}

static inline int cap_grid(size_t work) {
    const size_t b = (work + 255) / 256;
    return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace bg

// ---- C ABI ------------------------------------------------------------------------------------------------------
extern "C" int bg_groupnorm_stats(const float* x, float* stats, int S, int P, int C, int G, float eps, bg_stream_t stream) {
    BG_REQUIRE(x && stats, BG_E_ARG, "bg_groupnorm_stats: null pointer");
    BG_REQUIRE(S > 0 && P > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0, BG_E_SHAPE,
               "bg_groupnorm_stats: need C %% G == 0 and (C/G) %% 4 == 0 (S=%d P=%d C=%d G=%d)", S, P, C, G);
    bg::ProfScope prof(bg::PK_MISC, 0.0, 8.0 * S * (double)P * C, (hipStream_t)stream);
    const int c4n = C / 4;
    if (G > 1 && G <= 256 && c4n <= 256 && 256 % c4n == 0 && (long long)P * c4n >= 256)
        hipLaunchKernelGGL(bg::gn_stats_rows_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, x, stats, P, C, G, eps);
    else
        hipLaunchKernelGGL(bg::gn_stats_kernel, dim3((S * G + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, stats, S, P, C, G, eps);
    return bg::launch_status("groupnorm_stats");
}

extern "C" int bg_im2col(const float* x, void* out, int out_dtype, int S, int Hin, int Win, int C, int kh, int kw, int up,
                         int stride, int pad_y, int pad_x, int Ho, int Wo, const float* stats, const float* gamma,
                         const float* beta, int G, int act, const float* add, bg_stream_t stream) {
    BG_REQUIRE(x && out, BG_E_ARG, "bg_im2col: null pointer");
    BG_REQUIRE(S > 0 && Hin > 0 && Win > 0 && C > 0 && kh > 0 && kw > 0 && (up == 0 || up == 1) && stride >= 1 &&
                   pad_y >= 0 && pad_x >= 0 && Ho > 0 && Wo > 0, BG_E_SHAPE, "bg_im2col: bad shape");
    BG_REQUIRE(out_dtype == BG_F32 || out_dtype == BG_BF16 || out_dtype == BG_F16, BG_E_DTYPE, "bg_im2col: out dtype %d", out_dtype);
    BG_REQUIRE(stats == nullptr || (gamma && beta && G > 0 && C % G == 0 && (C / G) % 4 == 0), BG_E_ARG,
               "bg_im2col: normalisation needs gamma, beta and (C/G) %% 4 == 0");
    BG_REQUIRE(add == nullptr || (kh == 1 && kw == 1 && C % 4 == 0), BG_E_ARG, "bg_im2col: residual add needs a 1x1 window");
    bg::Im2colArgs a{x, out, out_dtype, S, Hin, Win, C, kh, kw, up, stride, pad_y, pad_x, Ho, Wo, stats, gamma, beta, G, act, add};
    const size_t rows = (size_t)S * Ho * Wo;
    bg::ProfScope prof(bg::PK_MISC, 0.0, rows * (double)kh * kw * C * (4.0 + (out_dtype == BG_F32 ? 4.0 : 2.0)),
                       (hipStream_t)stream);
    const unsigned long long total4 = rows * (unsigned long long)(C / 4);
    if (C % 4 == 0 && kh == 1 && kw == 1 && up == 0 && stride == 1 && pad_y == 0 && pad_x == 0 && Ho == Hin && Wo == Win &&
        total4 < (1ull << 32) - 1024) {
        hipLaunchKernelGGL(bg::norm_act_kernel, dim3((unsigned)((total4 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, a,
                           (unsigned)total4, (unsigned)((size_t)Hin * Win * (C / 4)), (unsigned)(C / 4));
    } else if (C % 4 == 0) {
        const unsigned long long work = (unsigned long long)rows * kh * kw * (C / 4);
        if (work < (1ull << 32) - (1ull << 22))               // grid stride 8192 x 256 = 2^21: no wrap-around of the 32-bit index
            hipLaunchKernelGGL(bg::im2col_kernel<unsigned>, dim3(bg::cap_grid(work)), dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(bg::im2col_kernel<size_t>, dim3(bg::cap_grid(work)), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        BG_REQUIRE(stats == nullptr, BG_E_SHAPE, "bg_im2col: normalised input needs C %% 4 == 0");
        hipLaunchKernelGGL(bg::im2col_scalar_kernel, dim3(bg::cap_grid(rows * kh * kw * C)), dim3(256), 0, (hipStream_t)stream, a);
    }
    return bg::launch_status("im2col");
}

extern "C" int bg_upsample1d_cubic(const float* x, float* y, int S, int L, int C, bg_stream_t stream) {
    BG_REQUIRE(x && y && S > 0 && L >= 3 && C > 0, BG_E_ARG, "bg_upsample1d_cubic: bad arguments (reflect pad needs L >= 3)");
    bg::ProfScope prof(bg::PK_MISC, 0.0, 12.0 * S * (double)L * C, (hipStream_t)stream);
    const unsigned long long total4 = (unsigned long long)S * 2 * L * (C / 4);
    if (C % 4 == 0 && total4 < (1ull << 32) - 256)
        hipLaunchKernelGGL(bg::upsample1d_cubic4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                           (unsigned)total4, (unsigned)L, (unsigned)(C / 4));
    else
        hipLaunchKernelGGL(bg::upsample1d_cubic_kernel, dim3(bg::cap_grid((size_t)S * 2 * L * C)), dim3(256), 0, (hipStream_t)stream,
                           x, y, S, L, C);
    return bg::launch_status("upsample1d_cubic");
}

extern "C" int bg_downsample1d_cubic(const float* x, float* y, int S, int L, int C, bg_stream_t stream) {
    BG_REQUIRE(x && y && S > 0 && L >= 4 && (L & 1) == 0 && C > 0, BG_E_ARG, "bg_downsample1d_cubic: bad arguments (even L >= 4)");
    bg::ProfScope prof(bg::PK_MISC, 0.0, 6.0 * S * (double)L * C, (hipStream_t)stream);
    hipLaunchKernelGGL(bg::downsample1d_cubic_kernel, dim3(bg::cap_grid((size_t)S * (L / 2) * C)), dim3(256), 0,
                       (hipStream_t)stream, x, y, S, L, C);
    return bg::launch_status("downsample1d_cubic");
}

extern "C" int bg_small_attn(const float* qkv, int ld, void* out, int out_dtype, int S, int T, int C, int nh, float scale,
                             bg_stream_t stream) {
    BG_REQUIRE(qkv && out && S > 0 && T > 0 && nh > 0 && C % nh == 0 && ld >= 3 * C, BG_E_ARG, "bg_small_attn: bad arguments");
    BG_REQUIRE(nh * T * T <= 8192, BG_E_SHAPE, "bg_small_attn: nh*T*T = %d exceeds the LDS score buffer", nh * T * T);
    BG_REQUIRE(out_dtype == BG_F32 || out_dtype == BG_BF16 || out_dtype == BG_F16, BG_E_DTYPE, "bg_small_attn: out dtype %d", out_dtype);
    bg::ProfScope prof(bg::PK_MISC, 4.0 * S * (double)T * T * C, 16.0 * S * (double)T * C, (hipStream_t)stream);
    const int d = C / nh;
    const size_t staged = d % 4 == 0 ? ((size_t)T * bg::sa_row_stride(C, d) + (size_t)nh * T * T) * sizeof(float) : ~(size_t)0;
    if (staged <= 40 * 1024 && C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)qkv & 15) == 0)        // >= 4 workgroups per CU
        hipLaunchKernelGGL(bg::small_attn_lds_kernel, dim3(S), dim3(256), staged, (hipStream_t)stream, qkv, ld, out, out_dtype, T, C,
                           nh, scale);
    else
        hipLaunchKernelGGL(bg::small_attn_kernel, dim3(S), dim3(256), (size_t)nh * T * T * sizeof(float), (hipStream_t)stream, qkv, ld,
                           out, out_dtype, T, C, nh, scale);
    return bg::launch_status("small_attn");
}
