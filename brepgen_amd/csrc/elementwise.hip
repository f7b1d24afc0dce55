// HBM-bound kernels of the denoising path: LayerNorm(768), sincos timestep embedding, the fused
// CFG + DDPM / PNDM scheduler updates.  All are judged in GB/s, not FLOP/s: 16-byte accesses per
// lane, one wave per row (LayerNorm) or a capped grid-stride loop (elementwise).
#include "bg_common.h"
#include <math.h>

namespace bg {

// ------------------------------------------------------------------------------------------------
// LayerNorm over 768 columns, one wave64 per row, 4 rows per 256-thread block.
// Each lane owns 3 float4 (columns lane*4 + 256*j): fully coalesced 1 KiB per wave-load.
// Two-pass statistics in registers (mean, then centred variance) == torch.nn.LayerNorm numerics.
// ------------------------------------------------------------------------------------------------
template <int OUT, bool SILU, bool SPLIT_IN = false>    // OUT: BG_F32 | BG_F16 | BG_BF16
__global__ __launch_bounds__(256) void ln768_kernel(const float* __restrict__ x, const void* __restrict__ x_lo,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, void* __restrict__ y, int M,
                                                    float eps, const int* __restrict__ m_dev) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (m_dev ? *m_dev : M)) return;                     // compacted batch: the row count lives on the device
    float4 v[3];
    if (SPLIT_IN) {         // x = hi + lo, two 16-bit planes of dtype OUT (the denoisers' split residual stream)
        const uint2* hr = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + (size_t)row * 768);
        const uint2* lr = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x_lo) + (size_t)row * 768);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float fh[4], fl[4];
            unpack4_16<OUT == BG_F16>(hr[lane + 64 * j], fh);
            unpack4_16<OUT == BG_F16>(lr[lane + 64 * j], fl);
            v[j] = make_float4(fh[0] + fl[0], fh[1] + fl[1], fh[2] + fl[2], fh[3] + fl[3]);
        }
    } else {
        const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * 768);
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = xr[lane + 64 * j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float mean = wave_sum(s) * (1.0f / 768.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / 768.0f) + eps);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float4 g = g4[lane + 64 * j], b = b4[lane + 64 * j];
        float4 o;
        o.x = v[j].x * rstd * g.x + b.x;
        o.y = v[j].y * rstd * g.y + b.y;
        o.z = v[j].z * rstd * g.z + b.z;
        o.w = v[j].w * rstd * g.w + b.w;
        if (SILU) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
        if (OUT != BG_F32) {
            uint2* yr = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + (size_t)row * 768);
            yr[lane + 64 * j] = pack4_16(o.x, o.y, o.z, o.w, OUT);
        } else {
            float4* yr = reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)row * 768);
            yr[lane + 64 * j] = o;
        }
    }
}

int layernorm768(const float* x, const float* g, const float* b, void* y, int y_dtype, int M, float eps, int silu,
                 hipStream_t s, const int* m_dev, double rows_hint) {
    if (M <= 0) return 0;
    dim3 grid((M + 3) / 4), block(256);
    ProfScope prof(PK_LAYERNORM, 0.0, (rows_hint > 0 ? rows_hint : (double)M) * 768 * (4.0 + (y_dtype == BG_F32 ? 4.0 : 2.0)), s);
    if (y_dtype == BG_BF16) {
        if (silu) hipLaunchKernelGGL((ln768_kernel<BG_BF16, true>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
        else hipLaunchKernelGGL((ln768_kernel<BG_BF16, false>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
    } else if (y_dtype == BG_F16) {
        if (silu) hipLaunchKernelGGL((ln768_kernel<BG_F16, true>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
        else hipLaunchKernelGGL((ln768_kernel<BG_F16, false>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
    } else if (y_dtype == BG_F32) {
        if (silu) hipLaunchKernelGGL((ln768_kernel<BG_F32, true>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
        else hipLaunchKernelGGL((ln768_kernel<BG_F32, false>), grid, block, 0, s, x, (const void*)nullptr, g, b, y, M, eps, m_dev);
    } else {
        set_error("layernorm: unsupported output dtype %d", y_dtype);
        return BG_E_DTYPE;
    }
    return launch_status("layernorm768");
}

int layernorm768_split(const void* hi, const void* lo, const float* g, const float* b, void* y, int y_dtype, int M,
                       float eps, hipStream_t s, const int* m_dev, double rows_hint) {
    if (M <= 0) return 0;
    dim3 grid((M + 3) / 4), block(256);
    ProfScope prof(PK_LAYERNORM, 0.0, (rows_hint > 0 ? rows_hint : (double)M) * 768 * 6.0, s);
    const float* x = reinterpret_cast<const float*>(hi);
    if (y_dtype == BG_BF16) hipLaunchKernelGGL((ln768_kernel<BG_BF16, false, true>), grid, block, 0, s, x, lo, g, b, y, M, eps, m_dev);
    else if (y_dtype == BG_F16) hipLaunchKernelGGL((ln768_kernel<BG_F16, false, true>), grid, block, 0, s, x, lo, g, b, y, M, eps, m_dev);
    else {
        set_error("layernorm (split input): unsupported dtype %d", y_dtype);
        return BG_E_DTYPE;
    }
    return launch_status("layernorm768_split");
}

// ------------------------------------------------------------------------------------------------
// sincos_embedding (network.py:1043-1063): freqs_i = exp(-ln(1e4) * i / 384); [cos | sin].
// ------------------------------------------------------------------------------------------------
__global__ void sincos_kernel(const int64_t* __restrict__ t, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 384) return;
    const int r = i / 384, c = i % 384;
    // same op order as the reference: exp(-log(10000) * arange / half), args = t.float() * freqs
    // (fp32 product, fp32 divide, then a correctly-rounded fp32 exp: t*freq reaches ~1e3, so one ulp of freq
    //  moves the cos/sin argument by up to 6e-5 -- keep freq as close to the CPU reference as possible)
    const float fa = -9.210340371976184f * (float)c;
    const float freq = (float)exp((double)(fa / 384.0f));
    const float arg = (float)t[r] * freq;
    out[(size_t)r * 768 + c] = cosf(arg);
    out[(size_t)r * 768 + 384 + c] = sinf(arg);
}

int sincos_embed(const int64_t* t, int n, float* out, hipStream_t s) {
    if (n <= 0) return 0;
    const int total = n * 384;
    hipLaunchKernelGGL(sincos_kernel, dim3((total + 255) / 256), dim3(256), 0, s, t, n, out);
    return launch_status("sincos_embed");
}

__global__ void cond_vector_kernel(const float* __restrict__ temb, int nt, const float* __restrict__ cls,
                                   const int64_t* __restrict__ label, float* __restrict__ c, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 768) return;
    const int b = i / 768, col = i % 768;
    float v = temb[(size_t)(nt == 1 ? 0 : b) * 768 + col];
    if (cls != nullptr) v += cls[(size_t)label[b] * 768 + col];
    c[i] = v;
}

int cond_vector(const float* temb, int nt, const float* cls, const int64_t* label, float* c, int B, hipStream_t s) {
    const int total = B * 768;
    hipLaunchKernelGGL(cond_vector_kernel, dim3((total + 255) / 256), dim3(256), 0, s, temb, nt, cls, label, c, B);
    return launch_status("cond_vector");
}

__global__ void cond_vector_table_kernel(const float* __restrict__ table, int table_rows, const int64_t* __restrict__ ts, int nt,
                                         const float* __restrict__ cls, const int64_t* __restrict__ label, float* __restrict__ c, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 768) return;
    const int b = i / 768, col = i % 768;
    const int64_t t = ts[nt == 1 ? 0 : b];
    float v = (t >= 0 && t < table_rows) ? table[(size_t)t * 768 + col] : __builtin_nanf("");
    if (cls != nullptr) v += cls[(size_t)label[b] * 768 + col];
    c[i] = v;
}

int cond_vector_table(const float* table, int table_rows, const int64_t* timesteps, int nt, const float* cls, const int64_t* label,
                      float* c, int B, hipStream_t s) {
    const int total = B * 768;
    hipLaunchKernelGGL(cond_vector_table_kernel, dim3((total + 255) / 256), dim3(256), 0, s, table, table_rows, timesteps, nt, cls, label, c, B);
    return launch_status("cond_vector_table");
}

__global__ void cast_f32_bf16_kernel(const float4* __restrict__ in, bf16x4* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        out[i] = to_bf16x4(v.x, v.y, v.z, v.w);
    }
}

int cast_f32_bf16(const float* in, void* out, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                       reinterpret_cast<bf16x4*>(out), n4);
    return launch_status("cast_f32_bf16");
}

// ------------------------------------------------------------------------------------------------
// Scheduler updates.  One pass: every operand is read once with 16-byte loads, the result is written
// once.  Algorithmic traffic: DDPM 4 tensors (eps, x, noise, out) = 16 B/element (+4 with CFG);
// PNDM 3..7 tensors.  Grid capped at 2048 blocks, grid-stride.
// ------------------------------------------------------------------------------------------------
struct DdpmArgs {
    const float* eps_c; const float* eps_u; float w;
    const float* x; const float* noise; float* out; size_t n;
    float sqrt_alpha_prod, sqrt_beta_prod, x0_coeff, xt_coeff, sigma, clip;
};

__device__ __forceinline__ float ddpm_one(float e, float x, float z, const DdpmArgs& a, bool has_noise) {
    float x0 = (x - a.sqrt_beta_prod * e) / a.sqrt_alpha_prod;
    if (a.clip > 0.f) x0 = fminf(fmaxf(x0, -a.clip), a.clip);
    float o = a.x0_coeff * x0 + a.xt_coeff * x;
    if (has_noise) o += a.sigma * z;
    return o;
}

template <bool VEC>
__global__ __launch_bounds__(256) void ddpm_step_kernel(DdpmArgs a) {
    const bool cfg = a.eps_u != nullptr;
    const bool has_noise = a.noise != nullptr && a.sigma != 0.f;
    const float wc = 1.0f + a.w;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (VEC) {
        const size_t n4 = a.n / 4;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 e = reinterpret_cast<const float4*>(a.eps_c)[i];
            if (cfg) {
                const float4 u = reinterpret_cast<const float4*>(a.eps_u)[i];
                e.x = e.x * wc - u.x * a.w; e.y = e.y * wc - u.y * a.w;
                e.z = e.z * wc - u.z * a.w; e.w = e.w * wc - u.w * a.w;
            }
            const float4 x = reinterpret_cast<const float4*>(a.x)[i];
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_noise) z = reinterpret_cast<const float4*>(a.noise)[i];
            float4 o;
            o.x = ddpm_one(e.x, x.x, z.x, a, has_noise);
            o.y = ddpm_one(e.y, x.y, z.y, a, has_noise);
            o.z = ddpm_one(e.z, x.z, z.z, a, has_noise);
            o.w = ddpm_one(e.w, x.w, z.w, a, has_noise);
            reinterpret_cast<float4*>(a.out)[i] = o;
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += stride) {
            float e = a.eps_c[i];
            if (cfg) e = e * wc - a.eps_u[i] * a.w;
            a.out[i] = ddpm_one(e, a.x[i], has_noise ? a.noise[i] : 0.f, a, has_noise);
        }
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int ddpm_step(const DdpmArgs& a, hipStream_t s) {
    if (a.n == 0) return 0;
    const bool vec = (a.n % 4 == 0) && aligned16(a.eps_c) && aligned16(a.eps_u) && aligned16(a.x) &&
                     aligned16(a.noise) && aligned16(a.out);
    ProfScope prof(PK_DDPM_STEP, 0.0, 4.0 * a.n * (3.0 + (a.eps_u ? 1 : 0) + (a.noise ? 1 : 0)), s);
    const size_t work = vec ? a.n / 4 : a.n;
    const int grid = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
    if (vec) hipLaunchKernelGGL(ddpm_step_kernel<true>, dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(ddpm_step_kernel<false>, dim3(grid), dim3(256), 0, s, a);
    return launch_status("cfg_ddpm_step");
}

struct PndmArgs {
    const float* eps_c; const float* eps_u; float w;
    const float* x;
    float* e_store;
    const float* acc; float* acc_out; float acc_scale_old, acc_scale_e;
    float c_e, c_acc;
    const float* h0; const float* h1; const float* h2; float c_h0, c_h1, c_h2;
    float sample_coeff, eps_coeff;
    float* out; size_t n;
};

__global__ __launch_bounds__(256) void pndm_step_kernel(PndmArgs a) {
    const bool cfg = a.eps_u != nullptr;
    const float wc = 1.0f + a.w;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += stride) {
        float e = a.eps_c[i];
        if (cfg) e = e * wc - a.eps_u[i] * a.w;
        const float acc = a.acc ? a.acc[i] : 0.f;
        float comb = a.c_e * e;
        if (a.c_acc != 0.f) comb += a.c_acc * acc;
        if (a.h0) comb += a.c_h0 * a.h0[i];
        if (a.h1) comb += a.c_h1 * a.h1[i];
        if (a.h2) comb += a.c_h2 * a.h2[i];
        const float o = a.sample_coeff * a.x[i] - a.eps_coeff * comb;
        if (a.e_store) a.e_store[i] = e;
        if (a.acc_out) a.acc_out[i] = a.acc_scale_old * acc + a.acc_scale_e * e;
        a.out[i] = o;
    }
}

int pndm_step(const PndmArgs& a, hipStream_t s) {
    if (a.n == 0) return 0;
    const int grid = (int)((a.n + 255) / 256 < 4096 ? (a.n + 255) / 256 : 4096);
    ProfScope prof(PK_PNDM_STEP, 0.0, 4.0 * a.n * (3.0 + (a.eps_u ? 1 : 0) + (a.e_store ? 1 : 0) + (a.acc ? 1 : 0) +
                   (a.acc_out ? 1 : 0) + (a.h0 ? 1 : 0) + (a.h1 ? 1 : 0) + (a.h2 ? 1 : 0)), s);
    hipLaunchKernelGGL(pndm_step_kernel, dim3(grid), dim3(256), 0, s, a);
    return launch_status("pndm_step");
}

// DDPMScheduler.add_noise (trainer.py:348,...): out[b,:] = sqrt_acp[t_b] * x0[b,:] + sqrt(1-acp[t_b]) * noise[b,:]
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                        const float* __restrict__ sa, const float* __restrict__ sb,
                                                        float* __restrict__ out, size_t per, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per;
        out[i] = sa[b] * x0[i] + sb[b] * noise[i];
    }
}

int add_noise(const float* x0, const float* noise, const float* sa, const float* sb, float* out, size_t per, size_t n,
              hipStream_t s) {
    if (n == 0) return 0;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_noise_kernel, dim3(grid), dim3(256), 0, s, x0, noise, sa, sb, out, per, n);
    return launch_status("add_noise");
}

}  // namespace bg

// ---- C ABI --------------------------------------------------------------------------------------
extern "C" int bg_sincos_embed(const int64_t* timesteps, int n, float* out, bg_stream_t stream) {
    BG_REQUIRE(timesteps && out && n >= 0, BG_E_ARG, "bg_sincos_embed: null pointer or negative n");
    return bg::sincos_embed(timesteps, n, out, (hipStream_t)stream);
}

extern "C" int bg_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int y_dtype, int M,
                                float eps, int fuse_silu, bg_stream_t stream) {
    BG_REQUIRE(x && gamma && beta && y && M >= 0, BG_E_ARG, "bg_layernorm_fwd: null pointer or negative M");
    BG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
                   ((uintptr_t)beta & 15) == 0, BG_E_ALIGN, "bg_layernorm_fwd: pointers must be 16-byte aligned");
    return bg::layernorm768(x, gamma, beta, y, y_dtype, M, eps, fuse_silu, (hipStream_t)stream);
}

namespace bg {
// Masked mean-squared error of the trainers' loss / validation forward (trainer.py:354, 538, 597, 950-952):
// rows with row_mask != 0 are skipped, columns [col0, col0 + ncols) are compared.  Deterministic: a fixed grid writes
// per-block (sum of squares, valid rows) partials in double, one block adds them in a fixed order.
constexpr int MSE_BLOCKS = 512;
__global__ __launch_bounds__(256) void masked_sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const uint8_t* __restrict__ mask, long long rows, int ld,
                                                            int col0, int ncols, double* __restrict__ partial) {
    __shared__ double sh[2][256];
    double s = 0.0, n = 0.0;
    for (long long r = blockIdx.x * 256ll + threadIdx.x; r < rows; r += 256ll * MSE_BLOCKS) {
        if (mask != nullptr && mask[r]) continue;
        const float* pa = a + r * ld + col0;
        const float* pb = b + r * ld + col0;
        float q = 0.f;
        for (int c = 0; c < ncols; ++c) { const float d = pa[c] - pb[c]; q += d * d; }
        s += (double)q;
        n += 1.0;
    }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sh[0][0]; partial[2 * blockIdx.x + 1] = sh[1][0]; }
}
__global__ __launch_bounds__(256) void masked_mse_final_kernel(const double* __restrict__ partial, int ncols,
                                                               float* __restrict__ out) {
    __shared__ double sh[2][256];
    double s = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < MSE_BLOCKS; i += 256) { s += partial[2 * i]; n += partial[2 * i + 1]; }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double S = sh[0][0], N = sh[1][0];
        out[0] = N > 0.0 ? (float)(S / (N * ncols)) : 0.f;      // nn.MSELoss()(pred[~mask], target[~mask])
        out[1] = (float)(S / ncols);                            // mse(reduction='none')(...).mean(-1).sum()
        out[2] = (float)N;                                      // valid rows
    }
}
}  // namespace bg

extern "C" int bg_masked_mse(const float* pred, const float* target, const uint8_t* row_mask, long long rows, int ld,
                             int col0, int ncols, double* scratch, float* out3, bg_stream_t stream) {
    BG_REQUIRE(pred && target && scratch && out3, BG_E_ARG, "bg_masked_mse: null pointer");
    BG_REQUIRE(rows >= 0 && ncols > 0 && col0 >= 0 && col0 + ncols <= ld, BG_E_SHAPE,
               "bg_masked_mse: bad shape rows=%lld ld=%d col0=%d ncols=%d", rows, ld, col0, ncols);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bg::masked_sqdiff_kernel, dim3(bg::MSE_BLOCKS), dim3(256), 0, s, pred, target, row_mask, rows, ld,
                       col0, ncols, scratch);
    hipLaunchKernelGGL(bg::masked_mse_final_kernel, dim3(1), dim3(256), 0, s, scratch, ncols, out3);
    return bg::launch_status("bg_masked_mse");
}

extern "C" int bg_layernorm_split_fwd(const void* hi, const void* lo, const float* gamma, const float* beta, void* y,
                                      int dtype, int M, float eps, bg_stream_t stream) {
    BG_REQUIRE(hi && lo && gamma && beta && y, BG_E_ARG, "bg_layernorm_split_fwd: null pointer");
    return bg::layernorm768_split(hi, lo, gamma, beta, y, dtype, M, eps, (hipStream_t)stream);
}

extern "C" int bg_cfg_ddpm_step(const float* eps_c, const float* eps_u, float guidance_w, const float* x,
                                const float* noise, float* out, size_t n, float sqrt_alpha_prod,
                                float sqrt_beta_prod, float x0_coeff, float xt_coeff, float sigma, float clip,
                                bg_stream_t stream) {
    BG_REQUIRE(eps_c && x && out, BG_E_ARG, "bg_cfg_ddpm_step: null pointer");
    BG_REQUIRE(sqrt_alpha_prod > 0.f, BG_E_ARG, "bg_cfg_ddpm_step: sqrt_alpha_prod must be > 0");
    bg::DdpmArgs a{eps_c, eps_u, guidance_w, x, noise, out, n, sqrt_alpha_prod, sqrt_beta_prod,
                   x0_coeff, xt_coeff, sigma, clip};
    return bg::ddpm_step(a, (hipStream_t)stream);
}

extern "C" int bg_pndm_step(const float* eps_c, const float* eps_u, float guidance_w, const float* x,
                            float* e_store, const float* acc, float* acc_out, float acc_scale_old,
                            float acc_scale_e, float c_e, float c_acc, const float* hist0, const float* hist1,
                            const float* hist2, float c_h0, float c_h1, float c_h2, float sample_coeff,
                            float eps_coeff, float* out, size_t n, bg_stream_t stream) {
    BG_REQUIRE(eps_c && x && out, BG_E_ARG, "bg_pndm_step: null pointer");
    bg::PndmArgs a{eps_c, eps_u, guidance_w, x, e_store, acc, acc_out, acc_scale_old, acc_scale_e, c_e, c_acc,
                   hist0, hist1, hist2, c_h0, c_h1, c_h2, sample_coeff, eps_coeff, out, n};
    return bg::pndm_step(a, (hipStream_t)stream);
}

extern "C" int bg_add_noise(const float* x0, const float* noise, const float* sqrt_alpha_prod,
                            const float* sqrt_one_minus_alpha_prod, float* out, int B, size_t per_sample,
                            bg_stream_t stream) {
    BG_REQUIRE(x0 && noise && sqrt_alpha_prod && sqrt_one_minus_alpha_prod && out && B >= 0, BG_E_ARG,
               "bg_add_noise: null pointer");
    return bg::add_noise(x0, noise, sqrt_alpha_prod, sqrt_one_minus_alpha_prod, out, per_sample,
                         (size_t)B * per_sample, (hipStream_t)stream);
}
