// Second half of every denoiser's output MLP in ONE launch (fc_out, network.py:1094-1099 / 1162-1167 / 1243-1248 / 1341-1346:
// Linear(768,768) -> LayerNorm -> SiLU -> Linear(768, c), c = 6 / 18 / 48):
//
//   eps[m, :c] = W3 . SiLU(LayerNorm(t0[m, :])) + b3          t0 = the 16-bit output of fc_out.0 (whose GEMM carries the final
//                                                             nn.LayerNorm of the encoder folded into its epilogue)
//
// As separate launches this was a LayerNorm kernel that wrote [M, 768] back (and the GEMM before it wrote fp32: 94 + 47 MB per
// evaluation at 512 x 60 tokens) and a 128 x 64-tile GEMM for a 6..48-column product: 46-48 us per evaluation for 0.1 % of its FLOPs.
// Here a wave owns 16 tokens: it reads their 16-bit rows ONCE (24 x 16-byte loads per lane, all in flight together), takes the
// two-pass row statistics in registers (fp32, like torch.nn.LayerNorm), normalises / activates slice by slice and feeds the 16-bit
// result straight into v_mfma_f32_16x16x32 against W3's rows from LDS -- the [M, 768] activation is never written.
//   A operand (tokens): lane l holds row l & 15, the 8 consecutive k of chunk 4 s + (l >> 4) of k-slice s -- exactly what its s-th
//   load fetched; B operand (W3 rows): the same k from the LDS image [16 NT][768], 16-byte chunks XOR-swizzled by the row so that the
//   16 lanes of a ds_read_b128 phase hit 16 distinct slots.  D: lane l holds output column l & 15 of tokens 4 (l >> 4) .. + 3.
// HBM-bound on the one read of t0 (1.5 KB per token); the SiLU's exp / rcp make it VALU-heavy (about 11 VALU per element).
// Variable-length batches: rows = the compact tokens (count on the device), the result rows are scattered through row_map into
// the zero-filled padded output.
#include "gemm16.h"
#include <math.h>

namespace bg {

typedef __attribute__((ext_vector_type(4))) unsigned ot_u32x4;

template <bool F16, int NT>
__global__ __launch_bounds__(512) void ln_silu_out_kernel(const void* __restrict__ t0, const float* __restrict__ gam,
                                                          const float* __restrict__ bet, const void* __restrict__ w3,
                                                          const float* __restrict__ b3, float* __restrict__ out, int n_out, int M,
                                                          const int* __restrict__ m_dev, const int* __restrict__ row_map, float eps) {
    using E = Elem<F16>;
    using V8 = typename E::V8;
    constexpr int D = BG_D_MODEL;                                  // 768
    __shared__ __attribute__((aligned(16))) unsigned char w_img[NT * 16 * D * 2];
    __shared__ __attribute__((aligned(16))) float gb[2][D];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int Mv = m_dev ? *m_dev : M;

    // the wave's first 16 tokens are requested BEFORE the LDS images are filled (24 x 16-byte loads per lane in flight): at the
    // face-LDM sizes a wave owns one or two blocks, so this is most of the overlap there is
    const int n_blk = (Mv + 15) >> 4;
    int blk = blockIdx.x * 8 + wave;
    ot_u32x4 v[24];
    auto request = [&](int b) {
        int row = (b << 4) + r16;
        row = row < Mv ? row : Mv - 1;
        row = row < 0 ? 0 : row;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(t0) + ((size_t)row * D + g * 8) * 2;
#pragma unroll
        for (int s = 0; s < 24; ++s) v[s] = *reinterpret_cast<const ot_u32x4*>(src + s * 64);
    };
    if (blk < n_blk) request(blk);

    // ---- W3 rows (the first 16 NT of the padded matrix) -> LDS, chunk' = chunk ^ (row & 15) inside its aligned group of 16 chunks;
    // gamma, beta beside them ----
    for (int c = threadIdx.x; c < NT * 16 * (D / 8); c += 512) {
        const int row = c / (D / 8), ch = c % (D / 8);
        const ot_u32x4 w = *reinterpret_cast<const ot_u32x4*>(reinterpret_cast<const unsigned char*>(w3) + ((size_t)row * D + ch * 8) * 2);
        *reinterpret_cast<ot_u32x4*>(w_img + row * (D * 2) + ((ch ^ (row & 15)) << 4)) = w;
    }
    for (int c = threadIdx.x; c < D; c += 512) { gb[0][c] = gam[c]; gb[1][c] = bet[c]; }
    __syncthreads();

    float bias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bias[t] = (t * 16 + r16) < n_out ? b3[t * 16 + r16] : 0.f;

    for (; blk < n_blk; blk += gridDim.x * 8) {
        const int r0 = blk << 4;

        // ---- row statistics, two passes over the registers; a row lives in the four lanes r16, r16 + 16, + 32, + 48 ----
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            float f[4];
            unpack4_16<F16>(make_uint2(v[s][0], v[s][1]), f);
            sum += (f[0] + f[1]) + (f[2] + f[3]);
            unpack4_16<F16>(make_uint2(v[s][2], v[s][3]), f);
            sum += (f[0] + f[1]) + (f[2] + f[3]);
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / D);
        // (the packed rows are made opaque between the passes: hipcc would otherwise keep all 192 unpacked floats of pass 1 alive
        //  for passes 2 and 3 -- 400+ dwords of scratch)
#pragma unroll
        for (int s = 0; s < 24; ++s) asm volatile("" : "+v"(v[s]));
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            float f[4];
            unpack4_16<F16>(make_uint2(v[s][0], v[s][1]), f);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = f[e] - mean; sq += d * d; }
            unpack4_16<F16>(make_uint2(v[s][2], v[s][3]), f);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = f[e] - mean; sq += d * d; }
        }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / D) + eps);
#pragma unroll
        for (int s = 0; s < 24; ++s) asm volatile("" : "+v"(v[s]));

        // ---- normalise, SiLU, round to the operand dtype, multiply ----
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            __builtin_amdgcn_sched_barrier(0);                     // (one slice at a time: hoisting all 24 slices' LDS reads spills)
            const int k0 = s * 32 + g * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(&gb[0][k0]), g1 = *reinterpret_cast<const float4*>(&gb[0][k0 + 4]);
            const float4 e0 = *reinterpret_cast<const float4*>(&gb[1][k0]), e1 = *reinterpret_cast<const float4*>(&gb[1][k0 + 4]);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
            float f[8];
            {
                float a4[4], b4[4];
                unpack4_16<F16>(make_uint2(v[s][0], v[s][1]), a4);
                unpack4_16<F16>(make_uint2(v[s][2], v[s][3]), b4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[e] = a4[e]; f[4 + e] = b4[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = silu_rcp((f[e] - mean) * rstd * gg[e] + bb[e]);
            union { V8 v8; uint2 u[2]; } a;
            a.u[0] = pack4_16(f[0], f[1], f[2], f[3], F16 ? BG_F16 : BG_BF16);
            a.u[1] = pack4_16(f[4], f[5], f[6], f[7], F16 ? BG_F16 : BG_BF16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const V8 b = *reinterpret_cast<const V8*>(w_img + (t * 16 + r16) * (D * 2) + (((s * 4 + g) ^ r16) << 4));
                if (F16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v8, b, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v8, b, acc[t], 0, 0, 0);
            }
        }

        // ---- D: column 16 t + r16 of tokens r0 + 4 g + r ----
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = r0 + g * 4 + r;
            if (tok >= Mv) continue;
            const size_t orow = row_map ? (size_t)row_map[tok] : (size_t)tok;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 16 + r16;
                if (col < n_out) out[orow * n_out + col] = acc[t][r] + bias[t];
            }
        }
        if (blk + (int)gridDim.x * 8 < n_blk) request(blk + gridDim.x * 8);
    }
}

bool ln_silu_out_supported(int n_out, int n_out_pad) { return n_out >= 1 && n_out <= 48 && n_out_pad >= ((n_out + 15) & ~15); }

// t0 [M, 768] 16-bit rows of `dtype`; w3 [n_out_pad, 768] of the same dtype, b3 [n_out_pad]; out fp32 [*, n_out] (row_map != null:
// row r of the result goes to out[row_map[r]]); m_dev: device-side row count (M stays the bound that sizes the grid)
int ln_silu_out(const void* t0, const float* gam, const float* bet, const void* w3, const float* b3, float* out, int n_out, int n_out_pad,
                int M, int dtype, float eps, hipStream_t s, const int* m_dev, const int* row_map, double rows_hint) {
    if (M <= 0) return 0;
    BG_REQUIRE(dtype == BG_BF16 || dtype == BG_F16, BG_E_DTYPE, "ln_silu_out: 16-bit rows expected (dtype %d)", dtype);
    BG_REQUIRE(ln_silu_out_supported(n_out, n_out_pad), BG_E_SHAPE, "ln_silu_out: 1 .. 48 output columns expected (n_out = %d, padded %d)", n_out, n_out_pad);
    BG_REQUIRE((((uintptr_t)t0 | (uintptr_t)w3) & 15) == 0, BG_E_ALIGN, "ln_silu_out: t0 / w3 must be 16-byte aligned");
    const int nt = (n_out + 15) >> 4;
    const int blocks = (M + 127) / 128;                            // 8 waves x 16 tokens per pass
    const int grid = blocks < 256 ? blocks : 256;
    const double rows = rows_hint > 0 ? rows_hint : (double)M;
    ProfScope prof(PK_OUT_TAIL, 2.0 * rows * BG_D_MODEL * n_out, rows * (2.0 * BG_D_MODEL + 4.0 * n_out) + 2.0 * n_out * BG_D_MODEL, s);
    const bool f16 = dtype == BG_F16;
#define OT_LAUNCH(F, N) hipLaunchKernelGGL((ln_silu_out_kernel<F, N>), dim3(grid), dim3(512), 0, s, t0, gam, bet, w3, b3, out, n_out, M, m_dev, row_map, eps)
    if (nt == 1) { if (f16) OT_LAUNCH(true, 1); else OT_LAUNCH(false, 1); }
    else if (nt == 2) { if (f16) OT_LAUNCH(true, 2); else OT_LAUNCH(false, 2); }
    else { if (f16) OT_LAUNCH(true, 3); else OT_LAUNCH(false, 3); }
#undef OT_LAUNCH
    return launch_status("ln_silu_out");
}

}  // namespace bg

extern "C" int bg_ln_silu_out_fwd(const void* t0, const float* ln_g, const float* ln_b, const void* w3, const float* b3, float* out,
                                  int n_out, int n_out_pad, int rows, int dtype, float eps, bg_stream_t stream) {
    BG_REQUIRE(t0 && ln_g && ln_b && w3 && b3 && out, BG_E_ARG, "bg_ln_silu_out_fwd: null pointer");
    return bg::ln_silu_out(t0, ln_g, ln_b, w3, b3, out, n_out, n_out_pad, rows, dtype, eps, (hipStream_t)stream, nullptr, nullptr, 0.0);
}
