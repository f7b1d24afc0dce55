// bf16 MFMA GEMM with fused epilogues -- the QKV / out-proj / FFN1 / FFN2 / embed / fc_out Linears of the
// denoisers (network.py:1076-1099), i.e. >95 % of the path's FLOPs.  MFMA-bound.
//
//   out[m,n] = act(sum_k a[m,k] * w[n,k] + bias[n]) (+ add[(m / add_div), n])        a, w bf16; fp32 accumulate
//
// Design (gfx950):
//   * tile BM x BN x 64, 256 threads = 4 waves, v_mfma_f32_32x32x16_bf16 (16 fp32 accumulators per 32x32 tile);
//   * both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]) so A and B fragments are the
//     same 16-byte-per-lane shape: lane l holds rows (l & 31), k-chunk (l >> 5) of each 16-wide k-slice;
//   * HBM/L2 -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B),
//     double-buffered over K, one barrier per 64-wide K-step;
//   * the LDS image is XOR-swizzled at 16-byte granularity: chunk' = chunk ^ ((row >> 1) & 7).  The DMA writes
//     lane-linear, so the permutation is applied to the per-lane *source* address and again on the ds_read_b128
//     side (same involution).  Every 16-lane service group of ds_read_b128 then touches 16 distinct 16-byte
//     slots of the 256-byte bank row: conflict-free;
//   * epilogue: accumulators -> wave-private LDS patch (ds_write_b32, conflict-free) -> row-wise float4 reads ->
//     bias / ReLU / residual-or-broadcast add -> 16-byte (fp32) or 8-byte (bf16) fully coalesced stores;
//   * 1-D grid, XCD-aware: logical tile ids that share an A row-panel run back-to-back on the same XCD (L2 reuse).
#include "bg_common.h"

namespace bg {

constexpr int G_BK = 64;            // bf16 elements per K-step = 128 bytes per tile row

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
    // dst = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;           // 32x32 MFMA tiles per wave
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.a);
    const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(g.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt_n = g.N_pad / BN;
    const int nblk = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nblk);
    const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;

    // ---- LDS-DMA source addresses (per lane), destination bases (per wave) ----
    constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;           // wave-instructions per wave per stage
    const __bf16* a_src[A_INSTR];
    const __bf16* b_src[B_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;                       // clamp: rows >= M are never stored
        a_src[j] = A + (size_t)grow * g.lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int row = (wave * B_INSTR + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        b_src[j] = W + (size_t)(n0 + row) * g.K + c * 8;
    }
    auto issue = [&](int stage, int k0) {
        unsigned char* sa = lds + stage * STAGE_BYTES;
        unsigned char* sb = sa + BM * 128;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) lds_dma16(a_src[j] + k0, sa + (wave * A_INSTR + j) * 1024);
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) lds_dma16(b_src[j] + k0, sb + (wave * B_INSTR + j) * 1024);
    };

    // ---- fragment read offsets ----
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + (lane & 31);
        a_off[i] = row * 128;
        a_sw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * (TN * 32) + i * 32 + (lane & 31);
        b_off[i] = BM * 128 + row * 128;
        b_sw[i] = (row >> 1) & 7;
    }
    const int h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = g.K / G_BK;
    issue(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces of stage kt landed
        __syncthreads();                                          // ... and everybody else's; stage kt^1 is free
        if (kt + 1 < KT) issue((kt + 1) & 1, (kt + 1) * G_BK);
        const unsigned char* st = lds + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();                                              // all fragment reads done: LDS is free

    // ---- epilogue: accumulators -> wave-private LDS patch -> coalesced rows ----
    constexpr int PW = TN * 32;                                   // patch width (floats)
    float* patch = reinterpret_cast<float*>(lds) + wave * (TM * 32 * PW);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // C/D layout of the 32x32 MFMA
                patch[pr * PW + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    __syncthreads();

    constexpr int LPR = PW / 4;                                   // lanes per patch row (float4 each)
    constexpr int RPI = 64 / LPR;                                 // rows per iteration
    const int rr = lane / LPR, c4 = (lane % LPR) * 4;
    const int gcol = n0 + wn * PW + c4;
    const bool vec = (g.N == g.N_pad) && ((g.ldc & 3) == 0) && (g.add == nullptr || (g.ld_add & 3) == 0) &&
                     (g.add2 == nullptr || (g.ld_add2 & 3) == 0);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bias = *reinterpret_cast<const float4*>(g.bias + gcol);
#pragma unroll 4
    for (int it = 0; it < TM * 32 / RPI; ++it) {
        const int pr = it * RPI + rr;
        const int grow = m0 + wm * (TM * 32) + pr;
        if (grow >= g.M) continue;
        float4 v = *reinterpret_cast<const float4*>(&patch[pr * PW + c4]);
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (g.act == BG_ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (vec) {
            if (g.add) {
                const float4 a4 = *reinterpret_cast<const float4*>(g.add + (size_t)(grow / g.add_div) * g.ld_add + gcol);
                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            if (g.add2) {
                const float4 a4 = *reinterpret_cast<const float4*>(g.add2 + (size_t)(grow / g.add2_div) * g.ld_add2 + gcol);
                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            if (g.out_dtype == BG_BF16)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.out) + (size_t)grow * g.ldc + gcol) =
                    to_bf16x4(v.x, v.y, v.z, v.w);
            else
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (size_t)grow * g.ldc + gcol) = v;
        } else {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = gcol + e;
                if (col >= g.N) continue;
                float o = vv[e];
                if (g.add) o += g.add[(size_t)(grow / g.add_div) * g.ld_add + col];
                if (g.add2) o += g.add2[(size_t)(grow / g.add2_div) * g.ld_add2 + col];
                if (g.out_dtype == BG_BF16) reinterpret_cast<__bf16*>(g.out)[(size_t)grow * g.ldc + col] = (__bf16)o;
                else reinterpret_cast<float*>(g.out)[(size_t)grow * g.ldc + col] = o;
            }
        }
    }
}

int gemm_bf16(const GemmArgs& g, hipStream_t s) {
    if (g.M <= 0) return 0;
    if (g.K % G_BK != 0 || g.N_pad % 64 != 0 || g.lda % 8 != 0) {
        set_error("gemm_bf16: need K %% 64 == 0, N_pad %% 64 == 0, lda %% 8 == 0 (K=%d N_pad=%d lda=%d)", g.K, g.N_pad, g.lda);
        return BG_E_SHAPE;
    }
    if ((reinterpret_cast<uintptr_t>(g.a) & 15) || (reinterpret_cast<uintptr_t>(g.w) & 15) ||
        (reinterpret_cast<uintptr_t>(g.out) & 15)) {
        set_error("gemm_bf16: a / w / out must be 16-byte aligned");
        return BG_E_ALIGN;
    }
    const int mt = (g.M + 127) / 128;
    // algorithmic cost: 2*M*N*K flops; bytes = operands once + output once (+ addends)
    const double osz = g.out_dtype == BG_BF16 ? 2.0 : 4.0;
    const double bytes = 2.0 * g.M * g.K + 2.0 * g.N * (double)g.K + osz * g.M * g.N +
                         (g.add ? 4.0 * (g.M / g.add_div) * g.N : 0.0) + (g.add2 ? 4.0 * (g.M / g.add2_div) * g.N : 0.0);
    ProfScope prof(g.N_pad % 128 == 0 ? PK_GEMM_BF16_128 : PK_GEMM_BF16_64, 2.0 * g.M * g.N * (double)g.K, bytes, s);
    if (g.N_pad % 128 == 0) {
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2>), dim3(mt * (g.N_pad / 128)), dim3(256), 0, s, g);
    } else {
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, 4, 1>), dim3(mt * (g.N_pad / 64)), dim3(256), 0, s, g);
    }
    return launch_status("gemm_bf16");
}

}  // namespace bg
