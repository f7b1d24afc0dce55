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
#include <type_traits>

namespace bg {

constexpr int G_BK = 64;            // bf16 elements per K-step = 128 bytes per tile row

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
    // dst = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// BM x BN block tile, WM x WN waves (each wave (BM/WM) x (BN/WN)), STAGES-deep LDS ring over K.
// ABL (measurement only): 0 = product kernel, 1 = no MFMA, 2 = no LDS-DMA inside the K loop, 3 = no epilogue stores
template <int BM, int BN, int WM, int WN, int STAGES, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;           // 32x32 MFMA tiles per wave
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int EPI_BYTES = BM * BN * 4;
    constexpr int LDS_BYTES = (STAGES * STAGE_BYTES > EPI_BYTES) ? STAGES * STAGE_BYTES : EPI_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA pieces must divide evenly over the waves");
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
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;   // wave-instructions per wave per stage
    constexpr int PER_STAGE = A_INSTR + B_INSTR;
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
    // one 1-KiB DMA piece (p < A_INSTR: activation rows, else weight rows) of K-step k0 into ring slot `stage`
    auto issue_piece = [&](int p, int stage, int k0) {
        unsigned char* sa = lds + stage * STAGE_BYTES;
        if (p < A_INSTR) lds_dma16(a_src[p] + k0, sa + (wave * A_INSTR + p) * 1024);
        else lds_dma16(b_src[p - A_INSTR] + k0, sa + BM * 128 + (wave * B_INSTR + (p - A_INSTR)) * 1024);
    };
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int p = 0; p < PER_STAGE; ++p) issue_piece(p, stage, k0);
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

    // ---- K loop: STAGES-1 tiles of LDS-DMA in flight, ONE barrier per 64-wide K-step.  Counted vmcnt + raw
    // s_barrier: __syncthreads() would drain the DMA queue (vmcnt(0)) at every step. ----
    const int KT = g.K / G_BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) issue(s, s * G_BK);
    int stage = 0;                                                // kt % STAGES
    auto ktile = [&](int kt, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        // tile kt must have landed; up to min(STAGES-2, KT-1-kt) younger tiles may stay in flight
        const int younger = (KT - 1 - kt) < (STAGES - 2) ? (KT - 1 - kt) : (STAGES - 2);
        if (younger >= 2) wait_vmcnt<2 * PER_STAGE>();
        else if (younger == 1) wait_vmcnt<PER_STAGE>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my fragment reads of tile kt-1 are complete
        __builtin_amdgcn_s_barrier();                             // everybody's DMA of tile kt landed; ring slot
                                                                  // (kt-1) % STAGES is free for tile kt+STAGES-1
        // The DMA pieces of tile kt+STAGES-1 are issued one at a time BETWEEN the MFMAs of this K-step: an LDS-DMA
        // instruction costs ~60-180 issue cycles, and both waves of a SIMD leave the barrier together, so issuing
        // all pieces up front would idle the matrix pipe for that long every K-step.
        int ns = stage + STAGES - 1;
        ns = ns >= STAGES ? ns - STAGES : ns;
        const int k0n = (kt + STAGES - 1) * G_BK;
        const unsigned char* st = lds + stage * STAGE_BYTES;
        bf16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[buf][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[buf][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
        };
        constexpr int NMFMA = 4 * TM * TN;
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ABL == 1) {
                        asm volatile("" ::"v"(af[ks & 1][i]), "v"(bf[ks & 1][j]));
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                    }
                    {   // DMA pieces [lo, hi) are scheduled right after MFMA number idx of NMFMA (compile-time)
                        const int idx = (ks * TM + i) * TN + j;
                        const int lo = idx * PER_STAGE / NMFMA, hi = (idx + 1) * PER_STAGE / NMFMA;
                        if (hi > lo) {
                            if (more) {
#pragma unroll
                                for (int p = lo; p < hi; ++p) issue_piece(p, ns, k0n);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    };
    // steady state issues the DMA pieces of tile kt+STAGES-1; the last STAGES-1 K-steps have nothing left to fetch
    int kt = 0;
    if (ABL != 2)
        for (; kt + STAGES - 1 < KT; ++kt) ktile(kt, std::true_type{});
    for (; kt < KT; ++kt) ktile(kt, std::false_type{});
    __syncthreads();                                              // all fragment reads done: LDS is free
    if (ABL == 3) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 123456.75f) reinterpret_cast<float*>(g.out)[0] = sum;
        return;
    }

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

// ------------------------------------------------------------------------------------------------------
// Large-tile kernel, register epilogue.
//
// Measured on MI355X (profiles/README.md, round 1): with 128x128 tiles the per-CU vector-memory path (64 B/clk)
// needs as many cycles to feed the LDS as the MFMAs need to consume it, and the LDS-staged epilogue adds a
// serial phase; DMA + MFMA + epilogue times simply add up.  This kernel cuts bytes per flop (256-row tiles) and
// removes the LDS round trip of the epilogue: the product is accumulated TRANSPOSED (weights are the MFMA "A"
// operand, tokens the "B" operand), so a lane ends up with 4 consecutive output features of ONE token per
// accumulator quad -> one 8-byte (bf16) / 16-byte (fp32) global access per quad, bias / ReLU / residual applied in
// registers, no barrier.
// ------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_t_kernel(GemmArgs g) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA pieces must divide evenly over the waves");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.a);
    const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(g.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5;

    const int nt_n = g.N_pad / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;

    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    constexpr int PER_STAGE = A_INSTR + B_INSTR;
    const __bf16* a_src[A_INSTR];
    const __bf16* b_src[B_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = m0 + row;
        grow = grow < g.M ? grow : g.M - 1;
        a_src[j] = A + (size_t)grow * g.lda + c * 8;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int row = (wave * B_INSTR + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        b_src[j] = W + (size_t)(n0 + row) * g.K + c * 8;
    }
    // one 1-KiB DMA piece (p < A_INSTR: activation rows, else weight rows) of K-step k0 into ring slot `stage`
    auto issue_piece = [&](int p, int stage, int k0) {
        unsigned char* sa = lds + stage * STAGE_BYTES;
        if (p < A_INSTR) lds_dma16(a_src[p] + k0, sa + (wave * A_INSTR + p) * 1024);
        else lds_dma16(b_src[p - A_INSTR] + k0, sa + BM * 128 + (wave * B_INSTR + (p - A_INSTR)) * 1024);
    };
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int p = 0; p < PER_STAGE; ++p) issue_piece(p, stage, k0);
    };

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

    f32x16 acc[TM][TN];                        // acc[i][j][r]: token m = i*32 + (lane&31), feature n = j*32 + 8*(r>>2) + 4*h + (r&3)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = g.K / G_BK;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) issue(s, s * G_BK);
    int stage = 0;
    auto ktile = [&](int kt, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const int younger = (KT - 1 - kt) < (STAGES - 2) ? (KT - 1 - kt) : (STAGES - 2);
        if (younger >= 2) wait_vmcnt<2 * PER_STAGE>();
        else if (younger == 1) wait_vmcnt<PER_STAGE>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int ns = stage + STAGES - 1;
        ns = ns >= STAGES ? ns - STAGES : ns;
        const int k0n = (kt + STAGES - 1) * G_BK;
        const unsigned char* st = lds + stage * STAGE_BYTES;
        bf16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[buf][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[buf][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
        };
        constexpr int NMFMA = 4 * TM * TN;
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[ks & 1][j], af[ks & 1][i], acc[i][j], 0, 0, 0);
                    const int idx = (ks * TM + i) * TN + j;
                    const int lo = idx * PER_STAGE / NMFMA, hi = (idx + 1) * PER_STAGE / NMFMA;
                    if (hi > lo) {
                        if (more) {
#pragma unroll
                            for (int p = lo; p < hi; ++p) issue_piece(p, ns, k0n);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    };
    int kt = 0;
    for (; kt + STAGES - 1 < KT; ++kt) ktile(kt, std::true_type{});
    for (; kt < KT; ++kt) ktile(kt, std::false_type{});

    // ---- register epilogue ----
    const bool vec = (g.N == g.N_pad) && ((g.ldc & 3) == 0) && (g.add == nullptr || (g.ld_add & 3) == 0) &&
                     (g.add2 == nullptr || (g.ld_add2 & 3) == 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gcol = n0 + wn * (TN * 32) + j * 32 + 8 * q + 4 * h;
            float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bias = *reinterpret_cast<const float4*>(g.bias + gcol);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int grow = m0 + wm * (TM * 32) + i * 32 + (lane & 31);
                if (grow >= g.M) continue;
                float4 v = make_float4(acc[i][j][4 * q + 0] + bias.x, acc[i][j][4 * q + 1] + bias.y,
                                       acc[i][j][4 * q + 2] + bias.z, acc[i][j][4 * q + 3] + bias.w);
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
    // algorithmic cost: 2*M*N*K flops; bytes = operands once + output once (+ addends)
    const double osz = g.out_dtype == BG_BF16 ? 2.0 : 4.0;
    const double bytes = 2.0 * g.M * g.K + 2.0 * g.N * (double)g.K + osz * g.M * g.N +
                         (g.add ? 4.0 * (g.M / g.add_div) * g.N : 0.0) + (g.add2 ? 4.0 * (g.M / g.add2_div) * g.N : 0.0);
    ProfScope prof(g.N_pad % 128 == 0 ? PK_GEMM_BF16_128 : PK_GEMM_BF16_64, 2.0 * g.M * g.N * (double)g.K, bytes, s);
    if (g.N_pad % 128 != 0) {
        hipLaunchKernelGGL((gemm_bf16_kernel<128, 64, 4, 1, 2>), dim3(((g.M + 127) / 128) * (g.N_pad / 64)), dim3(256), 0, s, g);
        return launch_status("gemm_bf16");
    }
    const int m128 = (g.M + 127) / 128, m256 = (g.M + 255) / 256, n128 = g.N_pad / 128;
    switch (g_tune[TUNE_GEMM_VARIANT]) {
        case 1:
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 3>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        case 2:
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 3>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 3:
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 2>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 4:
            if (g.N_pad % 256 == 0) {
                hipLaunchKernelGGL((gemm_bf16_kernel<128, 256, 2, 4, 3>), dim3(m128 * (g.N_pad / 256)), dim3(512), 0, s, g);
                break;
            }
            [[fallthrough]];
        case 5:
            if (g.N_pad % 256 == 0) {
                hipLaunchKernelGGL((gemm_bf16_t_kernel<256, 256, 2, 4, 2>), dim3(m256 * (g.N_pad / 256)), dim3(512), 0, s, g);
                break;
            }
            [[fallthrough]];
        case 6:
            if (g.N_pad % 192 == 0) {
                hipLaunchKernelGGL((gemm_bf16_t_kernel<256, 192, 4, 2, 2>), dim3(m256 * (g.N_pad / 192)), dim3(512), 0, s, g);
                break;
            }
            [[fallthrough]];
        case 7:
            hipLaunchKernelGGL((gemm_bf16_t_kernel<256, 128, 4, 2, 2>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 8:
            hipLaunchKernelGGL((gemm_bf16_t_kernel<128, 128, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        case 9:
            hipLaunchKernelGGL((gemm_bf16_t_kernel<256, 128, 4, 2, 3>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 11:
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2, 1>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        case 12:
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        case 13:
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2, 3>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        case 21:
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 3, 1>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 22:
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 3, 2>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 23:
            hipLaunchKernelGGL((gemm_bf16_kernel<256, 128, 4, 2, 3, 3>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        default:
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
    }
    return launch_status("gemm_bf16");
}

}  // namespace bg
