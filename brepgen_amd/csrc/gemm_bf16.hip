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

// ------------------------------------------------------------------------------------------------------
// Loader / consumer kernel ("LC").
//
// Why: in the kernels above every wave both issues LDS-DMA and MFMAs.  Instruction issue is in order per wave, and
// the per-CU vector-memory path (~64 B/clk) is close to saturated by the DMA stream, so a wave that wants to
// issue its next DMA piece blocks -- and with it the MFMAs queued behind it.  The ablations (profiles/r01) show the
// result: DMA time + MFMA time add up instead of overlapping.  Here the roles are split by wave:
//   waves 0..3  consumers: ds_read_b128 fragments + MFMA only, one per SIMD, wave tile 128 x 64
//   waves 4..7  loaders:   LDS-DMA only, 12 one-KiB pieces per K-step each, always one K-step ahead (3-slot ring)
// One s_barrier per K-step joins all eight waves: on passing barrier kt the consumers know tile kt has landed
// (each loader waited on its own vmcnt before arriving) and the loaders know ring slot (kt-1)%3 has been read.
// 256 x 128 tile: 48 KiB of DMA per K-step against 32 MFMAs (1024 cycles) per consumer -- the two streams are
// balanced instead of the 128x128 tile's 64 KiB per 1024 cycles per pair of co-resident blocks.
// ------------------------------------------------------------------------------------------------------
template <int STAGES, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_bf16_lc_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 128, TM = 4, TN = 2;             // consumer wave tile 128 x 64
    constexpr int STAGE_BYTES = (BM + BN) * 128;                  // 48 KiB
    constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    constexpr int A_PIECES = BM / 8 / 4, B_PIECES = BN / 8 / 4;   // per loader wave: 8 + 4
    constexpr int PER_STAGE = A_PIECES + B_PIECES;
    static_assert(4 * 128 * 64 * 4 <= LDS_BYTES, "epilogue patches must fit the ring");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt_n = g.N_pad / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;
    const int KT = g.K / G_BK;

    if (wave >= 4) {
        // ------------------------------ loader ------------------------------
        const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.a);
        const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(g.w);
        const int lw = wave - 4;
        const __bf16* a_src[A_PIECES];
        const __bf16* b_src[B_PIECES];
#pragma unroll
        for (int j = 0; j < A_PIECES; ++j) {
            const int row = (lw * A_PIECES + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int grow = m0 + row;
            grow = grow < g.M ? grow : g.M - 1;
            a_src[j] = A + (size_t)grow * g.lda + c * 8;
        }
#pragma unroll
        for (int j = 0; j < B_PIECES; ++j) {
            const int row = (lw * B_PIECES + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            b_src[j] = W + (size_t)(n0 + row) * g.K + c * 8;
        }
        auto issue = [&](int slot, int k0) {
            unsigned char* sa = lds + slot * STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < A_PIECES; ++j) lds_dma16(a_src[j] + k0, sa + (lw * A_PIECES + j) * 1024);
#pragma unroll
            for (int j = 0; j < B_PIECES; ++j) lds_dma16(b_src[j] + k0, sa + BM * 128 + (lw * B_PIECES + j) * 1024);
        };
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < KT) issue(s, s * G_BK);
        int slot = STAGES - 1;                                    // ring slot of tile kt + STAGES - 1
        for (int kt = 0; kt < KT; ++kt) {
            const int younger = (KT - 1 - kt) < (STAGES - 2) ? (KT - 1 - kt) : (STAGES - 2);
            if (younger >= 2) wait_vmcnt<2 * PER_STAGE>();
            else if (younger == 1) wait_vmcnt<PER_STAGE>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                         // barrier kt
            if (ABL != 2 && kt + STAGES - 1 < KT) issue(slot, (kt + STAGES - 1) * G_BK);
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }
        __builtin_amdgcn_s_barrier();                             // consumers are done with the ring
        return;
    }

    // ------------------------------ consumer ------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
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
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int stage = 0;
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my reads of tile kt-1 are complete
        __builtin_amdgcn_s_barrier();                             // barrier kt: tile kt landed
        const unsigned char* st = lds + stage * STAGE_BYTES;
        if (ABL != 1) {
        bf16x8 af[2][TM], bf[2][TN];
        auto load_frags = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[buf][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[buf][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
        };
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // every consumer finished reading the ring

    // ---- epilogue: wave-private LDS patch (128 x 64 fp32 = 32 KiB per consumer) -> coalesced rows ----
    constexpr int PW = TN * 32;
    float* patch = reinterpret_cast<float*>(lds) + wave * (TM * 32 * PW);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                patch[pr * PW + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // patch is wave-private: wave-level ordering only
    __builtin_amdgcn_wave_barrier();

    constexpr int LPR = PW / 4, RPI = 64 / LPR;
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
// Persistent kernel ("P"): the shipped configuration for N % 128 == 0.
//
// Measured (profiles/r01/gemm_ablation.log, gemm_lc_abl2.log): at K = 768 a 128x128 tile spends ~12 x 1 us in
// its K loop and another ~5 us in epilogue + wave drain + workgroup launch + first-DMA latency of its successor.
// Here 2 workgroups per CU stay resident and walk the tile list; the LDS-DMA stream never stops at a tile
// boundary (the first K-step of the NEXT tile is issued during the last K-step of the current one), the epilogue
// runs out of a small wave-private LDS patch that does not alias the ring, and nothing is re-launched.
//   LDS: ring 2 x 32 KiB + 4 x 4 KiB patches = 80 KiB  ->  2 workgroups per CU.
//   bf16 output: neighbouring lanes swap one accumulator so each lane owns a bf16 pair, the patch holds a
//   32 x 64 bf16 slab (128-byte rows) -> 16-byte-per-lane, full-line global stores.
//   fp32 output / residual: 32 x 32 fp32 slab per MFMA tile -> 128-byte row segments, residual added in flight.
// ------------------------------------------------------------------------------------------------------
template <bool INSTR>
__global__ __launch_bounds__(256, 2) void gemm_bf16_p_kernel(GemmArgs g, int n_tiles, unsigned long long* dbg, int dephase) {
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    constexpr int STAGE_BYTES = (BM + BN) * 128;                  // 32 KiB
    constexpr int RING = 2 * STAGE_BYTES;
    constexpr int A_INSTR = 4, B_INSTR = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING + 4 * 4096];

    const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.a);
    const __bf16* __restrict__ W = reinterpret_cast<const __bf16*>(g.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
    const int nt_n = g.N_pad / BN;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);                   // tiles first, first+G, ... : co-running workgroups
    if (first >= n_tiles) return;                                 // of one XCD take consecutive tiles (shared A panel)

    // per-lane DMA source rows / swizzled chunks (tile independent part)
    int a_row[A_INSTR], a_chunk[A_INSTR], b_row[B_INSTR], b_chunk[B_INSTR];
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        a_row[j] = (wave * A_INSTR + j) * 8 + (lane >> 3);
        a_chunk[j] = ((lane & 7) ^ ((a_row[j] >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        b_row[j] = (wave * B_INSTR + j) * 8 + (lane >> 3);
        b_chunk[j] = ((lane & 7) ^ ((b_row[j] >> 1) & 7)) * 8;
    }
    const __bf16* a_src[A_INSTR];
    const __bf16* b_src[B_INSTR];
    auto set_tile = [&](int L, int& m0, int& n0) {
        m0 = (L / nt_n) * BM;
        n0 = (L % nt_n) * BN;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            int grow = m0 + a_row[j];
            grow = grow < g.M ? grow : g.M - 1;
            a_src[j] = A + (size_t)grow * g.lda + a_chunk[j];
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) b_src[j] = W + (size_t)(n0 + b_row[j]) * g.K + b_chunk[j];
    };
    auto issue = [&](int slot, int k0) {
        unsigned char* sa = lds + slot * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) lds_dma16(a_src[j] + k0, sa + (wave * A_INSTR + j) * 1024);
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) lds_dma16(b_src[j] + k0, sa + BM * 128 + (wave * B_INSTR + j) * 1024);
    };

    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * 64 + i * 32 + (lane & 31);
        a_off[i] = row * 128;
        a_sw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int row = wn * 64 + i * 32 + (lane & 31);
        b_off[i] = BM * 128 + row * 128;
        b_sw[i] = (row >> 1) & 7;
    }

    unsigned* patch = reinterpret_cast<unsigned*>(lds + RING + wave * 4096);
    const bool bf16_fast = g.out_dtype == BG_BF16 && g.add == nullptr && g.add2 == nullptr;
    const int KT = g.K / G_BK;

    if (dephase > 0 && (int)blockIdx.x >= G / 2) {               // experiment: offset the second resident workgroup
        for (int i = 0; i < dephase; ++i) __builtin_amdgcn_s_sleep(16);   // 16 * 64 clocks per iteration
    }
    int m0, n0;
    unsigned long long t_wait = 0, t_comp = 0, t_epi = 0, t_begin = 0, n_done = 0;
    if (INSTR) t_begin = __builtin_amdgcn_s_memtime();
    set_tile(first, m0, n0);
    issue(0, 0);
    int slot = 0;
    for (int L = first; L < n_tiles; L += G) {
        const bool has_next = L + G < n_tiles;
        const int cm0 = m0, cn0 = n0;                             // coordinates of the tile being computed
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int kt = 0; kt < KT; ++kt) {
            unsigned long long ta = 0;
            if (INSTR) ta = __builtin_amdgcn_s_memtime();
            wait_vmcnt<0>();                                      // my pieces of this K-step (and older stores) done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            unsigned long long tb = 0;
            if (INSTR) { tb = __builtin_amdgcn_s_memtime(); t_wait += tb - ta; }
            if (kt + 1 < KT) {
                issue(slot ^ 1, (kt + 1) * G_BK);
            } else if (has_next) {                                // keep the DMA stream running across the tile seam
                set_tile(L + G, m0, n0);
                issue(slot ^ 1, 0);
            }
            const unsigned char* st = lds + slot * STAGE_BYTES;
            bf16x8 af[2][TM], bf[2][TN];
            auto load_frags = [&](int ks, int buf) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[buf][i] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[buf][j] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
            };
            load_frags(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot ^= 1;
            if (INSTR) {
                asm volatile("s_nop 0" ::: "memory");
                t_comp += __builtin_amdgcn_s_memtime() - tb;
            }
        }
        unsigned long long te = 0;
        if (INSTR) te = __builtin_amdgcn_s_memtime();

        // ---------------- epilogue (wave-private patch; the ring already receives the next tile) ----------------
        const int rbase = cm0 + wm * 64, cbase = cn0 + wn * 64;
        float bias_l[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_l[j] = g.bias ? g.bias[cbase + j * 32 + (lane & 31)] : 0.f;
        if (bf16_fast) {
            __bf16* out = reinterpret_cast<__bf16*>(g.out);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        float a = acc[i][j][2 * rp] + bias_l[j], b = acc[i][j][2 * rp + 1] + bias_l[j];
                        if (g.act == BG_ACT_RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                        // even lane keeps row r (cols c, c+1), odd lane row r+1 (cols c-1, c)
                        const float send = (lane & 1) ? a : b;
                        // neighbour exchange inside lane pairs: DPP quad_perm [1,0,3,2] (pure VALU, no LDS crossbar)
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                        const float lo = (lane & 1) ? recv : a, hi = (lane & 1) ? b : recv;
                        const int r = 2 * rp + (lane & 1);
                        const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                        union { __bf16 v[2]; unsigned u; } pk;
                        pk.v[0] = (__bf16)lo; pk.v[1] = (__bf16)hi;
                        patch[prow * 32 + j * 16 + ((lane & 31) >> 1)] = pk.u;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int prow = it * 8 + (lane >> 3), chunk = lane & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(&patch[prow * 32 + chunk * 4]);
                    const int grow = rbase + i * 32 + prow;
                    if (grow < g.M)
                        *reinterpret_cast<uint4*>(out + (size_t)grow * g.ldc + cbase + chunk * 8) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            float* pf = reinterpret_cast<float*>(patch);
            // residual / broadcast addend: all 16 row-segment loads of this wave go out FIRST (64 VGPRs), so their
            // HBM/MALL latency overlaps the patch traffic instead of forming 16 serial round trips
            float4 res[2][4];                                     // two MFMA tiles of residual rows in flight
            auto load_res = [&](int t, int buf) {
                const int i = t >> 1, j = t & 1;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    int grow = rbase + i * 32 + it * 8 + (lane >> 3);
                    grow = grow < g.M ? grow : g.M - 1;
                    res[buf][it] = *reinterpret_cast<const float4*>(
                        g.add + (size_t)(grow / g.add_div) * g.ld_add + cbase + j * 32 + (lane & 7) * 4);
                }
            };
            if (g.add) { load_res(0, 0); load_res(1, 1); }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + bias_l[j];
                        if (g.act == BG_ACT_RELU) v = fmaxf(v, 0.f);
                        pf[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (lane & 31)] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int prow = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                        float4 v = *reinterpret_cast<const float4*>(&pf[prow * 32 + c4]);
                        const int grow = rbase + i * 32 + prow, gcol = cbase + j * 32 + c4;
                        if (g.add) {
                            const float4 a4 = res[(i * TN + j) & 1][it];
                            v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                        }
                        if (grow < g.M) {
                            if (g.add2) {
                                const float4 a4 = *reinterpret_cast<const float4*>(g.add2 + (size_t)(grow / g.add2_div) * g.ld_add2 + gcol);
                                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                            }
                            if (g.out_dtype == BG_BF16)
                                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(g.out) + (size_t)grow * g.ldc + gcol) =
                                    to_bf16x4(v.x, v.y, v.z, v.w);
                            else
                                *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (size_t)grow * g.ldc + gcol) = v;
                        }
                    }
                    if (g.add && i * TN + j + 2 < TM * TN) load_res(i * TN + j + 2, (i * TN + j) & 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
        }
        if (INSTR) { t_epi += __builtin_amdgcn_s_memtime() - te; ++n_done; }
    }
    if (INSTR && dbg != nullptr && lane == 0) {
        unsigned long long* o = dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = t_wait; o[1] = t_comp; o[2] = t_epi; o[3] = __builtin_amdgcn_s_memtime() - t_begin; o[4] = n_done;
        o[5] = t_begin;
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
    const int variant = g_tune[TUNE_GEMM_VARIANT] == 0 ? 30 : g_tune[TUNE_GEMM_VARIANT];   // 30 = shipped: persistent kernel
    switch (variant) {
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
        case 20:
            hipLaunchKernelGGL((gemm_bf16_lc_kernel<3>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 31:
        case 30: {
            const bool ok = (g.ldc % 8 == 0) && (g.N == g.N_pad) && (g.add == nullptr || g.ld_add % 4 == 0) &&
                            (g.add2 == nullptr || g.ld_add2 % 4 == 0);
            if (ok) {
                const int nt = m128 * n128;
                const int grid = nt < 512 ? nt : 512;             // 2 resident workgroups per CU x 256 CUs
                if (g_tune[TUNE_GEMM_VARIANT] == 31) {
                    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(((unsigned long long)(unsigned)g_tune[2] << 32) | (unsigned)g_tune[1]);
                    hipLaunchKernelGGL(gemm_bf16_p_kernel<true>, dim3(grid), dim3(256), 0, s, g, nt, dbg, g_tune[3]);
                } else {
                    hipLaunchKernelGGL(gemm_bf16_p_kernel<false>, dim3(grid), dim3(256), 0, s, g, nt, (unsigned long long*)nullptr, g_tune[3]);
                }
                break;
            }
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
        }
        case 14:
            hipLaunchKernelGGL((gemm_bf16_lc_kernel<3, 1>), dim3(m256 * n128), dim3(512), 0, s, g);
            break;
        case 15:
            hipLaunchKernelGGL((gemm_bf16_lc_kernel<3, 2>), dim3(m256 * n128), dim3(512), 0, s, g);
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
        default:                                                  // 10: non-persistent 128x128 / 2-stage
            hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g);
            break;
    }
    return launch_status("gemm_bf16");
}

}  // namespace bg
