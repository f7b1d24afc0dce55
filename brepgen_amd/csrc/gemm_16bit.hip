// 16-bit (bf16 / fp16) MFMA GEMM with fused epilogues -- the QKV / out-proj / FFN1 / FFN2 / embed / fc_out Linears
// of the denoisers (network.py:1076-1099) and every convolution of the VAE decoders (as im2col GEMMs), i.e. >95 % of
// the path's FLOPs.  MFMA-bound.
//
//   out[m,n] = act(sum_k a[m,k] * w[n,k] + bias[n]) (+ add[(m / add_div), n] (+ add2[...]))   a, w 16-bit; fp32 accumulate
// plus, for the 16-bit modes of the denoisers:
//   * LayerNorm fold (consumer): a = raw residual rows, w = T(gamma * W); the epilogue applies per-row (rstd, -mean*rstd)
//     summed from per-64-column (sum, sum of squares) partials a producer GEMM left behind;
//   * split residual (producer): the fp32 result + residual is stored as two 16-bit planes hi = T(v), lo = T(v - hi),
//     hi being the next GEMM's A operand, together with those partials.
//
// Design (gfx950):
//   * tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), v_mfma_f32_32x32x16_{bf16,f16}: 16 fp32 accumulators per
//     32x32 MFMA tile, 2 x 2 MFMA tiles per wave;
//   * both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]) so A and B fragments share one
//     16-byte-per-lane shape: lane l holds rows (l & 31), k-chunk (l >> 5) of each 16-wide k-slice;
//   * L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows x 128 B) into a ring over K,
//     counted vmcnt + raw s_barrier, ONE barrier per 64-wide K-step;
//   * the LDS image is XOR-swizzled at 16-byte granularity, chunk' = chunk ^ ((row >> 1) & 7).  The DMA writes
//     lane-linear, so the permutation is applied to the per-lane SOURCE address and again on the ds_read_b128 side
//     (same involution): every 16-lane service group of ds_read_b128 touches 16 distinct 16-byte slots of the
//     256-byte bank row -- measured SQ_LDS_BANK_CONFLICT = 0;
//   * fragment reads are software-pipelined one k-slice ahead of the MFMAs (order pinned with sched_barrier);
//   * shipped kernel = the PERSISTENT one (gemm16_persistent_kernel): 2 workgroups per CU stay resident and walk
//     the tile list, the DMA stream runs across tile seams, the epilogue works out of a small wave-private LDS patch
//     that does not alias the ring, everything it needs from memory (bias / column sums, residual rows, row statistics)
//     is requested at the start of the tile's last K-step, full-line global stores;
//   * XCD-aware tile order: XCD x walks row panels x, x + 8, ... column-fastest (shared A row panels per L2).
// The exploration behind this shape (3-stage rings, 256-row tiles, register epilogue, loader/consumer wave
// specialisation, ablations) is summarised in DESIGN.md section 4 with the logs under profiles/r01/; those kernel
// variants live in the git history (commit "Persistent 128x128 GEMM ...").
#include "gemm16.h"

namespace bg {

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
    // dst = wave-uniform base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// BM x BN block tile, WM x WN waves (each wave (BM/WM) x (BN/WN)), STAGES-deep LDS ring over K.
// One barrier per K-step, all waves in lock-step.  (The 8-phase K loop that was tried here as an experimental variant in round 2
// lives in gemm_p256.hip now, as the K loop of the 256 x 256 persistent kernel.)
template <bool F16, int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64) void gemm16_kernel(GemmArgs g) {
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;           // 32x32 MFMA tiles per wave
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    // epilogue patch: a wave stages its rows 64 at a time (EH = MFMA row tiles per pass), so that the 256 x 256 tile's
    // patches (8 waves x 16 KiB) fit where the ring was
    constexpr int EH = TM > 2 ? 2 : TM;
    constexpr int EPI_BYTES = NW * EH * 32 * (BN / WN) * 4;
    constexpr int LDS_BYTES = (STAGES * STAGE_BYTES > EPI_BYTES) ? STAGES * STAGE_BYTES : EPI_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA pieces must divide evenly over the waves");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const T* __restrict__ A = reinterpret_cast<const T*>(g.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(g.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nt_n = g.N_pad / BN;
    const int nblk = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nblk);
    const int m0 = (tile / nt_n) * BM, n0 = (tile % nt_n) * BN;
    const int Mv = g.m_dev ? *g.m_dev : g.M;                      // rows present (compacted batch: device-side count)
    if (m0 >= Mv) return;                                         // uniform per workgroup, before any barrier

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

    {
        // ---- LDS-DMA source addresses (per lane), destination bases (per wave) ----
        constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;   // wave-instructions per wave per stage
        constexpr int PER_STAGE = A_INSTR + B_INSTR;
        const T* a_src[A_INSTR];
        const T* b_src[B_INSTR];
    #pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            const int row = (wave * A_INSTR + j) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int grow = m0 + row;
            grow = grow < Mv ? grow : Mv - 1;                         // clamp: rows >= M are never stored
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
            V8 af[2][TM], bf[2][TN];
            auto load_frags = [&](int ks, int buf) {
    #pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[buf][i] = *reinterpret_cast<const V8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
    #pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[buf][j] = *reinterpret_cast<const V8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
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
                        acc[i][j] = E::mfma(af[ks & 1][i], bf[ks & 1][j], acc[i][j]);
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
        for (; kt + STAGES - 1 < KT; ++kt) ktile(kt, std::true_type{});
        for (; kt < KT; ++kt) ktile(kt, std::false_type{});
    }
    __syncthreads();                                              // all fragment reads done: LDS is free

    // ---- epilogue: accumulators -> wave-private LDS patch -> coalesced rows, EH x 32 rows per pass ----
    constexpr int PW = TN * 32;                                   // patch width (floats)
    float* patch = reinterpret_cast<float*>(lds) + wave * (EH * 32 * PW);
    constexpr int LPR = PW / 4;                                   // lanes per patch row (float4 each)
    constexpr int RPI = 64 / LPR;                                 // rows per iteration
    static_assert(PW == 64, "the row-statistics / LayerNorm-fold code assumes one 64-column group per wave");
    const int rr = lane / LPR, c4 = (lane % LPR) * 4;
    const int gcol = n0 + wn * PW + c4;
    const bool vec = (g.N == g.N_pad) && ((g.ldc & 3) == 0) && (g.add == nullptr || (g.ld_add & 3) == 0) &&
                     (g.add2 == nullptr || (g.ld_add2 & 3) == 0);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), csum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias) bias = *reinterpret_cast<const float4*>(g.bias + gcol);
    if (g.stats_in) csum = *reinterpret_cast<const float4*>(g.colsum + gcol);
    const int nparts_in = g.K / G_BK;
#pragma unroll
    for (int eh = 0; eh < TM / EH; ++eh) {
    if (eh > 0) {                                                 // the patch is wave-private: a wave-level fence is enough
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < EH; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pr = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;      // C/D layout of the 32x32 MFMA
                patch[pr * PW + j * 32 + (lane & 31)] = acc[eh * EH + i][j][r];
            }
    if (TM == EH) {                                               // one pass (every shipped instantiation)
        __syncthreads();
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll 4
    for (int it = 0; it < EH * 32 / RPI; ++it) {
        const int pr = it * RPI + rr;
        const int grow = m0 + wm * (TM * 32) + eh * (EH * 32) + pr;
        const bool row_ok = grow < Mv;
        const int crow = row_ok ? grow : Mv - 1;
        const int prow = g.row_map ? g.row_map[crow] : crow;      // index in the padded token layout (compacted batches)
        const int arow1 = g.map_add ? prow : crow, arow2 = g.map_add2 ? prow : crow, orow = g.map_out ? prow : grow;
        float4 v = *reinterpret_cast<const float4*>(&patch[pr * PW + c4]);
        if (g.stats_in) {                                         // LayerNorm fold: the 16 lanes of a row share its statistics
            float2 pq = make_float2(0.f, 0.f);
            if (lane % LPR < nparts_in)
                pq = *reinterpret_cast<const float2*>(g.stats_in + ((size_t)(lane % LPR) * g.M + crow) * 2);
            const float2 cf = ln_fold_coeffs(group16_sum(pq.x), group16_sum(pq.y), g.K, g.ln_eps);
            v.x = ln_fold_apply(v.x, cf.x, cf.y, csum.x, bias.x);
            v.y = ln_fold_apply(v.y, cf.x, cf.y, csum.y, bias.y);
            v.z = ln_fold_apply(v.z, cf.x, cf.y, csum.z, bias.z);
            v.w = ln_fold_apply(v.w, cf.x, cf.y, csum.w, bias.w);
        } else {
            v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        }
        if (g.act == BG_ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (vec) {
            if (g.add) {
                const float4 a4 = *reinterpret_cast<const float4*>(g.add + (size_t)(arow1 / g.add_div) * g.ld_add + gcol);
                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            if (g.res_hi) {                                       // split residual stream: x_old = hi + lo
                const size_t o = (size_t)crow * g.ld_res + gcol;
                float fh[4], fl[4];
                unpack4_16<F16>(*reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(g.res_hi) + o), fh);
                unpack4_16<F16>(*reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(g.res_lo) + o), fl);
                v.x += fh[0] + fl[0]; v.y += fh[1] + fl[1]; v.z += fh[2] + fl[2]; v.w += fh[3] + fl[3];
            }
            if (g.add2) {
                const float4 a4 = *reinterpret_cast<const float4*>(g.add2 + (size_t)(arow2 / g.add2_div) * g.ld_add2 + gcol);
                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
            }
            if (g.stats_out) {                                    // per-64-column (sum, sum of squares) of the fp32 result
                const float s4 = (v.x + v.y) + (v.z + v.w);
                const float q4 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                const float S = group16_sum(s4), Q = group16_sum(q4);
                if (row_ok && lane % LPR == 0)
                    *reinterpret_cast<float2*>(g.stats_out + ((size_t)((n0 + wn * PW) / 64) * g.M + grow) * 2) =
                        make_float2(S, Q);
            }
            if (!row_ok) continue;
            if (g.out_lo) {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                uint2 hi, lo;
                split4_16<F16>(vv, hi, lo);
                *reinterpret_cast<uint2*>(reinterpret_cast<T*>(g.out) + (size_t)orow * g.ldc + gcol) = hi;
                *reinterpret_cast<uint2*>(reinterpret_cast<T*>(g.out_lo) + (size_t)orow * g.ldc + gcol) = lo;
            } else if (g.out_dtype != BG_F32)
                *reinterpret_cast<V4*>(reinterpret_cast<T*>(g.out) + (size_t)orow * g.ldc + gcol) =
                    E::pack4(v.x, v.y, v.z, v.w);
            else
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (size_t)orow * g.ldc + gcol) = v;
        } else {
            if (!row_ok) continue;
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = gcol + e;
                if (col >= g.N) continue;
                float o = vv[e];
                if (g.add) o += g.add[(size_t)(arow1 / g.add_div) * g.ld_add + col];
                if (g.add2) o += g.add2[(size_t)(arow2 / g.add2_div) * g.ld_add2 + col];
                asm volatile("" : "+v"(o));                               // (no add + fp16-conversion fusion)
                if (g.out_dtype != BG_F32) reinterpret_cast<T*>(g.out)[(size_t)orow * g.ldc + col] = (T)o;
                else reinterpret_cast<float*>(g.out)[(size_t)orow * g.ldc + col] = o;
            }
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
//   16-bit output: neighbouring lanes swap one accumulator so each lane owns a bf16 pair, the patch holds a
//   32 x 64 bf16 slab (128-byte rows) -> 16-byte-per-lane, full-line global stores.
//   fp32 output / residual: 32 x 32 fp32 slab per MFMA tile -> 128-byte row segments, residual added in flight.
//   split output: 16 x 64 fp32 slab -> a lane owns 8 columns of a row: hi / lo / residual as 16-byte accesses.
// ------------------------------------------------------------------------------------------------------
// MODE selects the epilogue (one instantiation each, so that no instantiation carries the registers of another):
//   P_PLAIN16  16-bit output, bias (+ReLU)                              -- QKV / FFN1 without the LayerNorm fold, VAE convs
//   P_FOLD16   same with the LayerNorm fold (stats_in / colsum)         -- QKV / FFN1 of the denoisers
//   P_GENERAL  fp32 or 16-bit output with fp32 addends (add / add2)     -- fp32 residual stream, embeds, VAE residuals
//   P_SPLIT    split (hi, lo) output, addend = split residual or fp32 broadcast rows, optional row statistics
//                                                                       -- token embeds of the denoisers (row maps, broadcast addends);
//                                                                          out-proj / FFN2 only as the tests' baseline: the product
//                                                                          path runs them on gemm_split.hip / gemm_p256.hip

// CONV: the A operand is gathered from a conv window (implicit GEMM).  PURE (with CONV): the convolution as the VAE passes issue it --
// fp32 output, fp32 residual of the same shape or none, no activation, no second addend -- with the epilogue's run-time option checks
// folded away (same arithmetic, 904 -> 242 VALU instructions per tile epilogue).
template <bool F16, int MODE, bool CONV = false, bool PURE = false>
__global__ __launch_bounds__(256, 2) void gemm16_persistent_kernel(GemmArgs g, int m_panels) {
    constexpr bool FAST = MODE == P_PLAIN16 || MODE == P_FOLD16, FOLD = MODE == P_FOLD16, SPLIT = MODE == P_SPLIT;
    static_assert(!PURE || CONV, "PURE specialises the convolution epilogue (fp32 out, no activation, add_div 1, no add2)");
    const bool k_relu = PURE ? false : g.act == BG_ACT_RELU, k_res_split = g.res_hi != nullptr;
    const bool k_add2 = PURE ? false : g.add2 != nullptr, k_stats = g.stats_out != nullptr;
    const bool k_map = PURE ? false : g.row_map != nullptr;
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    constexpr int STAGE_BYTES = (BM + BN) * 128;                  // 32 KiB
    constexpr int RING = 2 * STAGE_BYTES;
    constexpr int A_INSTR = 4, B_INSTR = 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING + 4 * 4096];

    const T* __restrict__ A = reinterpret_cast<const T*>(g.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(g.w);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5;
    const int nt_n = g.N_pad / BN;
    const int G = gridDim.x;
    const int Mv = g.m_dev ? *g.m_dev : g.M;                      // rows present (compacted batch: device-side count)
    if (g.m_dev) m_panels = (Mv + BM - 1) / BM;
    const int p0 = g.hybrid ? (g.rows256_dev ? *g.rows256_dev : g.rows256_host) >> 7 : 0;      // hybrid launches: the 256 x 256 kernel owns the panels below p0
    // XCD-aware tile walk (block b runs on XCD b % 8; each XCD has a private 4 MiB L2): XCD x owns the row panels x, x + 8, ...;
    // its G / 8 workgroups walk that sub-grid column-fastest, so the ~64 concurrently running tiles of an XCD share a few A row
    // panels and keep W resident.  (Column groups per XCD, a row-major walk and adjacent-column pairing of the two workgroups of a
    // CU were measured in rounds 1-2 and are flat or slower: DESIGN.md section 4.)
    const int xcd = blockIdx.x & 7, w_local = blockIdx.x >> 3, cnt = G >> 3;      // G % 8 == 0 (launcher)
    auto tile_at = [&](int t, int& tm0, int& tn0) -> bool {
        const int panel = p0 + xcd + (t / nt_n) * 8;
        tm0 = panel * BM;
        tn0 = (t % nt_n) * BN;
        return panel < m_panels;
    };

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
    const T* a_src[A_INSTR];
    const T* b_src[B_INSTR];
    // implicit-GEMM convolution: output pixel (sample base, oy, ox) of each DMA row; a_src then points at channel 0 of the
    // CURRENT tap's source pixel (or at the zero page for a padded tap) and is re-derived whenever the K loop crosses a tap
    int cv_base[CONV ? A_INSTR : 1], cv_oy[CONV ? A_INSTR : 1], cv_ox[CONV ? A_INSTR : 1];
    auto set_tap = [&](int tap) {
        const int ky = tap / g.cv_kw, kx = tap - ky * g.cv_kw;
#pragma unroll
        for (int j = 0; j < (CONV ? A_INSTR : 0); ++j) {
            const int iy = cv_oy[j] + ky - (g.cv_kh >> 1), ix = cv_ox[j] + kx - (g.cv_kw >> 1);
            const bool ok = (unsigned)iy < (unsigned)(g.cv_H << g.cv_up) && (unsigned)ix < (unsigned)(g.cv_W << g.cv_up);
            const T* pix = A + ((size_t)(cv_base[j] + (iy >> g.cv_up) * g.cv_W + (ix >> g.cv_up))) * g.cv_C;
            a_src[j] = (ok ? pix : reinterpret_cast<const T*>(g.cv_zero)) + a_chunk[j];
        }
    };
    auto set_src = [&](int m0, int n0) {
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            int grow = m0 + a_row[j];
            grow = grow < Mv ? grow : Mv - 1;
            if (CONV) {
                cv_ox[CONV ? j : 0] = grow & ((1 << g.cv_wo_log2) - 1);
                const int t = grow >> g.cv_wo_log2;
                cv_oy[CONV ? j : 0] = t & ((1 << g.cv_ho_log2) - 1);
                cv_base[CONV ? j : 0] = (t >> g.cv_ho_log2) * g.cv_H * g.cv_W;
            } else {
                a_src[j] = A + (size_t)grow * g.lda + a_chunk[j];
            }
        }
        if (CONV) set_tap(0);
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) b_src[j] = W + (size_t)(n0 + b_row[j]) * g.K + b_chunk[j];
    };
    auto issue = [&](int slot, int k0) {
        unsigned char* sa = lds + slot * STAGE_BYTES;
        int ka = k0;                                              // A: offset inside the row / inside the tap's C channels
        if (CONV) {
            const int kt_ = k0 / G_BK, spt_mask = (1 << g.cv_spt_log2) - 1;
            if (k0 != 0 && (kt_ & spt_mask) == 0) set_tap(kt_ >> g.cv_spt_log2);   // (tap 0 was set with the tile)
            ka = (kt_ & spt_mask) * G_BK;
        }
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) lds_dma16(a_src[j] + ka, sa + (wave * A_INSTR + j) * 1024);
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
    constexpr bool half_fast = FAST;
    const bool has_res = !FAST && (g.add != nullptr || (SPLIT && g.res_hi != nullptr));   // prefetched addend rows
    const int KT = g.K / G_BK;

    int m0, n0;
    const int t_first = w_local;
    if (!tile_at(t_first, m0, n0)) return;
    set_src(m0, n0);
    issue(0, 0);
    int slot = 0;
    for (int t = t_first;; t += cnt) {
        const int cm0 = m0, cn0 = n0;                             // coordinates of the tile being computed
        const bool has_next = tile_at(t + cnt, m0, n0);           // (m0, n0) now name the NEXT tile
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // residual / broadcast addend rows (fp32 epilogue): the first two MFMA tiles' worth (8 float4) are requested
        // at the START of the last K-step, so their HBM / MALL latency hides behind that step's MFMAs
        const int rbase = cm0 + wm * 64, cbase = cn0 + wn * 64;
        // (P_SPLIT works on 16-row x 64-column slabs, slab t = i * 2 + half, 8 columns per lane: res[buf][2 * it],
        //  res[buf][2 * it + 1] hold either the (hi, lo) octets of the split residual or 8 fp32 addend values)
        float4 res[2][4];
        // compacted batches (SPLIT only; the launcher sends every other mapped GEMM to the generic kernel): the broadcast
        // addends are indexed by the row's position in the PADDED token layout.  The 8 indices a lane needs (slab t,
        // half it -> row rbase + 16 t + 8 it + lane / 8) are requested at the START of the tile, a whole K loop ahead of
        // the residual prefetch that depends on them.
        int ridx[8];
        if (SPLIT && k_map) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int grow = rbase + k * 8 + (lane >> 3);
                grow = grow < Mv ? grow : Mv - 1;
                ridx[k] = g.row_map[grow];
            }
        }
        auto load_res = [&](int t, int buf) {
            if (SPLIT) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    int grow = rbase + t * 16 + it * 8 + (lane >> 3);
                    grow = grow < Mv ? grow : Mv - 1;
                    // two 16-byte loads either way (hi / lo octets, or 8 fp32 addends): select the ADDRESSES, so the
                    // loads themselves stay unconditional and in flight together
                    const size_t o = (size_t)grow * g.ld_res + cbase + (lane & 7) * 8;
                    const int arow = (k_map && g.map_add) ? ridx[t * 2 + it] : grow;
                    const float* ap = g.add + (size_t)(arow / g.add_div) * g.ld_add + cbase + (lane & 7) * 8;
                    const void* p0 = k_res_split ? (const void*)(reinterpret_cast<const T*>(g.res_hi) + o) : (const void*)ap;
                    const void* p1 = k_res_split ? (const void*)(reinterpret_cast<const T*>(g.res_lo) + o) : (const void*)(ap + 4);
                    res[buf][2 * it] = *reinterpret_cast<const float4*>(p0);
                    res[buf][2 * it + 1] = *reinterpret_cast<const float4*>(p1);
                }
                return;
            }
            const int i = t >> 1, j = t & 1;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int grow = rbase + i * 32 + it * 8 + (lane >> 3);
                grow = grow < Mv ? grow : Mv - 1;
                res[buf][it] = *reinterpret_cast<const float4*>(
                    g.add + (size_t)(PURE ? grow : grow / g.add_div) * g.ld_add + cbase + j * 32 + (lane & 7) * 4);
            }
        };
        // per-column epilogue vectors (bias; LayerNorm fold: column sums) are requested there too: loaded in the
        // epilogue itself they would expose one L2 round trip per tile
        float bias_l[TN] = {0.f, 0.f}, cs_l[TN] = {0.f, 0.f};
        auto load_cols = [&]() {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (g.bias) bias_l[j] = g.bias[cbase + j * 32 + (lane & 31)];
                if (FOLD) cs_l[j] = g.colsum[cbase + j * 32 + (lane & 31)];
            }
        };
        // LayerNorm fold (consumer): lane l owns the statistics of row rbase + l; its 12 partial pairs are requested
        // at the start of the last K-step as well
        float2 st[16];
        auto load_stats = [&]() {
            int grow = rbase + lane;
            grow = grow < Mv ? grow : Mv - 1;
            const float2* sp = reinterpret_cast<const float2*>(g.stats_in) + grow;     // part-major: [KT][M] pairs
#pragma unroll
            for (int p = 0; p < 16; ++p)                          // K == 768 (launcher): 12 unconditional loads in flight together
                st[p] = p < FOLD_PARTS ? sp[(size_t)p * g.M] : make_float2(0.f, 0.f);
        };
        auto kstep = [&](int kt, auto last_c) {
            constexpr bool last = decltype(last_c)::value;
            wait_vmcnt<0>();                                      // my pieces of this K-step (and older stores) done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!last) {
                issue(slot ^ 1, (kt + 1) * G_BK);
            } else {
                if (has_next) {                                   // keep the DMA stream running across the tile seam
                    set_src(m0, n0);
                    issue(slot ^ 1, 0);
                }
                load_cols();
                if (!FAST && has_res) { load_res(0, 0); load_res(1, 1); }
                if (FOLD) load_stats();
            }
            const unsigned char* st = lds + slot * STAGE_BYTES;
            V8 af[2][TM], bf[2][TN];
            auto load_frags = [&](int ks, int buf) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[buf][i] = *reinterpret_cast<const V8*>(st + a_off[i] + (((ks * 2 + h) ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[buf][j] = *reinterpret_cast<const V8*>(st + b_off[j] + (((ks * 2 + h) ^ b_sw[j]) << 4));
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
                        acc[i][j] = E::mfma(af[ks & 1][i], bf[ks & 1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot ^= 1;
        };
        for (int kt = 0; kt + 1 < KT; ++kt) kstep(kt, std::false_type{});
        kstep(KT - 1, std::true_type{});

        // ---------------- epilogue (wave-private patch; the ring already receives the next tile) ----------------
        if (half_fast) {
            T* out = reinterpret_cast<T*>(g.out);
            constexpr bool fold = FOLD;
            float2 cf = make_float2(1.f, 0.f);                    // (rstd, -mean * rstd) of row rbase + lane
            if (fold) {
                float ps[16], pq[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) { ps[p] = st[p].x; pq[p] = st[p].y; }
                cf = ln_fold_coeffs(tree16(ps), tree16(pq), g.K, g.ln_eps);
            }
            // every lane needs the coefficients of its 2 x 16 accumulator rows: 64 float2 go through the (not yet
            // used) patch, read back as broadcasts (all lanes of a half wave read the same address)
            float2 cAB[TM][16];
            if (fold) {
                float2* pc = reinterpret_cast<float2*>(patch);
                pc[lane] = cf;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cAB[i][r] = pc[i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int rp = 0; rp < 8; ++rp) {
                        float a, b;
                        if (fold) {
                            a = ln_fold_apply(acc[i][j][2 * rp], cAB[i][2 * rp].x, cAB[i][2 * rp].y, cs_l[j], bias_l[j]);
                            b = ln_fold_apply(acc[i][j][2 * rp + 1], cAB[i][2 * rp + 1].x, cAB[i][2 * rp + 1].y, cs_l[j], bias_l[j]);
                        } else {
                            a = acc[i][j][2 * rp] + bias_l[j]; b = acc[i][j][2 * rp + 1] + bias_l[j];
                        }
                        if (g.act == BG_ACT_RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                        // even lane keeps row r (cols c, c+1), odd lane row r+1 (cols c-1, c)
                        const float send = (lane & 1) ? a : b;
                        // neighbour exchange inside lane pairs: DPP quad_perm [1,0,3,2] (pure VALU, no LDS crossbar)
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                            0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                        float lo = (lane & 1) ? recv : a, hi = (lane & 1) ? b : recv;
                        asm volatile("" : "+v"(lo), "+v"(hi));       // (no fma + fp16-conversion fusion: gemm16.h Elem<true>::pack4)
                        const int r = 2 * rp + (lane & 1);
                        const int prow = (r & 3) + 8 * (r >> 2) + 4 * h;
                        union { T v[2]; unsigned u; } pk;
                        pk.v[0] = (T)lo; pk.v[1] = (T)hi;
                        patch[prow * 32 + j * 16 + ((lane & 31) >> 1)] = pk.u;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int prow = it * 8 + (lane >> 3), chunk = lane & 7;
                    const uint4 v = *reinterpret_cast<const uint4*>(&patch[prow * 32 + chunk * 4]);
                    const int grow = rbase + i * 32 + prow;
                    if (grow < Mv)
                        *reinterpret_cast<uint4*>(out + (size_t)grow * g.ldc + cbase + chunk * 8) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        } else if (SPLIT) {
            // split residual stream: slabs of 16 rows x 64 columns (accumulator registers 8*half .. 8*half+7 of both
            // column blocks) go through the 4 KiB patch; on the way back every lane owns 8 consecutive columns of a
            // row, so hi and lo leave as 16-byte stores of 128-byte row segments, the residual arrives the same way,
            // and one 8-lane butterfly yields the row's (sum, sum of squares) over this wave's 64 columns.
            float* pf = reinterpret_cast<float*>(patch);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int i = t >> 1, half = t & 1;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        float v = acc[i][j][half * 8 + rr] + bias_l[j];
                        if (k_relu) v = fmaxf(v, 0.f);
                        pf[((rr & 3) + 8 * (rr >> 2) + 4 * h) * 64 + j * 32 + (lane & 31)] = v;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int prow = it * 8 + (lane >> 3), c8 = (lane & 7) * 8;
                    const float4 p0 = *reinterpret_cast<const float4*>(&pf[prow * 64 + c8]);
                    const float4 p1 = *reinterpret_cast<const float4*>(&pf[prow * 64 + c8 + 4]);
                    float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                    const int grow = rbase + t * 16 + prow, gcol = cbase + c8;
                    const bool row_ok = grow < Mv;
                    if (has_res) {
                        const float4 r0 = res[t & 1][2 * it], r1 = res[t & 1][2 * it + 1];
                        if (k_res_split) {                        // x_old = hi + lo
                            float fh[4], fl[4];
                            unpack4_16<F16>(make_uint2(__float_as_uint(r0.x), __float_as_uint(r0.y)), fh);
                            unpack4_16<F16>(make_uint2(__float_as_uint(r1.x), __float_as_uint(r1.y)), fl);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
                            unpack4_16<F16>(make_uint2(__float_as_uint(r0.z), __float_as_uint(r0.w)), fh);
                            unpack4_16<F16>(make_uint2(__float_as_uint(r1.z), __float_as_uint(r1.w)), fl);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
                        } else {
                            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                            v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        }
                    }
                    if (k_add2) {
                        const int crow = (k_map && g.map_add2) ? ridx[t * 2 + it] : (row_ok ? grow : Mv - 1);
                        const float* ap = g.add2 + (size_t)(crow / g.add2_div) * g.ld_add2 + gcol;
                        const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                        v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w;
                        v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
                    }
                    if (k_stats) {
                        // same association order as the generic kernel: 4-column chunk partials, then a butterfly
                        const float s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                        const float q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                                         ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                        const float S = group8_sum(s8), Q = group8_sum(q8);
                        // part-major [N/64][M] pairs: the 8 rows of this group are 64 contiguous bytes
                        if (row_ok && (lane & 7) == 0)
                            reinterpret_cast<float2*>(g.stats_out)[(size_t)(cbase / 64) * g.M + grow] = make_float2(S, Q);
                    }
                    if (row_ok) {
                        const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
                        uint2 ha, la, hb, lb;
                        split4_16<F16>(va, ha, la);
                        split4_16<F16>(vb, hb, lb);
                        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(g.out) + (size_t)grow * g.ldc + gcol) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(g.out_lo) + (size_t)grow * g.ldc + gcol) = make_uint4(la.x, la.y, lb.x, lb.y);
                    }
                }
                if (has_res && t + 2 < 4) load_res(t + 2, t & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            float* pf = reinterpret_cast<float*>(patch);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[i][j][r] + bias_l[j];
                        if (k_relu) v = fmaxf(v, 0.f);
                        pf[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (lane & 31)] = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int prow = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                        float4 v = *reinterpret_cast<const float4*>(&pf[prow * 32 + c4]);
                        const int grow = rbase + i * 32 + prow, gcol = cbase + j * 32 + c4;
                        if (has_res) {
                            const float4 a4 = res[(i * TN + j) & 1][it];
                            v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                        }
                        if (grow < Mv) {
                            if (k_add2) {
                                const float4 a4 = *reinterpret_cast<const float4*>(g.add2 + (size_t)(grow / g.add2_div) * g.ld_add2 + gcol);
                                v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w;
                            }
                            if (!PURE && g.out_dtype != BG_F32)
                                *reinterpret_cast<V4*>(reinterpret_cast<T*>(g.out) + (size_t)grow * g.ldc + gcol) =
                                    E::pack4(v.x, v.y, v.z, v.w);
                            else if (PURE && g.N < BN) {          // narrow convolution: the columns below N only (ldc need not be a multiple of 4)
                                float* o = reinterpret_cast<float*>(g.out) + (size_t)grow * g.ldc + gcol;
                                if (gcol < g.N) o[0] = v.x;
                                if (gcol + 1 < g.N) o[1] = v.y;
                                if (gcol + 2 < g.N) o[2] = v.z;
                                if (gcol + 3 < g.N) o[3] = v.w;
                            } else
                                *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.out) + (size_t)grow * g.ldc + gcol) = v;
                        }
                    }
                    if (has_res && i * TN + j + 2 < TM * TN) load_res(i * TN + j + 2, (i * TN + j) & 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
        }
        if (!has_next) break;
    }
}


// opt-in profiler accounting of `rows` output rows: 2 M N K flops; bytes = what the launch has to move once -- operands, output,
// addends, and for the residual-stream forms the split residual read (hi + lo), the lo plane and the row statistics
static void gemm_cost(const GemmArgs& g, double rows, double& flops, double& bytes) {
    const double osz = g.out_dtype == BG_F32 ? 4.0 : 2.0;
    flops = 2.0 * rows * g.N * (double)g.K;
    bytes = 2.0 * rows * g.K + 2.0 * g.N * (double)g.K + osz * rows * g.N + (g.add ? 4.0 * (rows / g.add_div) * g.N : 0.0) +
            (g.add2 ? 4.0 * (rows / g.add2_div) * g.N : 0.0);
    if (g.bias) bytes += 4.0 * g.N;
    if (g.out_lo) bytes += 2.0 * rows * g.N;                      // lo plane
    if (g.res_hi) bytes += 4.0 * rows * g.N;                      // split residual in: hi + lo
    if (g.stats_out) bytes += 8.0 * rows * (g.N_pad / 64);        // (sum, sum of squares) per row and 64-column group
    if (g.stats_in) bytes += 8.0 * rows * (g.K / G_BK) + 4.0 * g.N;   // LayerNorm fold: the row partials + column sums
}

// phase-group delay of a split-residual launch on the 256 x 256 kernel (bg_tune key 8: > 0 = that many x 1024 cycles, < 0 = off;
// default: P256_SPLIT_STAGGER from P256_SPLIT_STAGGER_ROUNDS rounds of tiles on)
static int split_stagger(const GemmArgs& g) {
    const int t = g_tune[TUNE_GEMM_STAGGER];
    if (t != 0) return t > 0 ? t : 0;
    const int nt_n = g.N_pad / 256, tiles = ((g.M + 255) / 256) * nt_n;
    return tiles >= P256_SPLIT_STAGGER_ROUNDS * ((256 / nt_n) * nt_n) ? P256_SPLIT_STAGGER : 0;
}

// Which kernel runs a 16-bit GEMM (every choice computes bit-identical results; the bg_tune keys named here exist so that the tests
// can compare them -- key 10: 256-kernel mode, 12: split-residual kernel, 15: small-launch threshold, 8: phase-group delay):
//   narrow outputs (N % 128 != 0: fc_out.3, conv_out)            -> generic kernel, 128 x 64 tiles
//   < SMALL_LAUNCH_TILES tiles of 128 x 128 (batch 16)            -> generic kernel, 64 x 64 tiles, 3-deep ring
//   16-bit output, QKV / FFN1 shapes                              -> 256 x 256 kernel on the rows that fill whole rounds (p256_rows)
//   split-residual launches of the encoder layers                 -> 256 x 256 kernel in two phase groups from P256_SPLIT_MIN_ROUNDS
//                                                                    rounds on, the pipelined 128 x 128 kernel (gemm_split.hip) otherwise
//   everything else (embeds, fp32 residual stream, VAE convs)     -> 128 x 128 persistent kernel (>= 64 tiles) / generic 128 x 128
template <bool F16>
static int launch16(const GemmArgs& g, hipStream_t s) {
    const double rows_all = g.rows_hint > 0 ? g.rows_hint : g.M;  // (profiler accounting)
    double rows_tail = rows_all;
    const int m128 = (g.M + 127) / 128, n128 = g.N_pad / 128;
    if (g.N_pad % 128 != 0) {                                     // narrow outputs (fc_out.3: 6/18/48 -> padded 64; conv_out 3)
        double fl, by;
        gemm_cost(g, rows_all, fl, by);
        ProfScope prof(PK_GEMM_BF16_64, fl, by, s);
        hipLaunchKernelGGL((gemm16_kernel<F16, 128, 64, 4, 1, 2>), dim3(m128 * (g.N_pad / 64)), dim3(256), 0, s, g);
        return launch_status("gemm16");
    }
    const int nt = m128 * n128;
    // Small launches (the reference's shipped batch of 16 samples: 960 face tokens -> 48 .. 144 tiles of 128 x 128 on 256 CUs): a
    // 128 x 128 tile per workgroup leaves most of the chip idle while every workgroup walks its 12 K-steps one DMA round trip at a
    // time.  64 x 64 tiles (two waves, 3-deep ring: two K-steps of DMA in flight) put four times as many workgroups on the chip
    // and halve the per-step latency chain; same k order, same epilogue code -> bit-identical results
    // (tests/test_gpu_round4.py).  bg_tune key 15: tile-count threshold (0 = default, -1 = off).
    const int small_nt = g_tune[TUNE_SMALL_TILES] < 0 ? 0 : (g_tune[TUNE_SMALL_TILES] > 0 ? g_tune[TUNE_SMALL_TILES] : SMALL_LAUNCH_TILES);
    if (g.cv_C == 0 && nt < small_nt && (g.stats_in == nullptr || g.K / G_BK <= 16)) {
        double fl, by;
        gemm_cost(g, rows_all, fl, by);
        ProfScope prof(PK_GEMM_BF16_64, fl, by, s);
        hipLaunchKernelGGL((gemm16_kernel<F16, 64, 64, 2, 1, 3>), dim3(((g.M + 63) / 64) * (g.N_pad / 64)), dim3(128), 0, s, g);
        return launch_status("gemm16(64x64)");
    }
    const bool persistent_ok = (g.ldc % 8 == 0) && (g.N == g.N_pad) && (g.add == nullptr || g.ld_add % 4 == 0) &&
                               (g.add2 == nullptr || g.ld_add2 % 4 == 0) && nt >= 64 &&
                               (g.row_map == nullptr || (g.out_lo != nullptr && !g.map_out));   // mapped addends: SPLIT epilogue only
    // (split output / split residual / row statistics / LayerNorm fold are validated in gemm_16bit; both the
    //  persistent and the generic kernel implement them, with bit-identical arithmetic)
    // narrow implicit-GEMM convolution (the VAEs' conv_out: 3 channels): ONE 128-column tile per row panel over weights zero-padded
    // to 128 rows; the epilogue stores only the columns below N (scalar stores: ldc = N)
    const bool narrow_conv = g.cv_C > 0 && g.N < g.N_pad && g.N_pad == 128 && nt >= 64 && g.out_dtype == BG_F32 && !g.add && !g.add2 &&
                             g.act == BG_ACT_NONE && g.add_div == 1 && !g.out_lo && !g.stats_in && !g.row_map && g.bias;
    if (narrow_conv) {
        double fl, by;
        gemm_cost(g, rows_all, fl, by);
        ProfScope prof(PK_GEMM_BF16_128, fl, by, s);
        hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_GENERAL, true, true>), dim3(nt < 512 ? (nt & ~7) : 512), dim3(256), 0, s, g, m128);
        return launch_status("gemm16(narrow conv)");
    }
    if (g.cv_C > 0 && (!persistent_ok || g.out_dtype != BG_F32 || g.out_lo || g.stats_in)) {
        set_error("gemm_16bit: the implicit-GEMM convolution needs the persistent kernel (N %% 128 == 0, >= 64 tiles, fp32 output)");
        return BG_E_SHAPE;
    }
    // 256 x 256 persistent kernel (gemm_p256.hip) on the row panels that fill complete rounds of tiles; a 128 x 128 kernel below
    // runs the remaining rows (bg_common.h: p256_rows).  With a device-side row count the split is only known on the device: both
    // kernels are launched and evaluate the same rule.  bg_tune key 10: 0 = hybrid, 1 = 256 kernel alone wherever eligible, 2 = never.
    bool tail_only = false;
    if (g_tune[TUNE_P256_MODE] != 2 && p256_eligible(g)) {
        double fl, by;
        const bool split = g.out_lo != nullptr;
        if (g_tune[TUNE_P256_MODE] == 1) {
            gemm_cost(g, rows_all, fl, by);
            ProfScope prof(split ? PK_GEMM_P256_SPLIT : PK_GEMM_P256, fl, by, s);     // (booked apart: HBM-bound vs MFMA-bound launches)
            GemmArgs h = g;
            if (split) h.p256_stagger = split_stagger(g);
            return launch_p256<F16>(h, s);
        }
        // upper bound of what the 256 kernel may own (a device-side row count can only be smaller: the split rule is not monotonic,
        // so with one the launch happens whenever ANY row count up to the bound could give it rows)
        const int hmode = g.concurrent ? 2 : 1;
        // Host-side view of the row count: exact without m_dev; with a device-side count the caller's estimate (rows_hint: the
        // drop-in modules count the valid tokens once per mask) or else the bound M.  It only chooses between "both kernels, the
        // device evaluates the rule" and "the 128 kernel alone, no rule" -- either is correct for any actual row count.
        // (rows_plan: the caller knows the device-side count exactly -- brepgen_hip.h)
        const bool plan = g.m_dev != nullptr && g.rows_plan > 0;
        const int rows_est = plan ? (int)fmin((double)g.M, g.rows_plan + 0.5)
                                  : ((g.m_dev != nullptr && g.rows_hint > 0) ? (int)fmin((double)g.M, g.rows_hint + 0.5) : g.M);
        const int rows_hi = p256_rows(rows_est, g.N_pad / 256, split, hmode == 2);
        if (plan && rows_hi >= rows_est) {
            // every row panel of the planned count belongs to the 256 x 256 kernel: it runs ALONE over all the rows present (no rule, no
            // tail kernel that would find nothing to do) -- correct for any actual count, like every other plan
            gemm_cost(g, rows_all, fl, by);
            ProfScope prof(split ? PK_GEMM_P256_SPLIT : PK_GEMM_P256, fl, by, s);
            GemmArgs h = g;
            if (split) h.p256_stagger = split_stagger(g);
            return launch_p256<F16>(h, s);
        }
        // (device-side row count: the partition is read from the table compact_rows wrote -- no table entry, no hybrid launch)
        const int ridx = p256_rule_index(g.N_pad / 256, split, hmode == 2);
        const bool can_tail = persistent_ok && (!g.stats_in || g.K == FOLD_PARTS * G_BK) &&
                              (g.m_dev == nullptr || (g.rule_table != nullptr && ridx >= 0));
        if (rows_hi > 0 && can_tail) {
            GemmArgs h = g;
            h.hybrid = hmode;
            if (g.m_dev) h.rows256_dev = g.rule_table + ridx;
            else h.rows256_host = p256_rows(g.M, g.N_pad / 256, split, hmode == 2);
            if (split) h.p256_stagger = split_stagger(g);
            const double rows_p = fmin((double)p256_rows((int)rows_all, g.N_pad / 256, split, hmode == 2), rows_all);
            rows_tail = rows_all - rows_p;
            gemm_cost(g, rows_p, fl, by);
            int rc;
            {
                ProfScope prof(split ? PK_GEMM_P256_SPLIT : PK_GEMM_P256, fl, by, s);
                rc = launch_p256<F16>(h, s);
            }
            if (rc) return rc;
            if (g.m_dev == nullptr && rows_hi >= g.M) return 0;   // exact host-side row count and everything fitted
            tail_only = true;
        }
    }
    GemmArgs gt = g;
    gt.hybrid = tail_only ? (g.concurrent ? 2 : 1) : 0;
    if (tail_only) {
        const bool split = g.out_lo != nullptr;
        if (g.m_dev) gt.rows256_dev = g.rule_table + p256_rule_index(g.N_pad / 256, split, g.concurrent != 0);
        else gt.rows256_host = p256_rows(g.M, g.N_pad / 256, split, g.concurrent != 0);
    }
    const GemmArgs& g_ = gt;
    double fl_t, by_t;
    gemm_cost(g, rows_tail, fl_t, by_t);
    // residual-stream GEMMs of the encoder layers: the software-pipelined split kernel (gemm_split.hip) takes the rows the 256 x 256
    // kernel does not (bg_tune key 12 = 1: the 128 x 128 persistent kernel's split epilogue instead, for the bit-equality tests)
    if (g_tune[TUNE_SPLIT_PIPE] != 1 && split_pipe_eligible(g_)) {
        ProfScope prof(PK_GEMM_SPLIT, fl_t, by_t, s);
        return launch_split_pipe<F16>(g_, s);
    }
    ProfScope prof(PK_GEMM_BF16_128, fl_t, by_t, s);
    if (!persistent_ok || (g.stats_in && g.K != FOLD_PARTS * G_BK)) {   // non-persistent 128 x 128, 2-stage ring
        hipLaunchKernelGGL((gemm16_kernel<F16, 128, 128, 2, 2, 2>), dim3(m128 * n128), dim3(256), 0, s, g_);
    } else {
        const int grid = nt < 512 ? (nt & ~7) : 512;              // 2 resident workgroups per CU x 256 CUs, a multiple of 8
        const bool fast = g.out_dtype != BG_F32 && g.add == nullptr && g.add2 == nullptr && g.out_lo == nullptr;
        if (g.cv_C > 0 && g.act == BG_ACT_NONE && !g.add2 && g.add_div == 1) {
            // implicit-GEMM convolution as the VAE passes issue it: fp32 output (+ fp32 residual of the same shape), the run-time
            // options of the general epilogue folded away (904 -> 242 VALU instructions per tile epilogue; -4.5 % per decode pass)
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_GENERAL, true, true>), dim3(grid), dim3(256), 0, s, g_, m128);
        } else if (g.cv_C > 0) {                                  // implicit-GEMM convolution, any other option set
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_GENERAL, true>), dim3(grid), dim3(256), 0, s, g_, m128);
        } else if (g.out_lo) {
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_SPLIT>), dim3(grid), dim3(256), 0, s, g_, m128);
        } else if (g.stats_in) {
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_FOLD16>), dim3(grid), dim3(256), 0, s, g_, m128);
        } else if (fast) {
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_PLAIN16>), dim3(grid), dim3(256), 0, s, g_, m128);
        } else {
            hipLaunchKernelGGL((gemm16_persistent_kernel<F16, P_GENERAL>), dim3(grid), dim3(256), 0, s, g_, m128);
        }
    }
    return launch_status("gemm16");
}

int gemm_16bit(const GemmArgs& g, int ab_dtype, hipStream_t s) {
    if (g.M <= 0) return 0;
    if (g.cv_C > 0 && g.N_pad % 128 != 0) {
        set_error("gemm_16bit: the implicit-GEMM convolution needs N %% 128 == 0");
        return BG_E_SHAPE;
    }
    if (g.K % G_BK != 0 || g.N_pad % 64 != 0 || g.lda % 8 != 0) {
        set_error("gemm_16bit: need K %% 64 == 0, N_pad %% 64 == 0, lda %% 8 == 0 (K=%d N_pad=%d lda=%d)", g.K, g.N_pad, g.lda);
        return BG_E_SHAPE;
    }
    if ((reinterpret_cast<uintptr_t>(g.a) & 15) || (reinterpret_cast<uintptr_t>(g.w) & 15) ||
        (reinterpret_cast<uintptr_t>(g.out) & 15)) {
        set_error("gemm_16bit: a / w / out must be 16-byte aligned");
        return BG_E_ALIGN;
    }
    if (g.out_dtype != BG_F32 && g.out_dtype != ab_dtype) {
        set_error("gemm_16bit: a 16-bit output must have the operand dtype (out %d, operands %d)", g.out_dtype, ab_dtype);
        return BG_E_DTYPE;
    }
    if (g.stats_in || g.out_lo || g.res_hi || g.stats_out) {
        const bool ok_common = g.N == g.N_pad && g.ldc % 8 == 0;
        const bool ok_fold = g.stats_in == nullptr || (g.colsum && g.bias && g.K / G_BK <= 16 && g.out_dtype == ab_dtype &&
                                                       !g.add && !g.add2 && !g.out_lo && !g.res_hi && !g.stats_out);
        const bool ok_split = g.out_lo == nullptr || g.out_dtype == ab_dtype;
        const bool ok_res = g.res_hi == nullptr || (g.res_lo && g.out_lo && !g.add && !g.add2 && g.ld_res % 8 == 0);
        const bool ok_stats = g.stats_out == nullptr || g.out_lo != nullptr;
        if (!(ok_common && ok_fold && ok_split && ok_res && ok_stats)) {
            set_error("gemm_16bit: invalid split-residual / LayerNorm-fold argument combination");
            return BG_E_ARG;
        }
    }
    return ab_dtype == BG_F16 ? launch16<true>(g, s) : launch16<false>(g, s);
}

}  // namespace bg
