// FFN1 + ReLU + FFN2 + split residual of a pre-LN encoder layer as ONE launch (network.py:1076-1078: dim_feedforward = 1024,
// norm_first = True -> torch/nn/modules/transformer.py _ff_block):
//
//   h[m, :]        = relu(rstd_m * (x_hi[m, :] W1'^T) - mean_m rstd_m colsum(W1') + b1')          (LayerNorm fold, 16-bit, [64, 1024] in LDS)
//   (hi, lo)[m, :] = split(h[m, :] W2^T + b2 + hi[m, :] + lo[m, :]),  row statistics per 64 columns   (in place)
//
// bit for bit what gemm(P_FOLD16, ReLU) followed by gemm(P_SPLIT + statistics) produce -- but the hidden tensor (2 KB per token,
// written and read back: 126 MB per layer at M = 30 720) never leaves the CU, and a layer loses a launch and a store burst.
//
// Organisation.  One 8-wave workgroup per CU walks 64-row panels.  A 64-row panel cannot amortise a weight tile staged in LDS
// (512 B of weights per MFMA whatever the staging), but its weight fragments are WAVE-PRIVATE: wave w owns hidden columns
// [128 w, 128 w + 128) in phase 1 and output columns [96 w, 96 w + 96) in phase 2, all 64 rows.  So the weights never touch the
// LDS: they are packed on the host in MFMA FRAGMENT ORDER (brepgen_amd/network.py: ffn_fragment_order: per wave, per 16-wide
// k-slice, per 32-column tile one contiguous KiB = lane l's eight k values of column l & 31) and stream from the L2 into registers
// as fully coalesced global_load_dwordx4, two k-slices ahead of their use; the activations -- the x panel (96 KiB), then the hidden
// panel (128 KiB) -- are the only LDS residents, read as the shared operand by all eight waves (0.25 / 0.33 KiB of LDS reads per
// MFMA; the 128 x 128 kernel reads 1 KiB).  No ring, no DMA waits, and NO BARRIER inside either K loop: five workgroup barriers per
// panel in all (panel in, phase 1 out, hidden in, phase 2 out, per output half).
//   phase 1: 48 k-slices x (4 weight fragments + 2 x fragments -> 8 MFMAs), accumulators 2 x 4 x 16 = 128 registers
//   phase 2: 64 k-slices x (3 weight fragments + 2 h fragments -> 6 MFMAs), accumulators 2 x 3 x 16 =  96 registers
// Both products are computed transposed (weights as the MFMA's A operand), as in gemm_p256.hip: a lane owns ONE row and four
// consecutive columns per accumulator quad -- LayerNorm coefficients are per-lane scalars, 16-bit packing needs no exchange -- and
// the k order is ascending 16-wide slices like every other GEMM kernel here, hence the same bits.
// Epilogue 2 goes through LDS in two halves of 32 rows x 768 fp32 columns (96 KiB): every wave parks its accumulators, then the
// workgroup re-reads them with a lane owning 8 consecutive columns of a row, which is the access pattern, the arithmetic and the
// statistics association of the split epilogues in gemm_p256.hip / gemm_split.hip (residual as 16-byte loads, (hi, lo) as 16-byte
// stores of whole 128-byte lines, one 8-lane butterfly per 64-column group).
#include "gemm16.h"

namespace bg {

constexpr int FF_ROWS = 64, FF_D = 768, FF_H = 1024;
constexpr int FF_XROW = FF_D * 2, FF_HROW = FF_H * 2, FF_OROW = FF_D * 4;      // LDS row pitches (bytes): x panel, hidden panel, fp32 output half
constexpr int FF_LDS = FF_ROWS * FF_HROW;                                        // 128 KiB: the hidden panel is the largest resident


__device__ __forceinline__ int ff_opaque(int x) {      // a value the compiler must treat as unknown here (see the panel loop)
    asm volatile("" : "+v"(x));
    return x;
}

template <bool F16>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnArgs g) {
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FF_LDS];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Mv = g.m_dev ? *g.m_dev : g.M;
    const int n_panel = (Mv + FF_ROWS - 1) / FF_ROWS;

    const unsigned char* xh_b = reinterpret_cast<const unsigned char*>(g.xh);
    // this wave's weight streams: wave-uniform bases (scalar registers) + the lane's 16 bytes; k-slice s at + s * 4 KiB / 3 KiB
    const unsigned char* w1b = reinterpret_cast<const unsigned char*>(g.w1f) + (size_t)wave * (48 * 4096);
    const unsigned char* w2b = reinterpret_cast<const unsigned char*>(g.w2f) + (size_t)wave * (64 * 3072);

    for (int p = blockIdx.x; p < n_panel; p += gridDim.x) {
        const int r0 = p * FF_ROWS;
        // Everything lane-dependent is derived from an OPAQUE thread id inside the panel loop: hoisted out of it, these addresses
        // would be live across both K loops, which run near the register limit, and be spilled (hipcc answers every scratch reload
        // with s_waitcnt vmcnt(0): the weight stream's loads in flight would be drained).
        const int tid = ff_opaque(threadIdx.x);
        const int ln = tid & 63, l31 = ln & 31, hq = ln >> 5;
        // shared-operand fragment addresses: row 32 i + l31, 16-byte chunk 2 s + hq, XOR-swizzled by the row's low four bits (both
        // pitches are multiples of 256 B: without it the 16 lanes of a ds_read_b128 group would all hit the same banks)
        const unsigned sw = (unsigned)(l31 & 15);
        const unsigned lane16 = (unsigned)ln * 16u;
        // ---- panel in: 64 rows x 1536 B -> LDS (12 x 16 B per thread), rows past the end clamped (never stored) ----
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int c = it * 512 + tid;                         // 16-byte chunk of the panel: row c / 96, chunk c % 96
            const int row = c / 96, ch = c % 96;
            int grow = r0 + row;
            grow = grow < Mv ? grow : Mv - 1;
            const uint4 v = *reinterpret_cast<const uint4*>(xh_b + (size_t)grow * FF_XROW + ch * 16);
            *reinterpret_cast<uint4*>(lds + row * FF_XROW + ((ch ^ (row & 15)) << 4)) = v;
        }
        // (rstd, -mean rstd) of the lane's two rows from the twelve partials, in the association order of every other kernel
        float2 cf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int grow = r0 + 32 * i + l31;
            grow = grow < Mv ? grow : Mv - 1;
            float ps[16], pq[16];
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                const float2 v = pp < FOLD_PARTS ? reinterpret_cast<const float2*>(g.stats)[(size_t)pp * g.m_stride + grow] : make_float2(0.f, 0.f);
                ps[pp] = v.x; pq[pp] = v.y;
            }
            cf[i] = ln_fold_coeffs(tree16(ps), tree16(pq), FF_D, g.ln_eps);
        }
        __syncthreads();

        // ---- phase 1: h = x W1'^T, wave w: hidden columns 128 w .. + 127 ----
        f32x16 acc1[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
        {
            // software pipeline: weight fragments three k-slices ahead (global -> registers), x fragments one slice ahead (LDS)
            V8 wf[3][4], xf[2][2];
            auto ldw = [&](int b, int s) {                          // (past the end: the last slice again, never used)
                const unsigned char* src = w1b + (size_t)(s < 47 ? s : 47) * 4096;
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[b][j] = *reinterpret_cast<const V8*>(src + j * 1024 + lane16);
                __builtin_amdgcn_sched_barrier(0);                  // (hipcc would otherwise sink every load to just before its use)
            };
            auto ldx = [&](int b, int s) {
                const unsigned ch = (((unsigned)(2 * (s < 47 ? s : 47) + hq)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[b][i] = *reinterpret_cast<const V8*>(lds + (unsigned)(32 * i + l31) * FF_XROW + ch);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mm = [&](int b, int xb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc1[i][j] = E::mfma(wf[b][j], xf[xb][i], acc1[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            ldw(0, 0); ldw(1, 1); ldw(2, 2); ldx(0, 0);
#pragma unroll 1
            for (int s = 0; s < 48; s += 6) {
                ldx(1, s + 1); mm(0, 0); ldw(0, s + 3);
                ldx(0, s + 2); mm(1, 1); ldw(1, s + 4);
                ldx(1, s + 3); mm(2, 0); ldw(2, s + 5);
                ldx(0, s + 4); mm(0, 1); ldw(0, s + 6);
                ldx(1, s + 5); mm(1, 0); ldw(1, s + 7);
                ldx(0, s + 6); mm(2, 1); ldw(2, s + 8);
            }
        }
        __syncthreads();                                            // every wave is done with the x panel: h may overwrite it

        // ---- epilogue 1: LayerNorm fold + bias + ReLU -> 16 bits -> the hidden panel (row-major, same swizzle) ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wave * 128 + j * 32 + 8 * q + 4 * hq;
                const float4 b = *reinterpret_cast<const float4*>(g.b1 + col);
                const float4 c = *reinterpret_cast<const float4*>(g.colsum1 + col);
                const float b4[4] = {b.x, b.y, b.z, b.w}, c4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(ln_fold_apply(acc1[i][j][4 * q + e], cf[i].x, cf[i].y, c4[e], b4[e]), 0.f);
                    union { V4 v; uint2 u; } pk;
                    pk.v = E::pack4(v[0], v[1], v[2], v[3]);
                    const unsigned c16 = (unsigned)(wave * 16 + j * 4 + q);
                    *reinterpret_cast<uint2*>(lds + (unsigned)(32 * i + l31) * FF_HROW + ((c16 ^ sw) << 4) + hq * 8) = pk.u;
                }
            }
        __syncthreads();

        // ---- phase 2: y = h W2^T, wave w: output columns 96 w .. + 95 ----
        f32x16 acc2[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        {
            V8 wf[4][3], hf[2][2];
            auto ldw = [&](int b, int s) {
                const unsigned char* src = w2b + (size_t)(s < 63 ? s : 63) * 3072;
#pragma unroll
                for (int j = 0; j < 3; ++j) wf[b][j] = *reinterpret_cast<const V8*>(src + j * 1024 + lane16);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto ldh = [&](int b, int s) {
                const unsigned ch = (((unsigned)(2 * (s < 63 ? s : 63) + hq)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) hf[b][i] = *reinterpret_cast<const V8*>(lds + (unsigned)(32 * i + l31) * FF_HROW + ch);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mm = [&](int b, int hb) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc2[i][j] = E::mfma(wf[b][j], hf[hb][i], acc2[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
            ldw(0, 0); ldw(1, 1); ldw(2, 2); ldw(3, 3); ldh(0, 0);
#pragma unroll 1
            for (int s = 0; s < 64; s += 4) {
                ldh(1, s + 1); mm(0, 0); ldw(0, s + 4);
                ldh(0, s + 2); mm(1, 1); ldw(1, s + 5);
                ldh(1, s + 3); mm(2, 0); ldw(2, s + 6);
                ldh(0, s + 4); mm(3, 1); ldw(3, s + 7);
            }
        }
        __syncthreads();                                            // the hidden panel is dead

        // ---- epilogue 2, per half of 32 rows: accumulators -> fp32 [32, 768] in LDS -> (8 rows x 64 columns) blocks ----
        const int ln2 = ff_opaque(threadIdx.x) & 63, k8 = ln2 & 7, r8 = ln2 >> 3;
        T* out_hi = reinterpret_cast<T*>(g.xh);
        T* out_lo = reinterpret_cast<T*>(g.xl);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned c4 = (unsigned)(wave * 24 + j * 8 + 2 * q + hq);            // 16-byte chunk of the fp32 row
                    *reinterpret_cast<float4*>(lds + l31 * FF_OROW + ((c4 ^ sw) << 4)) =
                        make_float4(acc2[i][j][4 * q], acc2[i][j][4 * q + 1], acc2[i][j][4 * q + 2], acc2[i][j][4 * q + 3]);
                }
            __syncthreads();
            // 4 row octets x 12 column groups = 48 blocks, six per wave; the residual of all six requested before the first is used
            uint4 rh[6], rl[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int u = wave + 8 * k, grp = u % 12, oct = u / 12;
                int grow = r0 + 32 * i + 8 * oct + r8;
                grow = grow < Mv ? grow : Mv - 1;
                const size_t o = (size_t)grow * FF_D + grp * 64 + k8 * 8;
                rh[k] = *reinterpret_cast<const uint4*>(out_hi + o);
                rl[k] = *reinterpret_cast<const uint4*>(out_lo + o);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int u = wave + 8 * k, grp = u % 12, oct = u / 12;
                const int prow = 8 * oct + r8;
                const unsigned rsw = (unsigned)(prow & 15);
                const unsigned c0 = (unsigned)(grp * 16 + 2 * k8);
                const float4 p0 = *reinterpret_cast<const float4*>(lds + prow * FF_OROW + ((c0 ^ rsw) << 4));
                const float4 p1 = *reinterpret_cast<const float4*>(lds + prow * FF_OROW + (((c0 + 1) ^ rsw) << 4));
                const float4 bias0 = *reinterpret_cast<const float4*>(g.b2 + grp * 64 + k8 * 8);
                const float4 bias1 = *reinterpret_cast<const float4*>(g.b2 + grp * 64 + k8 * 8 + 4);
                float v[8] = {p0.x + bias0.x, p0.y + bias0.y, p0.z + bias0.z, p0.w + bias0.w,
                              p1.x + bias1.x, p1.y + bias1.y, p1.z + bias1.z, p1.w + bias1.w};
                float fh[4], fl[4];
                unpack4_16<F16>(make_uint2(rh[k].x, rh[k].y), fh);
                unpack4_16<F16>(make_uint2(rl[k].x, rl[k].y), fl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
                unpack4_16<F16>(make_uint2(rh[k].z, rh[k].w), fh);
                unpack4_16<F16>(make_uint2(rl[k].z, rl[k].w), fl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
                const int grow = r0 + 32 * i + prow;
                const bool row_ok = grow < Mv;                    // (in place: a clamped duplicate row must not be written)
                const float s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                const float q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                                 ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                const float S = group8_sum(s8), Q = group8_sum(q8);
                if (row_ok && k8 == 0) reinterpret_cast<float2*>(g.stats)[(size_t)grp * g.m_stride + grow] = make_float2(S, Q);
                if (row_ok) {
                    const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
                    uint2 ha, la, hb, lb;
                    split4_16<F16>(va, ha, la);
                    split4_16<F16>(vb, hb, lb);
                    const size_t o = (size_t)grow * FF_D + grp * 64 + k8 * 8;
                    *reinterpret_cast<uint4*>(out_hi + o) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                    *reinterpret_cast<uint4*>(out_lo + o) = make_uint4(la.x, la.y, lb.x, lb.y);
                }
            }
            __syncthreads();                                      // the half is consumed: the next one (or the next panel) may overwrite it
        }
    }
}

bool ffn_fused_eligible(const FfnArgs& g, int dtype) {
    auto al = [](const void* p) { return p != nullptr && ((uintptr_t)p & 15) == 0; };
    return (dtype == BG_BF16 || dtype == BG_F16) && al(g.xh) && al(g.xl) && al(g.stats) && al(g.w1f) && al(g.b1) && al(g.colsum1) &&
           al(g.w2f) && al(g.b2) && g.M > 0 && g.m_stride >= g.M && (size_t)g.M * FF_XROW < 0xffffffffull;
}

int ffn_fused(const FfnArgs& g, int dtype, hipStream_t s, double rows_hint) {
    if (g.M <= 0) return 0;
    const int panels = (g.M + FF_ROWS - 1) / FF_ROWS;
    const int grid = panels < 256 ? panels : 256;
    const double rows = rows_hint > 0 ? rows_hint : (double)g.M;
    // algorithmic cost: both products; bytes = x_hi rows in + (hi, lo) residual in and out + statistics in and out + the weights once
    ProfScope ps(PK_FFN_FUSED, 2.0 * rows * (double)FF_D * FF_H * 2.0,
                 rows * (FF_D * 2.0 * 5 + 2 * 12 * 8.0) + 2.0 * FF_D * FF_H * 2.0 + (FF_H * 2 + FF_D) * 4.0, s);
    if (dtype == BG_F16) hipLaunchKernelGGL((ffn_fused_kernel<true>), dim3(grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((ffn_fused_kernel<false>), dim3(grid), dim3(512), 0, s, g);
    return launch_status("ffn_fused");
}

}  // namespace bg

extern "C" int bg_ffn_fused_fwd(void* x_hi, void* x_lo, float* stats, const void* w1_frag, const float* b1, const float* colsum1,
                                const void* w2_frag, const float* b2, int M, int m_stride, const int* m_dev, int dtype, float ln_eps,
                                bg_stream_t stream) {
    bg::FfnArgs g{x_hi, x_lo, stats, w1_frag, b1, colsum1, w2_frag, b2, M, m_stride, m_dev, ln_eps};
    BG_REQUIRE(M >= 0, BG_E_ARG, "bg_ffn_fused_fwd: negative M");
    if (M == 0) return 0;
    BG_REQUIRE(bg::ffn_fused_eligible(g, dtype), BG_E_ARG,
               "bg_ffn_fused_fwd: 16-bit operands, 16-byte aligned non-null pointers, m_stride >= M, M * 1536 < 2^32");
    return bg::ffn_fused(g, dtype, (hipStream_t)stream, 0.0);
}
