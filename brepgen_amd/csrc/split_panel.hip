// out-proj / FFN2 of an encoder layer (split residual in place + row statistics: gemm(P_SPLIT + statistics)) on 64-ROW PANELS with the
// weights streamed straight into registers -- the second half of csrc/ffn_fused.hip as a launch of its own (round 6 experiment):
//
//   (hi, lo)[m, :] = split(a[m, :] W^T + b + hi[m, :] + lo[m, :]),  row statistics per 64 columns        a: [M, K] 16-bit, K = 768 | 1024
//
// Same organisation as there: one 8-wave workgroup per CU, the A panel (64 x K, 96 / 128 KiB) is the only LDS resident and the shared
// MFMA operand, wave w owns output columns [96 w, 96 w + 96) and streams ITS weight fragments (host-packed in MFMA fragment order,
// network.ffn_fragment_order) from the L2 into registers several k-slices ahead, no barrier inside the K loop, wave-private epilogue
// through a 3 KiB patch (8-row fp32 slabs; residual octets requested three slabs ahead, the first three before the K loop), the next
// panel's A rows requested mid-epilogue.  Two workgroup barriers per panel.  Bit-identical to the split-residual GEMM kernels
// (same ascending 16-wide k-slices, the same epilogue arithmetic and statistics association).
#include "gemm16.h"

namespace bg {

constexpr int SP_ROWS = 64, SP_D = 768;
constexpr int SP_PROW = 96 * 4, SP_PATCH_W = 8 * SP_PROW;                        // epilogue patch of a wave: 8 rows x 96 fp32 columns = 3 KiB
constexpr int SP_PANEL = SP_ROWS * 1024 * 2;                                     // 128 KiB: the A panel at K = 1024 (96 KiB at K = 768)
constexpr int SP_PATCH = SP_PANEL, SP_PART = SP_PATCH + 8 * SP_PATCH_W;          // [128K, 152K) patches, [152K, 156K) half-group partials
constexpr int SP_LDS = SP_PART + 8 * SP_ROWS * 8;

__device__ __forceinline__ int sp_opaque(int x) {      // a value the compiler must treat as unknown here (ffn_fused.hip: the panel loop)
    asm volatile("" : "+v"(x));
    return x;
}

template <bool F16, int K>
__global__ __launch_bounds__(512) void split_panel_kernel(SplitPanelArgs g) {
    using E = Elem<F16>;
    using V8 = typename E::V8;
    constexpr int KS = K / 16, AROW = K * 2, CH = K / 8, NREG = (SP_ROWS * CH) / 512;       // k-slices, A row pitch, 16-byte chunks per row, chunks per thread (12 | 16)
    static_assert(K == 768 || K == 1024, "encoder-layer shapes");
    __shared__ __attribute__((aligned(16))) unsigned char lds[SP_LDS];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Mv = g.m_dev ? *g.m_dev : g.M;
    const int n_panel = (Mv + SP_ROWS - 1) / SP_ROWS;
    if ((int)blockIdx.x >= n_panel) return;                        // uniform per workgroup, before any barrier

    const unsigned char* a_b = reinterpret_cast<const unsigned char*>(g.a);
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(g.wf) + (size_t)wave * (KS * 3072);
    unsigned char* const patch = lds + SP_PATCH + wave * SP_PATCH_W;
    const int odd = wave & 1;
    const int grp_full = 3 * (wave >> 1) + 2 * odd;               // (the shared group is 3 * (wave >> 1) + 1)

    // the A panel on its way to the LDS: NREG x 16 B per thread, rows past the end clamped; requested one panel ahead (named
    // registers: ffn_fused.hip says why), written after the barrier that retires the previous panel
    uint4 a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15;
#define SP_AREGS12(X) X(0, a0) X(1, a1) X(2, a2) X(3, a3) X(4, a4) X(5, a5) X(6, a6) X(7, a7) X(8, a8) X(9, a9) X(10, a10) X(11, a11)
#define SP_AREGS16(X) SP_AREGS12(X) X(12, a12) X(13, a13) X(14, a14) X(15, a15)
#define SP_REQ(it, reg)                                                                                  \
    if (it < NREG) {                                                                                     \
        const int c = it * 512 + tid_;                                                                   \
        int grow = r0_ + c / CH;                                                                         \
        grow = grow < Mv ? grow : Mv - 1;                                                                \
        reg = *reinterpret_cast<const uint4*>(a_b + ((unsigned)grow * AROW + (unsigned)(c % CH) * 16u)); \
    }
#define SP_STO(it, reg)                                                                                  \
    if (it < NREG) {                                                                                     \
        const int c = it * 512 + tid_;                                                                   \
        const int row = c / CH, ch = c % CH;                                                             \
        *reinterpret_cast<uint4*>(lds + row * AROW + ((ch ^ (row & 15)) << 4)) = reg;                    \
    }
#define SP_PANEL_REQUEST(r0v)                                  \
    {                                                          \
        const int tid_ = sp_opaque(threadIdx.x), r0_ = (r0v);  \
        SP_AREGS16(SP_REQ)                                     \
        __builtin_amdgcn_sched_barrier(0);                     \
    }
#define SP_PANEL_STORE()                                       \
    {                                                          \
        const int tid_ = sp_opaque(threadIdx.x);               \
        SP_AREGS16(SP_STO)                                     \
    }
    a12 = a13 = a14 = a15 = make_uint4(0, 0, 0, 0);
    SP_PANEL_REQUEST(blockIdx.x * SP_ROWS)

    for (int p = blockIdx.x; p < n_panel; p += gridDim.x) {
        const int r0 = p * SP_ROWS;
        const bool has_next = p + (int)gridDim.x < n_panel;
        const int r0n = has_next ? (p + (int)gridDim.x) * SP_ROWS : r0;   // (no next panel: this one again, unused -- unconditional on purpose, see ffn_fused.hip)
        const int ln = sp_opaque(threadIdx.x) & 63, l31 = ln & 31, hq = ln >> 5;
        const unsigned sw = (unsigned)(l31 & 15);
        const unsigned lane16 = (unsigned)ln * 16u;
        SP_PANEL_STORE()
        __syncthreads();

        unsigned char* const out_hi = reinterpret_cast<unsigned char*>(g.xh);     // (32-bit byte offsets on uniform bases: M * 1536 < 2^32)
        unsigned char* const out_lo = reinterpret_cast<unsigned char*>(g.xl);
        const int ln2 = sp_opaque(threadIdx.x) & 63;
        const int k8 = ln2 & 7, rF = ln2 >> 3;                      // whole group: lane = 8 row + octet
        const int k4 = ln2 & 3, rH = (ln2 >> 2) & 7;                // half group (lanes 0-31; 32-63 shadow them, nothing stored)
        const int colF = 96 * wave + 32 * odd + 8 * k8, colH = 96 * wave + 64 * (1 - odd) + 8 * k4;
        uint4 rb[4][4];                                             // residual octets of a slab: whole-group item (hi, lo), half-group item (hi, lo)
        auto res_request = [&](int b, int t) {
            int gF = r0 + 8 * t + rF, gH = r0 + 8 * t + rH;
            gF = gF < Mv ? gF : Mv - 1;
            gH = gH < Mv ? gH : Mv - 1;
            const unsigned oF = ((unsigned)gF * SP_D + (unsigned)colF) * 2u, oH = ((unsigned)gH * SP_D + (unsigned)colH) * 2u;
            rb[b][0] = *reinterpret_cast<const uint4*>(out_hi + oF);
            rb[b][1] = *reinterpret_cast<const uint4*>(out_lo + oF);
            rb[b][2] = *reinterpret_cast<const uint4*>(out_hi + oH);
            rb[b][3] = *reinterpret_cast<const uint4*>(out_lo + oH);
            __builtin_amdgcn_sched_barrier(0);
        };
        res_request(0, 0); res_request(1, 1); res_request(2, 2);

        // ---- K loop: y = a W^T, wave w: output columns 96 w .. + 95; weight fragments four k-slices ahead, A fragments one ----
        f32x16 acc2[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        {
            V8 wf[4][3], hf[2][2];
            auto ldw = [&](int b, int s) {                          // (past the end: the last slice again, never used)
                const unsigned char* src = wb + (size_t)(s < KS - 1 ? s : KS - 1) * 3072;
#pragma unroll
                for (int j = 0; j < 3; ++j) wf[b][j] = *reinterpret_cast<const V8*>(src + j * 1024 + lane16);
                __builtin_amdgcn_sched_barrier(0);                  // (hipcc would otherwise sink every load to just before its use)
            };
            auto ldh = [&](int b, int s) {
                const unsigned ch = (((unsigned)(2 * (s < KS - 1 ? s : KS - 1) + hq)) ^ sw) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) hf[b][i] = *reinterpret_cast<const V8*>(lds + (unsigned)(32 * i + l31) * AROW + ch);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto mm = [&](int b, int hb) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc2[i][j] = E::mfma(wf[b][j], hf[hb][i], acc2[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int b = 0; b < 4; ++b) ldw(b, b);
            ldh(0, 0);
#pragma unroll 1
            for (int s = 0; s < KS; s += 4) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    ldh((b + 1) & 1, s + b + 1); mm(b, b & 1); ldw(b, s + b + 4);
                }
            }
        }
        // (no barrier: the epilogue works out of the wave's private patch; the A panel is retired by the barrier at its end)

        // ---- epilogue: split residual + statistics, wave-private, eight slabs of 8 rows x 96 fp32 columns ----
        {
            const unsigned chF = (unsigned)(8 * odd + 2 * k8), chH = (unsigned)(16 * (1 - odd) + 2 * k4);      // first 16-byte chunk in the patch row
            const float4 bF0 = *reinterpret_cast<const float4*>(g.bias + colF), bF1 = *reinterpret_cast<const float4*>(g.bias + colF + 4);
            const float4 bH0 = *reinterpret_cast<const float4*>(g.bias + colH), bH1 = *reinterpret_cast<const float4*>(g.bias + colH + 4);
            // one item: 8 consecutive columns of a row -> v = acc + bias + hi + lo; returns (sum, sum of squares) of the octet, stores (hi, lo)
            auto item = [&](const unsigned char* prow_base, unsigned ch, unsigned rsw, const float4& b0, const float4& b1v, const uint4& h4, const uint4& l4,
                            int grow, int col, bool live, float& s8, float& q8) {
                const float4 p0 = *reinterpret_cast<const float4*>(prow_base + ((ch ^ rsw) << 4));
                const float4 p1 = *reinterpret_cast<const float4*>(prow_base + (((ch + 1) ^ rsw) << 4));
                float v[8] = {p0.x + b0.x, p0.y + b0.y, p0.z + b0.z, p0.w + b0.w, p1.x + b1v.x, p1.y + b1v.y, p1.z + b1v.z, p1.w + b1v.w};
                float fh[4], fl[4];
                unpack4_16<F16>(make_uint2(h4.x, h4.y), fh);
                unpack4_16<F16>(make_uint2(l4.x, l4.y), fl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
                unpack4_16<F16>(make_uint2(h4.z, h4.w), fh);
                unpack4_16<F16>(make_uint2(l4.z, l4.w), fl);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
                s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                if (live) {
                    const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
                    uint2 ha, la, hb, lb;
                    split4_16<F16>(va, ha, la);
                    split4_16<F16>(vb, hb, lb);
                    const unsigned o = ((unsigned)grow * SP_D + (unsigned)col) * 2u;
                    *reinterpret_cast<uint4*>(out_hi + o) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                    *reinterpret_cast<uint4*>(out_lo + o) = make_uint4(la.x, la.y, lb.x, lb.y);
                }
            };
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = t >> 2;
                // the next x panel travels while the second half of this one's rows is finished (requested here, not earlier: the
                // registers of the first row tile's accumulators are free now)
                if (t == 4) { SP_PANEL_REQUEST(r0n) }
                if ((l31 >> 3) == (t & 3)) {
                    const unsigned prow = (unsigned)(l31 & 7);
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned c = (unsigned)(8 * j + 2 * q + hq);
                            *reinterpret_cast<float4*>(patch + prow * SP_PROW + ((c ^ prow) << 4)) =
                                make_float4(acc2[i][j][4 * q], acc2[i][j][4 * q + 1], acc2[i][j][4 * q + 2], acc2[i][j][4 * q + 3]);
                        }
                }
                __builtin_amdgcn_wave_barrier();                  // LDS executes a wave's accesses in order: no wait needed
                if (t + 3 < 8) res_request((t + 3) & 3, t + 3);
                {
                    const int grow = r0 + 8 * t + rF;
                    float s8, q8;
                    item(patch + rF * SP_PROW, chF, (unsigned)rF, bF0, bF1, rb[t & 3][0], rb[t & 3][1], grow, colF, grow < Mv, s8, q8);
                    const float S = group8_sum(s8), Q = group8_sum(q8);
                    if (grow < Mv && k8 == 0) reinterpret_cast<float2*>(g.stats)[(size_t)grp_full * g.m_stride + grow] = make_float2(S, Q);
                }
                {
                    const int grow = r0 + 8 * t + rH;
                    float s8, q8;
                    item(patch + rH * SP_PROW, chH, (unsigned)rH, bH0, bH1, rb[t & 3][2], rb[t & 3][3], grow, colH, grow < Mv && ln2 < 32, s8, q8);
                    s8 += dpp_mov<0xB1>(s8); s8 += dpp_mov<0x4E>(s8);        // the quad's four octets: (o0 + o1) + (o2 + o3)
                    q8 += dpp_mov<0xB1>(q8); q8 += dpp_mov<0x4E>(q8);
                    // the shared group's statistics = this wave's half + the neighbour's: parked here, merged after the barrier
                    if (ln2 < 32 && k4 == 0) reinterpret_cast<float2*>(lds + SP_PART)[wave * SP_ROWS + 8 * t + rH] = make_float2(s8, q8);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                            // A panel and patches retired; every half-group partial is in place
        if (threadIdx.x < 256) {
            const int pi = threadIdx.x >> 6, row = threadIdx.x & 63;
            const float2 a = reinterpret_cast<const float2*>(lds + SP_PART)[(2 * pi) * SP_ROWS + row];          // first half  (even wave)
            const float2 b = reinterpret_cast<const float2*>(lds + SP_PART)[(2 * pi + 1) * SP_ROWS + row];      // second half (odd wave)
            const int grow = r0 + row;
            if (grow < Mv) reinterpret_cast<float2*>(g.stats)[(size_t)(3 * pi + 1) * g.m_stride + grow] = make_float2(a.x + b.x, a.y + b.y);
        }
    }
}

bool split_panel_eligible(const SplitPanelArgs& g, int dtype) {
    auto al = [](const void* p) { return p != nullptr && ((uintptr_t)p & 15) == 0; };
    return (dtype == BG_BF16 || dtype == BG_F16) && (g.K == 768 || g.K == 1024) && al(g.a) && al(g.wf) && al(g.bias) && al(g.xh) && al(g.xl) &&
           al(g.stats) && g.M > 0 && g.m_stride >= g.M && (size_t)g.M * 2048 < 0xffffffffull;
}

int split_panel(const SplitPanelArgs& g, int dtype, hipStream_t s, double rows_hint) {
    if (g.M <= 0) return 0;
    const int panels = (g.M + SP_ROWS - 1) / SP_ROWS;
    const int grid = panels < 256 ? panels : 256;
    const double rows = rows_hint > 0 ? rows_hint : (double)g.M;
    ProfScope ps(PK_SPLIT_PANEL, 2.0 * rows * (double)SP_D * g.K, rows * (g.K * 2.0 + SP_D * 2.0 * 4 + 12 * 8.0) + 2.0 * SP_D * g.K + SP_D * 4.0, s);
    const bool f16 = dtype == BG_F16;
    if (g.K == 768) {
        if (f16) hipLaunchKernelGGL((split_panel_kernel<true, 768>), dim3(grid), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((split_panel_kernel<false, 768>), dim3(grid), dim3(512), 0, s, g);
    } else {
        if (f16) hipLaunchKernelGGL((split_panel_kernel<true, 1024>), dim3(grid), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((split_panel_kernel<false, 1024>), dim3(grid), dim3(512), 0, s, g);
    }
    return launch_status("split_panel");
}

}  // namespace bg

extern "C" int bg_split_panel_fwd(const void* a, int K, const void* w_frag, const float* bias, void* x_hi, void* x_lo, float* stats, int M,
                                  int m_stride, const int* m_dev, int dtype, bg_stream_t stream) {
    bg::SplitPanelArgs g{a, w_frag, bias, x_hi, x_lo, stats, M, m_stride, K, m_dev};
    BG_REQUIRE(M >= 0, BG_E_ARG, "bg_split_panel_fwd: negative M");
    if (M == 0) return 0;
    BG_REQUIRE(bg::split_panel_eligible(g, dtype), BG_E_ARG,
               "bg_split_panel_fwd: 16-bit operands, K = 768 or 1024, 16-byte aligned non-null pointers, m_stride >= M, M * 2048 < 2^32");
    return bg::split_panel(g, dtype, (hipStream_t)stream, 0.0);
}
