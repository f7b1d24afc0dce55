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
constexpr int FF_XROW = FF_D * 2, FF_HROW = FF_H * 2;      // LDS row pitches (bytes): x panel, hidden panel
constexpr int FF_PANEL = FF_ROWS * FF_HROW;                                      // 128 KiB: the hidden panel is the largest resident ([0, 96K): the x panel)
constexpr int FF_PROW = 96 * 4, FF_PATCH_W = 8 * FF_PROW;                        // epilogue-2 patch of a wave: 8 rows x 96 fp32 columns = 3 KiB
constexpr int FF_PATCH = FF_PANEL, FF_CONST = FF_PATCH + 8 * FF_PATCH_W;         // [128K, 152K) patches, [152K, 156K) b1' (fp32)
constexpr int FF_PART = FF_CONST + FF_H * 4;                                     // [156K, 160K) half-group statistics partials: 8 waves x 64 rows x (sum, sum of squares)
constexpr int FF_LDS = FF_PART + 8 * FF_ROWS * 8;                                // 160 KiB


__device__ __forceinline__ int ff_opaque(int x) {      // a value the compiler must treat as unknown here (see the panel loop)
    asm volatile("" : "+v"(x));
    return x;
}

template <bool F16>
__global__ __launch_bounds__(512) void ffn_fused_kernel(FfnArgs g) {
    using E = Elem<F16>;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[FF_LDS];
    unsigned char* const cst = lds + FF_CONST;                     // b1' [1024] fp32: staged once per workgroup

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Mv = g.m_dev ? *g.m_dev : g.M;
    const int n_panel = (Mv + FF_ROWS - 1) / FF_ROWS;
    if ((int)blockIdx.x >= n_panel) return;                        // uniform per workgroup, before any barrier

    const unsigned char* xh_b = reinterpret_cast<const unsigned char*>(g.xh);
    // this wave's weight streams: wave-uniform bases (scalar registers) + the lane's 16 bytes; k-slice s at + s * 4 KiB / 3 KiB
    const unsigned char* w1b = reinterpret_cast<const unsigned char*>(g.w1f) + (size_t)wave * (48 * 4096);
    const unsigned char* w2b = reinterpret_cast<const unsigned char*>(g.w2f) + (size_t)wave * (64 * 3072);
    unsigned char* const patch = lds + FF_PATCH + wave * FF_PATCH_W;
    // epilogue 2: the wave's 96 output columns are one whole 64-column statistics group and one half of the group it shares with its
    // neighbour (even wave: columns 0-63 whole, 64-95 = first half of the shared group; odd wave: 0-31 = its second half, 32-95 whole)
    const int odd = wave & 1;
    const int grp_full = 3 * (wave >> 1) + 2 * odd;               // (the shared group is 3 * (wave >> 1) + 1)

    if (threadIdx.x < 256) reinterpret_cast<float4*>(cst)[threadIdx.x] = reinterpret_cast<const float4*>(g.b1)[threadIdx.x];

    int n_stamp = 0;
    auto stamp = [&]() {
        if (g.stamps && blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 32) g.stamps[n_stamp] = (long long)__builtin_amdgcn_s_memtime();
        ++n_stamp;
    };

    // the x panel of a row block on its way to the LDS: 64 rows x 1536 B = 12 x 16 B per thread, rows past the end clamped (never
    // stored).  Requested one panel AHEAD (during epilogue 2 of the previous panel), parked in registers, written after the barrier
    // that retires the hidden panel.
    // (twelve named registers, not an array: hipcc leaves a 192-byte array that lives across the panel loop in SCRATCH, i.e. waits for
    //  every load right where it is issued in order to store it again)
    uint4 a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11;
#define FF_AREGS(X) X(0, a0) X(1, a1) X(2, a2) X(3, a3) X(4, a4) X(5, a5) X(6, a6) X(7, a7) X(8, a8) X(9, a9) X(10, a10) X(11, a11)
#define FF_REQ(it, reg)                                                                                         \
    {                                                                                                           \
        const int c = it * 512 + tid_;               /* 16-byte chunk of the panel: row c / 96, chunk c % 96 */ \
        int grow = r0_ + c / 96;                                                                                \
        grow = grow < Mv ? grow : Mv - 1;                                                                       \
        reg = *reinterpret_cast<const uint4*>(xh_b + ((unsigned)grow * FF_XROW + (unsigned)(c % 96) * 16u)); /* 32-bit offset on a uniform base */                   \
    }
#define FF_STO(it, reg)                                                                                         \
    {                                                                                                           \
        const int c = it * 512 + tid_;                                                                          \
        const int row = c / 96, ch = c % 96;                                                                    \
        *reinterpret_cast<uint4*>(lds + row * FF_XROW + ((ch ^ (row & 15)) << 4)) = reg;                        \
    }
#define FF_PANEL_REQUEST(r0v)                                  \
    {                                                          \
        const int tid_ = ff_opaque(threadIdx.x), r0_ = (r0v);  \
        FF_AREGS(FF_REQ)                                       \
        __builtin_amdgcn_sched_barrier(0);                     \
    }
#define FF_PANEL_STORE()                                       \
    {                                                          \
        const int tid_ = ff_opaque(threadIdx.x);               \
        FF_AREGS(FF_STO)                                       \
    }
    // (rstd, -mean rstd) of the lane's two rows (32 i + lane & 31) from the twelve partials, in the association order of every other kernel
    auto coeffs = [&](int r0, float2 (&cf)[2]) {
        const int l31 = ff_opaque(threadIdx.x) & 31;
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
    };

    float2 cf[2];
    FF_PANEL_REQUEST(blockIdx.x * FF_ROWS)
    coeffs(blockIdx.x * FF_ROWS, cf);

    for (int p = blockIdx.x; p < n_panel; p += gridDim.x) {
        const int r0 = p * FF_ROWS;
        const bool has_next = p + (int)gridDim.x < n_panel;
        const int r0n = has_next ? (p + (int)gridDim.x) * FF_ROWS : r0;   // (no next panel: this one again -- loads whose results are never used.
        //  Unconditional on purpose: behind a branch the OLD registers would stay live through both K loops, as the value of the other path)
        stamp();
        // Everything lane-dependent is derived from an OPAQUE thread id inside the panel loop: hoisted out of it, these addresses
        // would be live across both K loops, which run near the register limit, and be spilled (hipcc answers every scratch reload
        // with s_waitcnt vmcnt(0): the weight stream's loads in flight would be drained).
        const int ln = ff_opaque(threadIdx.x) & 63, l31 = ln & 31, hq = ln >> 5;
        // shared-operand fragment addresses: row 32 i + l31, 16-byte chunk 2 s + hq, XOR-swizzled by the row's low four bits (both
        // pitches are multiples of 256 B: without it the 16 lanes of a ds_read_b128 group would all hit the same banks)
        const unsigned sw = (unsigned)(l31 & 15);
        const unsigned lane16 = (unsigned)ln * 16u;
        FF_PANEL_STORE()
        __syncthreads();
        stamp();

        // ---- phase 1: h = x W1'^T, wave w: hidden columns 128 w .. + 127 ----
        f32x16 acc1[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
        {
            // software pipeline: weight fragments four k-slices ahead (global -> registers), x fragments one slice ahead (LDS)
            V8 wf[4][4], xf[2][2];
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
            ldw(0, 0); ldw(1, 1); ldw(2, 2); ldw(3, 3); ldx(0, 0);
#pragma unroll 1
            for (int s = 0; s < 48; s += 4) {
                ldx(1, s + 1); mm(0, 0); ldw(0, s + 4);
                ldx(0, s + 2); mm(1, 1); ldw(1, s + 5);
                ldx(1, s + 3); mm(2, 0); ldw(2, s + 6);
                ldx(0, s + 4); mm(3, 1); ldw(3, s + 7);
            }
        }
        stamp();
        // the lane's 64 column sums of W1' travel while the workgroup gathers at the barrier (the weight registers are free now)
        float4 csum[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) csum[j][q] = *reinterpret_cast<const float4*>(g.colsum1 + wave * 128 + j * 32 + 8 * q + 4 * hq);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                            // every wave is done with the x panel: h may overwrite it
        stamp();

        // ---- epilogue 1: LayerNorm fold + bias + ReLU -> 16 bits -> the hidden panel (row-major, same swizzle) ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wave * 128 + j * 32 + 8 * q + 4 * hq;
                const float4 b = *reinterpret_cast<const float4*>(cst + col * 4);
                const float4 c = csum[j][q];
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
        coeffs(r0n, cf);                              // (the fold coefficients of this panel are consumed; the next panel's rows are untouched by this launch)
        __syncthreads();
        stamp();

        // ---- phase 2: y = h W2^T, wave w: output columns 96 w .. + 95 ----
        // Epilogue-2 bookkeeping first: the residual octets of its first three slabs are requested HERE, before the K loop (48
        // registers that loop can spare), so that they have landed when the loop ends; slab t then requests slab t + 3.
        unsigned char* const out_hi = reinterpret_cast<unsigned char*>(g.xh);     // (32-bit byte offsets on uniform bases: M * 1536 < 2^32)
        unsigned char* const out_lo = reinterpret_cast<unsigned char*>(g.xl);
        const int ln2 = ff_opaque(threadIdx.x) & 63;
        const int k8 = ln2 & 7, rF = ln2 >> 3;                      // whole group: lane = 8 row + octet
        const int k4 = ln2 & 3, rH = (ln2 >> 2) & 7;                // half group (lanes 0-31; 32-63 shadow them, nothing stored)
        const int colF = 96 * wave + 32 * odd + 8 * k8, colH = 96 * wave + 64 * (1 - odd) + 8 * k4;
        uint4 rb[4][4];                                             // residual octets of a slab: whole-group item (hi, lo), half-group item (hi, lo)
        auto res_request = [&](int b, int t) {
            int gF = r0 + 8 * t + rF, gH = r0 + 8 * t + rH;
            gF = gF < Mv ? gF : Mv - 1;
            gH = gH < Mv ? gH : Mv - 1;
            const unsigned oF = ((unsigned)gF * FF_D + (unsigned)colF) * 2u, oH = ((unsigned)gH * FF_D + (unsigned)colH) * 2u;
            rb[b][0] = *reinterpret_cast<const uint4*>(out_hi + oF);
            rb[b][1] = *reinterpret_cast<const uint4*>(out_lo + oF);
            rb[b][2] = *reinterpret_cast<const uint4*>(out_hi + oH);
            rb[b][3] = *reinterpret_cast<const uint4*>(out_lo + oH);
            __builtin_amdgcn_sched_barrier(0);
        };
        res_request(0, 0); res_request(1, 1); res_request(2, 2);

        f32x16 acc2[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
        {
            V8 wf[4][3], hf[2][2];                                  // weight fragments four k-slices ahead
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
#pragma unroll
            for (int b = 0; b < 4; ++b) ldw(b, b);
            ldh(0, 0);
#pragma unroll 1
            for (int s = 0; s < 64; s += 4) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    ldh((b + 1) & 1, s + b + 1); mm(b, b & 1); ldw(b, s + b + 4);
                }
            }
        }
        stamp();
        // (no barrier: the epilogue below works out of the wave's private patch; the hidden panel is retired by the barrier at its end)

        // ---- epilogue 2: split residual + statistics, wave-private, eight slabs of 8 rows x 96 fp32 columns ----
        {
            const unsigned chF = (unsigned)(8 * odd + 2 * k8), chH = (unsigned)(16 * (1 - odd) + 2 * k4);      // first 16-byte chunk in the patch row
            const float4 bF0 = *reinterpret_cast<const float4*>(g.b2 + colF), bF1 = *reinterpret_cast<const float4*>(g.b2 + colF + 4);
            const float4 bH0 = *reinterpret_cast<const float4*>(g.b2 + colH), bH1 = *reinterpret_cast<const float4*>(g.b2 + colH + 4);
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
                    const unsigned o = ((unsigned)grow * FF_D + (unsigned)col) * 2u;
                    *reinterpret_cast<uint4*>(out_hi + o) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                    *reinterpret_cast<uint4*>(out_lo + o) = make_uint4(la.x, la.y, lb.x, lb.y);
                }
            };
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int i = t >> 2;
                // the next x panel travels while the second half of this one's rows is finished (requested here, not earlier: the
                // registers of the first row tile's accumulators are free now)
                if (t == 4) FF_PANEL_REQUEST(r0n)
                if ((l31 >> 3) == (t & 3)) {
                    const unsigned prow = (unsigned)(l31 & 7);
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned c = (unsigned)(8 * j + 2 * q + hq);
                            *reinterpret_cast<float4*>(patch + prow * FF_PROW + ((c ^ prow) << 4)) =
                                make_float4(acc2[i][j][4 * q], acc2[i][j][4 * q + 1], acc2[i][j][4 * q + 2], acc2[i][j][4 * q + 3]);
                        }
                }
                __builtin_amdgcn_wave_barrier();                  // LDS executes a wave's accesses in order: no wait needed
                if (t + 3 < 8) res_request((t + 3) & 3, t + 3);
                {
                    const int grow = r0 + 8 * t + rF;
                    float s8, q8;
                    item(patch + rF * FF_PROW, chF, (unsigned)rF, bF0, bF1, rb[t & 3][0], rb[t & 3][1], grow, colF, grow < Mv, s8, q8);
                    const float S = group8_sum(s8), Q = group8_sum(q8);
                    if (grow < Mv && k8 == 0) reinterpret_cast<float2*>(g.stats)[(size_t)grp_full * g.m_stride + grow] = make_float2(S, Q);
                }
                {
                    const int grow = r0 + 8 * t + rH;
                    float s8, q8;
                    item(patch + rH * FF_PROW, chH, (unsigned)rH, bH0, bH1, rb[t & 3][2], rb[t & 3][3], grow, colH, grow < Mv && ln2 < 32, s8, q8);
                    s8 += dpp_mov<0xB1>(s8); s8 += dpp_mov<0x4E>(s8);        // the quad's four octets: (o0 + o1) + (o2 + o3)
                    q8 += dpp_mov<0xB1>(q8); q8 += dpp_mov<0x4E>(q8);
                    // the shared group's statistics = this wave's half + the neighbour's: parked here, merged after the barrier
                    if (ln2 < 32 && k4 == 0) reinterpret_cast<float2*>(lds + FF_PART)[wave * FF_ROWS + 8 * t + rH] = make_float2(s8, q8);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp();
        __syncthreads();                                            // hidden panel and patches retired; every half-group partial is in place
        if (threadIdx.x < 256) {
            const int pi = threadIdx.x >> 6, row = threadIdx.x & 63;
            const float2 a = reinterpret_cast<const float2*>(lds + FF_PART)[(2 * pi) * FF_ROWS + row];          // first half  (even wave)
            const float2 b = reinterpret_cast<const float2*>(lds + FF_PART)[(2 * pi + 1) * FF_ROWS + row];      // second half (odd wave)
            const int grow = r0 + row;
            if (grow < Mv) reinterpret_cast<float2*>(g.stats)[(size_t)(3 * pi + 1) * g.m_stride + grow] = make_float2(a.x + b.x, a.y + b.y);
        }
        stamp();
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
    FfnArgs a = g;
    a.stamps = reinterpret_cast<long long*>(((uintptr_t)(unsigned)g_tune[TUNE_DEBUG_PTR_HI] << 32) | (uintptr_t)(unsigned)g_tune[TUNE_DEBUG_PTR_LO]);
    if (dtype == BG_F16) hipLaunchKernelGGL((ffn_fused_kernel<true>), dim3(grid), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((ffn_fused_kernel<false>), dim3(grid), dim3(512), 0, s, a);
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
