// Residual-stream GEMMs of the encoder layers (out-proj / FFN2 + residual, network.py:1076-1078) -- round 5: the software-pipelined
// epilogue of gemm_split.hip on a 256 x 128 tile with a THREE-slot LDS-DMA ring.
//
//   (hi, lo)[m,n] = split(sum_k a[m,k] w[n,k] + bias[n] + res_hi[m,n] + res_lo[m,n]),   stats[n / 64][m] = (sum, sum of squares)
//
// What round 4 measured: the 128 x 128 kernel's K loop is a DMA round trip per K-step (2-slot ring: vmcnt(0) + barrier, every stage
// has exactly one K-step to land), and seven organisations of its epilogue were within 3 % of each other.  What round 5's probe added
// (profiles/r05/gemm_ring_depth_probe_generic_kernel.log, the lock-step generic kernel at K = 4096): a third slot does nothing for a
// workgroup with ONE wave per SIMD (128 x 128, 4 waves: 868 -> 786 TF) and a lot for one with two (256 x 128, 8 waves: 735 -> 940 TF).
// Hence this geometry:
//   tile 256 x 128 x 64, 8 waves as 4 (rows) x 2 (columns) of 64 x 64, one persistent workgroup per CU (ring 3 x 48 KiB + 8 x 2 KiB
//   patches = 160 KiB), stages t + 1 and t + 2 in flight while stage t is multiplied (96 KiB per CU behind a COUNTED wait), 384 B of
//   operand DMA per MFMA (128 x 128: 512), transposed product (a lane owns one output row).
// Epilogue of tile t inside the K loop of tile t + 1, in slabs of 8 rows x 64 columns per wave (8 per tile, 2 KiB patch):
//   step s:   [B] finish slab s - 2 (patch -> + bias + residual -> statistics -> (hi, lo) stores)      s = 2 .. 9
//             [Z] request the residual octets of slab s (2 x 16-byte loads per lane; + the bias at s = 0)   s = 0 .. 7
//             DMA of stage s + 2, one piece at a time between the MFMAs; 16 MFMAs
//             [A] slab s - 1 -> patch                                                                  s = 1 .. 8
// The two waves of a SIMD (w, w + 4) take [B] at opposite ends of the step -- waves 0-3 before their MFMAs, waves 4-7 behind
// them -- so one wave's ~70 VALU instructions run under the other's matrix work instead of both idling the pipe together.
// Vector-memory discipline: every load of the loop is inline asm (LDS-DMA, and the residual / bias loads into registers: a
// compiler-visible load would make hipcc wait with vmcnt(0) at its first use and drain the two stages in flight), so the counted
// waits are exact in LOADS: the wait of step s is "stage s landed" = vmcnt(number of loads issued after it).  Loads retire in
// order among themselves; stores (the [B] steps') share the counter and may retire out of order with respect to loads -- a
// pending store can only make a counted wait wait LONGER (the count includes it), never pass early, because no wait ever counts
// a store as something that "may stay in flight" (round 3's wrong variant did exactly that).
// Bit-identical to gemm_split.hip / gemm_16bit.hip / gemm_p256.hip (same k order, same slab arithmetic; tests/test_gpu_round5.py).
#include "gemm16.h"

namespace bg {

constexpr int S3_STAGE = 49152, S3_WOFF = 32768, S3_RING = 3 * S3_STAGE, S3_PATCH = 2048, S3_LDS = S3_RING + 8 * S3_PATCH;
static_assert(S3_LDS == 163840, "one workgroup per CU: the whole 160 KiB");
typedef __attribute__((ext_vector_type(4))) unsigned s3_u32x4;

__device__ __forceinline__ int s3_opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// LO = false (MEASUREMENT ONLY, bg_tune key 12 = 4): a 16-bit-only residual stream -- the lo plane is neither read nor written (4 B of
// residual traffic per output element instead of 8); the results are NOT those of the library's residual-stream form
template <bool F16, bool LO = true>
__global__ __launch_bounds__(512) void gemm16_split3_kernel(GemmArgs g, int m_panels) {
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[S3_LDS];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bool late = wave >= 4;                                  // waves w and w + 4 share a SIMD
    const int l31 = lane & 31, hq = lane >> 5, k8 = lane & 7, r8 = lane >> 3;
    const int nt_n = g.N_pad >> 7;
    const int G = gridDim.x;
    const int Mv = g.m_dev ? *g.m_dev : g.M;                      // rows present (compacted batch: device-side count)
    if (g.m_dev) m_panels = (Mv + 255) >> 8;
    // hybrid launches: the 256 x 256 kernel owns the row panels below p0 (bg_common.h p256_rows; both kernels read the same answer)
    const int p0 = g.hybrid ? (g.rows256_dev ? *g.rows256_dev : g.rows256_host) >> 8 : 0;
    // XCD-aware walk: XCD x owns the 256-row panels x, x + 8, ...; its G / 8 workgroups walk that sub-grid column-fastest
    const int xcd = blockIdx.x & 7, w_local = blockIdx.x >> 3, cnt = G >> 3;                  // G % 8 == 0 (launcher)
    auto tile_at = [&](int t, int& tm0, int& tn0) -> bool {
        const int panel = p0 + xcd + (t / nt_n) * 8;
        tm0 = panel << 8;
        tn0 = (t % nt_n) << 7;
        return panel < m_panels;
    };

    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.a);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.w);
    const unsigned lda_b = (unsigned)g.lda * 2u, ldw_b = (unsigned)g.K * 2u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    unsigned char* patch = lds + S3_RING + wave * S3_PATCH;

    // ---- LDS-DMA: wave w moves A pieces 4 w .. 4 w + 3 and W pieces 2 w, 2 w + 1 of a stage (1 KiB = 8 rows x 128 B each) ----
    auto dma = [&](unsigned dst, const unsigned char* src, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    unsigned va[4], vw[2];
    auto chunk_of = [&](int ln, int row) { return (unsigned)(((ln & 7) ^ ((row >> 1) & 7)) * 16); };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        vw[j] = (unsigned)row * ldw_b + chunk_of(lane, row);
    }
    // (recomputed at every tile seam from a lane id that costs no live register (mbcnt): values kept alive across the K loop for this would be spilled, and
    //  hipcc follows a scratch reload with vmcnt(0) -- which drains the two stages in flight)
    auto a_offsets = [&](int m0t) {
        const int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)s3_opaque(0)));     // lane id, from nothing
        const int last = Mv - 1 - m0t;                            // rows >= Mv are clamped (never stored)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = (wave * 4 + j) * 8 + (ln >> 3);
            const unsigned ch = chunk_of(ln, row);
            row = row < last ? row : last;
            va[j] = (unsigned)row * lda_b + ch;
        }
    };
    // piece p of a stage: 0-3 activation rows, 4-5 weight rows
    auto issue_piece = [&](int p, int slot, const unsigned char* a_src, const unsigned char* w_src) {
        const unsigned base = lds0 + (unsigned)(slot * S3_STAGE);
        if (p < 4) dma(base + (unsigned)((wave * 4 + p) * 1024), a_src, va[p]);
        else dma(base + (unsigned)(S3_WOFF + (wave * 2 + (p - 4)) * 1024), w_src, vw[p - 4]);
    };

    // ---- fragment reads inside a ring slot (A rows first, W rows at + 32 KiB) ----
    unsigned xk[4];
    {
        const int sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xk[ks] = (unsigned)(((ks * 2 + hq) ^ sw) << 4);
    }
    const unsigned a_rd = (unsigned)(wm * 64 + l31) * 128u;       // + i * 4096
    const unsigned b_rd = (unsigned)S3_WOFF + (unsigned)(wn * 64 + l31) * 128u;

    const unsigned char* res_hi = reinterpret_cast<const unsigned char*>(g.res_hi);
    const unsigned char* res_lo = reinterpret_cast<const unsigned char*>(g.res_lo);
    const unsigned char* bias_b = reinterpret_cast<const unsigned char*>(g.bias);
    T* out_hi = reinterpret_cast<T*>(g.out);
    T* out_lo = reinterpret_cast<T*>(g.out_lo);

    // ---- the slab pipeline: slab c of a 64 x 64 wave block = rows 8 c .. 8 c + 7 (row tile c >> 2, eighth c & 3) ----
    // register loads the compiler does not see (file header); `off` is a byte offset below 4 GiB (launcher)
    auto load16 = [&](s3_u32x4& dst, const unsigned char* base, unsigned off) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base) : "memory");
    };
    auto slab_request = [&](int c, int rbase, int cbase, s3_u32x4 (&rb)[2]) {
        int grow = rbase + c * 8 + r8;
        grow = grow < Mv ? grow : Mv - 1;
        const unsigned off = ((unsigned)grow * (unsigned)g.ld_res + (unsigned)(cbase + k8 * 8)) * 2u;
        load16(rb[0], res_hi, off);
        if (LO) load16(rb[1], res_lo, off);
    };
    auto bias_request = [&](int cbase, s3_u32x4 (&bs)[2]) {
        const unsigned off = (unsigned)(cbase + k8 * 8) * 4u;
        load16(bs[0], bias_b, off);
        load16(bs[1], bias_b, off + 16u);
    };
    // accumulators -> patch: rows of 256 B (64 fp32 columns), 16-byte chunk XOR-swizzled by the row.  Transposed product: the lanes
    // with (l31 >> 3) == (c & 3) own the slab's 8 rows.
    auto slab_write = [&](int c, const f32x16 (&p)[2][2]) {
        const int i = c >> 2, sub = c & 3;
        if ((l31 >> 3) == sub) {
            const int prow = l31 & 7;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c16 = j * 8 + 2 * q + hq;
                    *reinterpret_cast<float4*>(patch + prow * 256 + ((c16 ^ prow) << 4)) =
                        make_float4(p[i][j][4 * q], p[i][j][4 * q + 1], p[i][j][4 * q + 2], p[i][j][4 * q + 3]);
                }
        }
    };
    // patch -> + bias + residual -> statistics -> (hi, lo): the arithmetic of gemm_split.hip / gemm_16bit.hip / gemm_p256.hip, in
    // their order; a lane owns row r8 of the slab, columns 8 k8 .. 8 k8 + 7
    auto slab_finish = [&](int c, int rbase, int cbase, const s3_u32x4 (&rb)[2], const s3_u32x4 (&bs)[2]) {
        const int prow = r8;
        const float4 p0 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8) ^ prow) << 4));
        const float4 p1 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8 + 1) ^ prow) << 4));
        float v[8] = {p0.x + __uint_as_float(bs[0][0]), p0.y + __uint_as_float(bs[0][1]), p0.z + __uint_as_float(bs[0][2]),
                      p0.w + __uint_as_float(bs[0][3]), p1.x + __uint_as_float(bs[1][0]), p1.y + __uint_as_float(bs[1][1]),
                      p1.z + __uint_as_float(bs[1][2]), p1.w + __uint_as_float(bs[1][3])};
        const s3_u32x4 h4 = rb[0], l4 = LO ? rb[1] : s3_u32x4{0u, 0u, 0u, 0u};
        float fh[4], fl[4];
        unpack4_16<F16>(make_uint2(h4[0], h4[1]), fh);
        unpack4_16<F16>(make_uint2(l4[0], l4[1]), fl);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
        unpack4_16<F16>(make_uint2(h4[2], h4[3]), fh);
        unpack4_16<F16>(make_uint2(l4[2], l4[3]), fl);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
        const int grow = rbase + c * 8 + prow;
        const bool row_ok = grow < Mv;                            // (in place: a clamped duplicate row must not be written)
        const float s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const float q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                         ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        const float S = group8_sum(s8), Q = group8_sum(q8);
        if (row_ok && k8 == 0)
            reinterpret_cast<float2*>(g.stats_out)[(size_t)(cbase >> 6) * g.M + grow] = make_float2(S, Q);
        if (row_ok) {
            const float va4[4] = {v[0], v[1], v[2], v[3]}, vb4[4] = {v[4], v[5], v[6], v[7]};
            uint2 ha, la, hb, lb;
            split4_16<F16>(va4, ha, la);
            split4_16<F16>(vb4, hb, lb);
            const size_t o = (size_t)grow * g.ldc + cbase + k8 * 8;
            const s3_u32x4 sh = {ha.x, ha.y, hb.x, hb.y}, sl = {la.x, la.y, lb.x, lb.y};
            *reinterpret_cast<s3_u32x4*>(out_hi + o) = sh;
            if (LO) *reinterpret_cast<s3_u32x4*>(out_lo + o) = sl;
        }
    };

    int m0, n0, nm0 = 0, nn0 = 0;                                 // the tile being computed / the workgroup's next tile
    if (!tile_at(w_local, m0, n0)) return;                        // uniform per workgroup, before any barrier
    const int KT = g.K / G_BK;                                    // >= 12 (launcher)
    const unsigned char* a_cur = Ab + (size_t)m0 * lda_b;
    const unsigned char* w_cur = Wb + (size_t)n0 * ldw_b;
    const unsigned char* a_nxt = a_cur;
    const unsigned char* w_nxt = w_cur;
    bool has_next = false;
    a_offsets(m0);
#pragma unroll
    for (int p = 0; p < 6; ++p) issue_piece(p, 0, a_cur, w_cur);
#pragma unroll
    for (int p = 0; p < 6; ++p) issue_piece(p, 1, a_cur + 2 * G_BK, w_cur + 2 * G_BK);
    int slot = 0;                                                 // ring slot of the stage being multiplied

    f32x16 acc[2][2], prev[2][2];                                 // `prev`: a finished tile whose epilogue is still to run
    int p_rbase = 0, p_cbase = 0;                                 // this wave's block of that tile
    s3_u32x4 rb[3][2] = {};                                       // residual octets of the slabs in flight (slab c -> rb[c % 3])
    s3_u32x4 bs[2] = {};                                          // bias of prev's columns

    // One K-step.  S: position in the tile's role schedule (PEND tiles: 0 .. 9 carry the roles of the file header, >= 10 none);
    // SEAM: 0 = stage kt + 2 of this tile is staged, 1 / 2 = the tile's last two steps: stages 0 / 1 of the NEXT tile (if there is one).
    // The wait: stage kt was issued two steps ago, behind that step's [Z] loads; younger loads = the previous step's [Z] (2, or 4 with
    // the bias at S = 1) + its 6 DMA pieces (none when the previous step was SEAM 1 without a next tile).
    int kt = 0;
    auto kstep = [&](auto s_c, auto pend_c, auto seam_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr bool PEND = decltype(pend_c)::value;
        constexpr int SEAM = decltype(seam_c)::value;
        constexpr int ZL = LO ? 2 : 1;                              // loads per slab request
        constexpr int NZ = !PEND ? 0 : (S == 1 ? ZL + 2 : ((S >= 2 && S <= 8) ? ZL : 0));
        if (SEAM == 2) { if (has_next) wait_vmcnt<6 + NZ>(); else wait_vmcnt<NZ>(); }
        else wait_vmcnt<6 + NZ>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my fragment (and patch) reads of the previous K-step are complete
        __builtin_amdgcn_s_barrier();
        constexpr bool ROLE_B = PEND && S >= 2 && S <= 9, ROLE_Z = PEND && S <= 7, ROLE_A = PEND && S >= 1 && S <= 8;
        if (ROLE_B) {
            // slab S - 2's residual (requested two steps ago, OLDER than the stage waited for above) and the bias have landed: name
            // them behind the wait so that no use is scheduled ahead of it
            constexpr int R = (S + 1) % 3;                        // (S - 2) mod 3
            asm volatile("" : "+v"(rb[R][0]), "+v"(rb[R][1]), "+v"(bs[0]), "+v"(bs[1]) :: "memory");
            if (!late) slab_finish(S - 2, p_rbase, p_cbase, rb[R], bs);
        }
        if (ROLE_Z) {
            if (S == 0) bias_request(p_cbase, bs);
            slab_request(S, p_rbase, p_cbase, rb[S % 3]);
        }
        const int ns = slot + 2 >= 3 ? slot - 1 : slot + 2;       // slot of stage kt + 2 (the one read in step kt - 1: free since the barrier)
        const unsigned char* a_src = SEAM == 0 ? a_cur + (unsigned)(kt + 2) * (2 * G_BK) : (SEAM == 1 ? a_nxt : a_nxt + 2 * G_BK);
        const unsigned char* w_src = SEAM == 0 ? w_cur + (unsigned)(kt + 2) * (2 * G_BK) : (SEAM == 1 ? w_nxt : w_nxt + 2 * G_BK);
        const bool stage_more = SEAM == 0 || has_next;
        if (SEAM == 1 && has_next) a_offsets(nm0);                // (this tile's last stage was issued a step ago)
        const unsigned char* st = lds + slot * S3_STAGE;
        V8 af[2][2], bf[2][2];
        auto load_frags = [&](int ks, int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[buf][i] = *reinterpret_cast<const V8*>(st + a_rd + i * 4096 + xk[ks]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[buf][j] = *reinterpret_cast<const V8*>(st + b_rd + j * 4096 + xk[ks]);
        };
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) load_frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {                     // transposed product: lane = output row, registers = columns
                    acc[i][j] = E::mfma(bf[ks & 1][j], af[ks & 1][i], acc[i][j]);
                    const int idx = (ks * 2 + i) * 2 + j;         // the 6 DMA pieces go out one at a time between the 16 MFMAs
                    const int lo = idx * 6 / 16, hi = (idx + 1) * 6 / 16;
                    if (hi > lo) {
                        if (stage_more) issue_piece(lo, ns, a_src, w_src);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ROLE_B) {
            constexpr int R = (S + 1) % 3;
            if (late) slab_finish(S - 2, p_rbase, p_cbase, rb[R], bs);
        }
        if (ROLE_A) slab_write(S - 1, prev);
        __builtin_amdgcn_sched_barrier(0);
        slot = slot == 2 ? 0 : slot + 1;
        ++kt;
    };
    auto tile_body = [&](auto pend_c) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        kt = 0;
        using Z0 = std::integral_constant<int, 0>;
        using Z1 = std::integral_constant<int, 1>;
        using Z2 = std::integral_constant<int, 2>;
        kstep(std::integral_constant<int, 0>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 1>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 2>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 3>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 4>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 5>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 6>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 7>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 8>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 9>{}, pend_c, Z0{});
        for (int r = 10; r + 2 < KT; ++r) kstep(std::integral_constant<int, 10>{}, pend_c, Z0{});
        kstep(std::integral_constant<int, 10>{}, pend_c, Z1{});
        kstep(std::integral_constant<int, 10>{}, pend_c, Z2{});
    };
    auto hand_over = [&]() {                                      // the finished tile goes to the epilogue pipeline
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) prev[i][j] = acc[i][j];
        p_rbase = m0 + wm * 64;
        p_cbase = n0 + wn * 64;
    };
    auto look_ahead = [&](int t) {
        has_next = tile_at(t + cnt, nm0, nn0);
        a_nxt = Ab + (size_t)nm0 * lda_b;
        w_nxt = Wb + (size_t)nn0 * ldw_b;
    };
    int t = w_local;
    look_ahead(t);
    tile_body(std::false_type{});
    hand_over();
    while (has_next) {
        t += cnt;
        m0 = nm0; n0 = nn0; a_cur = a_nxt; w_cur = w_nxt;
        look_ahead(t);
        tile_body(std::true_type{});
        hand_over();
    }

    // ---- the last tile's epilogue has no K loop to hide in: four slabs' residuals in flight at a time ----
    wait_vmcnt<0>();
    bias_request(p_cbase, bs);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        s3_u32x4 rr[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) slab_request(half * 4 + c, p_rbase, p_cbase, rr[c]);
        wait_vmcnt<0>();                                          // (also drains the previous half's stores: no store between a load and its use)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(rr[c][0]), "+v"(rr[c][1]) :: "memory");
        asm volatile("" : "+v"(bs[0]), "+v"(bs[1]) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            slab_write(half * 4 + c, prev);
            __builtin_amdgcn_wave_barrier();                      // LDS executes a wave's accesses in order: no wait needed
            slab_finish(half * 4 + c, p_rbase, p_cbase, rr[c], bs);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// shape / argument checks: the residual-stream form of the encoder layers, at least 12 K-steps (ten of them carry the epilogue roles),
// 32-bit byte offsets for the operand DMA and the residual loads
bool split3_eligible(const GemmArgs& g) {
    if (!(g.out_lo && g.res_hi && g.res_lo && g.stats_out && g.bias) || g.stats_in || g.add || g.add2 || g.row_map || g.act != BG_ACT_NONE ||
        g.cv_C > 0 || g.out_dtype == BG_F32)
        return false;
    if (g.N != g.N_pad || g.N_pad % 128 != 0 || g.K % G_BK != 0 || g.K < 12 * G_BK || g.ldc % 8 != 0 || g.ld_res % 8 != 0)
        return false;
    if ((reinterpret_cast<uintptr_t>(g.res_hi) | reinterpret_cast<uintptr_t>(g.res_lo) | reinterpret_cast<uintptr_t>(g.bias)) & 15) return false;
    return (size_t)255 * g.lda * 2 + 128 < 0xffffffffull && (size_t)127 * g.K * 2 + 128 < 0xffffffffull &&
           (size_t)g.M * g.ld_res * 2 < 0xffffffffull;
}

template <bool F16>
int launch_split3(const GemmArgs& g, hipStream_t s) {
    const int m256 = (g.M + 255) / 256, nt = m256 * (g.N_pad / 128);
    const int grid = nt < 256 ? ((nt + 7) & ~7) : 256;            // one resident workgroup per CU, a multiple of 8 (XCD walk)
    if (g_tune[TUNE_SPLIT_PIPE] == 4) hipLaunchKernelGGL((gemm16_split3_kernel<F16, false>), dim3(grid), dim3(512), 0, s, g, m256);
    else hipLaunchKernelGGL((gemm16_split3_kernel<F16>), dim3(grid), dim3(512), 0, s, g, m256);
    return launch_status("gemm16_split3");
}
template int launch_split3<false>(const GemmArgs&, hipStream_t);
template int launch_split3<true>(const GemmArgs&, hipStream_t);

}  // namespace bg
