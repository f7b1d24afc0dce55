// Residual-stream GEMMs of the encoder layers (out-proj and FFN2 with "+ residual", network.py:1076-1078 ->
// torch/nn/modules/transformer.py) in the 16-bit modes:
//
//   (hi, lo)[m,n] = split(sum_k a[m,k] w[n,k] + bias[n] + res_hi[m,n] + res_lo[m,n])      hi = T(v), lo = T(v - hi)
//   stats[n / 64][m] = (sum, sum of squares) of v over the 64-column group                  (LayerNorm fold of the consumer)
//
// These launches move 8 B per output element (4 in, 4 out) next to 2 K FLOP: at K = 768 they are bound by the memory system, not by
// the matrix pipe.  The 128 x 128 persistent kernel (gemm_16bit.hip) ran them as K loop, epilogue, K loop, epilogue: while a
// workgroup is in its epilogue it issues no MFMAs, while it is in its K loop it moves nothing but operands, and with 1.4-3 tiles
// per workgroup at the face-LDM sizes the whole chip is in the same phase at the same time (profiles/r03/gemm128_*phase*.log: a
// split tile = 22 k cycles of K loop + 15 k of epilogue per wave).
//
// This kernel software-pipelines the two: the accumulators of tile t are handed to a second register set and their epilogue is
// executed slab by slab INSIDE the K loop of tile t + 1 -- the residual rows are requested in one K-step, and a K-step later
// (behind the wait + barrier every K-step has anyway) they are added, split, reduced to row statistics and stored; the loads'
// latency and the stores' acknowledgement hide behind MFMA work of both co-resident workgroups, and the memory traffic of the
// launch is spread over its whole duration instead of arriving in bursts.  Only the epilogue of a workgroup's LAST tile runs on
// its own.
//
// Geometry = the 128 x 128 persistent kernel: tile 128 x 128 x 64, 4 waves (2 x 2) of 64 x 64, 2-slot LDS-DMA ring over K with a
// 16-byte XOR swizzle, one barrier per K-step, two workgroups per CU (64 KiB ring + 4 x 4 KiB patches), XCD-aware tile walk.
// The product is computed TRANSPOSED (weights as the MFMA's A operand, as in gemm_p256.hip): a lane owns one output row and four
// consecutive columns per accumulator quad, so a 16-row x 64-column slab goes to the wave's patch as eight ds_write_b128 and
// comes back as whole 8-column octets per lane -- the slab arithmetic (order of additions, statistics butterfly, layout of the
// statistics) is that of the other two kernels, bit for bit (tests/test_gpu_round4.py).
//
// K-steps of a tile, KT = K / 64 = 4 chunks of KT / 4 steps; chunk c serves slab c (rows 16 c .. 16 c + 15 of the wave's 64)
// of the PREVIOUS tile:
//   step 0 of the chunk   [A]  write the slab to the patch (its residual + bias were requested in the previous chunk's last step and
//                              stay in flight across this step: the step's wait is vmcnt(6), "my DMA pieces landed")
//   step 1                [B]  read the patch back, add, statistics, split, store -- at the START of the step, so the stores have
//                              the whole step to be acknowledged before the next step's vmcnt(0)
//   last step             [Z]  request the next slab's residual octets + bias (6 x 16-byte loads)
// Vector-memory ordering: every K-step starts with a wait for its own DMA pieces; the only counted one ([A]) has nothing but older
// loads in flight -- no counted wait ever has a store between a load and its use (loads and stores share vmcnt and retire out of
// order with respect to each other: profiles/r03/gemm_p256_split_counted_waits_WRONG.log).
#include "gemm16.h"

namespace bg {

constexpr int SP_STAGE = 32768, SP_RING = 2 * SP_STAGE, SP_PATCH = 4096;
typedef __attribute__((ext_vector_type(4))) unsigned sp_u32x4;

template <bool F16>
__global__ __launch_bounds__(256, 2) void gemm16_split_pipe_kernel(GemmArgs g, int m_panels) {
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[SP_RING + 4 * SP_PATCH];       // 80 KiB: two workgroups per CU

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, hq = lane >> 5, k8 = lane & 7, r8 = lane >> 3;
    const int nt_n = g.N_pad >> 7;
    const int G = gridDim.x;
    const int Mv = g.m_dev ? *g.m_dev : g.M;                      // rows present (compacted batch: device-side count)
    if (g.m_dev) m_panels = (Mv + 127) >> 7;
    // hybrid launches: the 256 x 256 kernel owns the row panels below p0 (bg_common.h p256_rows; both kernels read the same answer)
    const int p0 = g.hybrid ? (g.rows256_dev ? *g.rows256_dev : g.rows256_host) >> 7 : 0;
    // XCD-aware walk (block b runs on XCD b % 8, private 4 MiB L2 each): XCD x owns row panels x, x + 8, ...; its G / 8 workgroups
    // walk that sub-grid column-fastest, so the tiles an XCD runs at a time share a few A panels and keep W resident
    const int xcd = blockIdx.x & 7, w_local = blockIdx.x >> 3, cnt = G >> 3;                  // G % 8 == 0 (launcher)
    auto tile_at = [&](int t, int& tm0, int& tn0) -> bool {
        const int panel = p0 + xcd + (t / nt_n) * 8;
        tm0 = panel << 7;
        tn0 = (t % nt_n) << 7;
        return panel < m_panels;
    };

    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.a);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.w);
    const unsigned lda_b = (unsigned)g.lda * 2u, ldw_b = (unsigned)g.K * 2u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    unsigned char* patch = lds + SP_RING + wave * SP_PATCH;

    // ---- LDS-DMA: wave w moves pieces 4 w .. 4 w + 3 (8 rows x 128 B each) of the A and of the W half of a ring slot.  Source =
    // wave-uniform base (SGPR pair: tile origin + k offset) + per-lane byte offset; the 16-byte chunk index is XOR-swizzled on
    // the SOURCE side (the DMA writes lane-linear), the fragment reads apply the same involution ----
    auto dma = [&](unsigned dst, const unsigned char* src, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    unsigned va[4], vw[4];                                        // per-lane source offsets of the four A / W pieces
    auto piece_row = [&](int j) { return (wave * 4 + j) * 8 + (lane >> 3); };
    auto piece_chunk = [&](int j) { return (unsigned)(((lane & 7) ^ ((piece_row(j) >> 1) & 7)) * 16); };
#pragma unroll
    for (int j = 0; j < 4; ++j) vw[j] = (unsigned)piece_row(j) * ldw_b + piece_chunk(j);
    auto a_offsets = [&](int m0t) {
        const int last = Mv - 1 - m0t;                            // rows >= Mv are clamped (never stored)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int row = piece_row(j);
            row = row < last ? row : last;
            va[j] = (unsigned)row * lda_b + piece_chunk(j);
        }
    };
    auto issue = [&](int slot, const unsigned char* a_src, const unsigned char* w_src) {
        const unsigned base = lds0 + (unsigned)(slot * SP_STAGE + wave * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(base + (unsigned)(j * 1024), a_src, va[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(base + (unsigned)(16384 + j * 1024), w_src, vw[j]);
    };

    // ---- fragment reads inside a ring slot (A rows first, W rows at +16 KiB) ----
    unsigned xk[4];
    {
        const int sw = (l31 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xk[ks] = (unsigned)(((ks * 2 + hq) ^ sw) << 4);
    }
    const unsigned a_rd = (unsigned)(wm * 64 + l31) * 128u;       // + i * 4096
    const unsigned b_rd = 16384u + (unsigned)(wn * 64 + l31) * 128u;

    const T* res_hi = reinterpret_cast<const T*>(g.res_hi);
    const T* res_lo = reinterpret_cast<const T*>(g.res_lo);
    T* out_hi = reinterpret_cast<T*>(g.out);
    T* out_lo = reinterpret_cast<T*>(g.out_lo);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- the slab pipeline: slab c of a 64 x 64 wave block = rows 16 c .. 16 c + 15 (row tile c >> 1, half c & 1) ----
    // request: residual octets (hi, lo) of the lane's two rows (it = 0, 1: row 8 it + r8 of the slab, columns 8 k8 .. + 7) + bias
    auto slab_request = [&](int c, int rbase, int cbase, sp_u32x4 (&rb)[2][2], f32x4& b0, f32x4& b1) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int grow = rbase + c * 16 + it * 8 + r8;
            grow = grow < Mv ? grow : Mv - 1;
            const size_t o = (size_t)grow * g.ld_res + cbase + k8 * 8;
            // (plain cache policy: non-temporal loads / stores here are -12 % for out-proj at M = 138 752 in isolation, +4-7 % at the
            //  face-LDM sizes, +-1 % on a whole edge-net evaluation -- profiles/r04/gemm_split_bench_nt_*.log, edge_ab_nt_*.log)
            rb[it][0] = *reinterpret_cast<const sp_u32x4*>(res_hi + o);
            rb[it][1] = *reinterpret_cast<const sp_u32x4*>(res_lo + o);
        }
        b0 = *reinterpret_cast<const f32x4*>(g.bias + cbase + k8 * 8);      // (bias != null: launcher)
        b1 = *reinterpret_cast<const f32x4*>(g.bias + cbase + k8 * 8 + 4);
    };
    // accumulators -> patch: rows of 256 B (64 fp32 columns), 16-byte chunk XOR-swizzled by the row (ds_write_b128 and
    // ds_read_b128 conflict-free).  Transposed product: the lanes with (l31 >> 4) == half own the slab's 16 rows.
    auto slab_write = [&](int c, const f32x16 (&p)[2][2]) {
        const int i = c >> 1, half = c & 1;
        if ((l31 >> 4) == half) {
            const int prow = l31 & 15;
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
    // patch -> + bias + residual -> statistics -> (hi, lo): the arithmetic of gemm_16bit.hip / gemm_p256.hip, in their order
    auto slab_finish = [&](int c, int rbase, int cbase, const sp_u32x4 (&rb)[2][2], f32x4 b0, f32x4 b1) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int prow = it * 8 + r8;
            const float4 p0 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8) ^ prow) << 4));
            const float4 p1 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8 + 1) ^ prow) << 4));
            float v[8] = {p0.x + b0[0], p0.y + b0[1], p0.z + b0[2], p0.w + b0[3], p1.x + b1[0], p1.y + b1[1], p1.z + b1[2], p1.w + b1[3]};
            const sp_u32x4 h4 = rb[it][0], l4 = rb[it][1];
            float fh[4], fl[4];
            unpack4_16<F16>(make_uint2(h4[0], h4[1]), fh);
            unpack4_16<F16>(make_uint2(l4[0], l4[1]), fl);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
            unpack4_16<F16>(make_uint2(h4[2], h4[3]), fh);
            unpack4_16<F16>(make_uint2(l4[2], l4[3]), fl);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
            const int grow = rbase + c * 16 + prow;
            const bool row_ok = grow < Mv;                        // (in place: a clamped duplicate row must not be written)
            const float s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            const float q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                             ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
            const float S = group8_sum(s8), Q = group8_sum(q8);
            // part-major [N / 64][M] pairs: the 8 rows of this lane group are 64 contiguous bytes
            if (row_ok && k8 == 0)
                reinterpret_cast<float2*>(g.stats_out)[(size_t)(cbase >> 6) * g.M + grow] = make_float2(S, Q);
            if (row_ok) {
                const float va4[4] = {v[0], v[1], v[2], v[3]}, vb4[4] = {v[4], v[5], v[6], v[7]};
                uint2 ha, la, hb, lb;
                split4_16<F16>(va4, ha, la);
                split4_16<F16>(vb4, hb, lb);
                const size_t o = (size_t)grow * g.ldc + cbase + k8 * 8;
                const sp_u32x4 sh = {ha.x, ha.y, hb.x, hb.y}, sl = {la.x, la.y, lb.x, lb.y};
                *reinterpret_cast<sp_u32x4*>(out_hi + o) = sh;
                *reinterpret_cast<sp_u32x4*>(out_lo + o) = sl;
            }
        }
    };

    int m0, n0, nm0 = 0, nn0 = 0;                                 // the tile being computed / the workgroup's next tile
    if (!tile_at(w_local, m0, n0)) return;                        // uniform per workgroup, before any barrier
    const int KT = g.K / G_BK, CH = KT >> 2;                      // K % 256 == 0, K >= 768 (launcher): CH >= 3
    const unsigned char* a_cur = Ab + (size_t)m0 * lda_b;
    const unsigned char* w_cur = Wb + (size_t)n0 * ldw_b;
    const unsigned char* a_nxt = a_cur;
    const unsigned char* w_nxt = w_cur;
    bool has_next = false;
    a_offsets(m0);
    issue(0, a_cur, w_cur);
    int slot = 0;

    f32x16 acc[2][2], prev[2][2];                                 // `prev`: a finished tile whose epilogue is still to run
    int p_rbase = 0, p_cbase = 0;                                 // this wave's block of that tile
    sp_u32x4 rb[2][2] = {};                                       // residual octets of the slab in flight
    f32x4 bs0 = zero4, bs1 = zero4;

    // One tile's K loop.  PEND (compile time: the workgroup's first tile has nothing pending, and a run-time test would make the
    // compiler merge the "loads consumed" and "loads in flight" states at every use): the K-steps carry the epilogue of `prev`.
    //
    // Roles of a chunk's K-steps (slab C of `prev`):  [A] write the slab to the patch   [B] finish it (the residual landed long ago)
    //                                                 [Z] request the NEXT slab's residual (+ bias)
    // Every step starts with a wait for its own DMA pieces.  That wait is vmcnt(0) except in [A]: there the only operations in flight
    // are the previous step's 8 DMA pieces and, younger, the 6 loads of its [Z] request -- loads retire in order, so vmcnt(6) says
    // "the DMA landed" and lets the residual travel through a second K-step.  (With vmcnt(0) everywhere the residual's miss latency
    // -- Infinity Cache / HBM, longer than the DMA's L2 hits -- sat on the critical path of four K-steps per tile, and the pipelined
    // kernel was exactly as fast as the serial one: profiles/r04/gemm_split_bench_first_run.log.)  A counted wait never has a store
    // between a load and its use: [B]'s stores are drained by the vmcnt(0) of the step that follows.
    // The tile's LAST step requests slab 0 of the tile itself (it becomes `prev`); the workgroup's last tile is served after the loop.
    auto tile_body = [&](auto pend_c) {
        constexpr bool PEND = decltype(pend_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        int kt = 0;
        // ROLE 0: plain; 1: [A]; 2: [B]; 3: [Z] (request slab C + 1).  `last`: the tile's final K-step (stages the next tile; its [Z]
        // request is for the tile itself)
        auto kstep = [&](auto role_c, auto slab_c, auto last_c) {
            constexpr int C = decltype(slab_c)::value;
            constexpr bool last = decltype(last_c)::value;
            constexpr int ROLE = (PEND || last) ? decltype(role_c)::value : 0;
            if (ROLE == 1) wait_vmcnt<6>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my fragment (and patch) reads of the previous K-step are complete
            __builtin_amdgcn_s_barrier();
            if (ROLE == 2) {
                // the slab's residual octets and bias were requested two K-steps ago and are in their registers now; this names
                // them behind the wait above, so the compiler's own wait for them (which cannot see that wait) lands here, where
                // nothing is in flight -- not behind the DMA issued below
                asm volatile("" : "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(bs0), "+v"(bs1) :: "memory");
            }
            if (!last) {
                issue(slot ^ 1, a_cur + (unsigned)(kt + 1) * (2 * G_BK), w_cur + (unsigned)(kt + 1) * (2 * G_BK));
            } else if (has_next) {                                // keep the DMA stream running across the tile seam
                a_offsets(nm0);
                issue(slot ^ 1, a_nxt, w_nxt);
            }
            if (ROLE == 3) {
                if (!last) {
                    slab_request(C + 1, p_rbase, p_cbase, rb, bs0, bs1);
                } else if (has_next) {                            // this tile becomes `prev`: its slab 0
                    slab_request(0, m0 + wm * 64, n0 + wn * 64, rb, bs0, bs1);
                }
            }
            if (ROLE == 2) slab_finish(C, p_rbase, p_cbase, rb, bs0, bs1);      // early in the step: its stores get the whole step to be acknowledged
            const unsigned char* st = lds + slot * SP_STAGE;
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
                    for (int j = 0; j < 2; ++j)                   // transposed product: lane = output row, registers = columns
                        acc[i][j] = E::mfma(bf[ks & 1][j], af[ks & 1][i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ROLE == 1) slab_write(C, prev);
            __builtin_amdgcn_sched_barrier(0);
            slot ^= 1;
            ++kt;
        };
        using R0 = std::integral_constant<int, 0>;
        using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>;
        using R3 = std::integral_constant<int, 3>;
        auto chunk = [&](auto slab_c) {
            constexpr int C = decltype(slab_c)::value;
            kstep(R1{}, slab_c, std::false_type{});
            kstep(R2{}, slab_c, std::false_type{});
            for (int r = 2; r + 1 < CH; ++r) kstep(R0{}, slab_c, std::false_type{});
            kstep(R3{}, slab_c, std::integral_constant<bool, C == 3>{});
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
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

    // ---- the last tile's epilogue has no K loop to hide in: all four slabs' residuals in flight at once, then slab by slab ----
    {
        sp_u32x4 rr[4][2][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) slab_request(c, p_rbase, p_cbase, rr[c], bs0, bs1);
        slab_write(0, prev);
        wait_vmcnt<0>();                                          // before the first store: no store between a load and its use
#pragma unroll
        for (int c = 0; c < 4; ++c)
            asm volatile("" : "+v"(rr[c][0][0]), "+v"(rr[c][0][1]), "+v"(rr[c][1][0]), "+v"(rr[c][1][1]) :: "memory");
        asm volatile("" : "+v"(bs0), "+v"(bs1) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c > 0) slab_write(c, prev);
            __builtin_amdgcn_wave_barrier();                      // LDS executes a wave's accesses in order: no wait needed
            slab_finish(c, p_rbase, p_cbase, rr[c], bs0, bs1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// shape / argument checks: the residual-stream form of the encoder layers (split residual in, (hi, lo) + statistics out, no
// activation, no broadcast addends, no row map, K a multiple of four K-steps)
bool split_pipe_eligible(const GemmArgs& g) {
    if (!(g.out_lo && g.res_hi && g.res_lo && g.stats_out && g.bias) || g.stats_in || g.add || g.add2 || g.row_map || g.act != BG_ACT_NONE ||
        g.cv_C > 0 || g.out_dtype == BG_F32)
        return false;
    if (g.N != g.N_pad || g.N_pad % 128 != 0 || g.K % (4 * G_BK) != 0 || g.K < 12 * G_BK || g.ldc % 8 != 0 || g.ld_res % 8 != 0)
        return false;
    const long long nt = (long long)((g.M + 127) / 128) * (g.N_pad / 128);
    return nt >= 64 && (size_t)127 * g.lda * 2 + 128 < 0xffffffffull && (size_t)127 * g.K * 2 + 128 < 0xffffffffull;
}

template <bool F16>
int launch_split_pipe(const GemmArgs& g, hipStream_t s) {
    const int m128 = (g.M + 127) / 128, nt = m128 * (g.N_pad / 128);
    const int grid = nt < 512 ? (nt & ~7) : 512;                  // two resident workgroups per CU, a multiple of 8 (XCD walk)
    hipLaunchKernelGGL((gemm16_split_pipe_kernel<F16>), dim3(grid), dim3(256), 0, s, g, m128);
    return launch_status("gemm16_split_pipe");
}
template int launch_split_pipe<false>(const GemmArgs&, hipStream_t);
template int launch_split_pipe<true>(const GemmArgs&, hipStream_t);

}  // namespace bg
