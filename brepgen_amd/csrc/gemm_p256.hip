// 16-bit MFMA GEMM, 256 x 256 x 64 tile, persistent, 8-phase K loop -- the MFMA-bound Linears of the denoisers (QKV and FFN1 with
// the LayerNorm fold, network.py:1076-1078 -> torch/nn/modules/transformer.py) at batch sizes that fill the chip.
//
//   out[m,n] = act(sum_k a[m,k] * w[n,k] + bias[n])          (P_PLAIN16)
//   out[m,n] = act(rstd_m * acc - mean_m rstd_m colsum[n] + bias[n])   (P_FOLD16: a = raw residual rows, see gemm_16bit.hip)
//   (hi, lo)[m,n] = split(acc + bias[n] + res_hi[m,n] + res_lo[m,n]), row statistics per 64 columns   (P_SPLIT: out-proj / FFN2, in place)
//
// Why a second kernel next to the 128 x 128 persistent one (gemm_16bit.hip): that kernel's K loop needs 512 B of LDS-DMA per MFMA and
// tops out at ~0.3 of the MFMA peak; a 256 x 256 tile needs 256 B per MFMA, and the 8-phase schedule of the CDNA programming guide
// (section 5, "the 256^2 8-phase template") keeps the matrix pipe of every SIMD fed by two waves that alternate between an MFMA
// segment and a fragment-read + DMA-issue segment.  Measured on MI355X (profiles/r03/gemm_variant7_8phase_generic.log): the loop
// alone runs 1185 / 1294 TF at 4096^3 / 8192^3 against 753 / 868 TF for the 128 x 128 kernel -- but as a one-tile-per-workgroup
// kernel with the generic epilogue it LOSES at K = 768 (12 K-steps per tile: pipeline fill + a 128 KiB epilogue per tile with
// nothing to hide behind).  This kernel removes that per-tile cost:
//   * persistent: one 8-wave workgroup per CU walks an XCD-aware tile list; the LDS-DMA stream never stops at a tile seam (the last
//     two K-steps of a tile already stage K-steps 0 and 1 of the next tile: the seam is just a change of base pointers);
//   * v_mfma_f32_32x32x16 computes the TRANSPOSED product (weights as the A operand): a lane then owns ONE output row and four
//     consecutive columns per accumulator quad, so the epilogue packs 16-bit pairs without any cross-lane exchange and reads
//     per-row LayerNorm coefficients from two registers.  The instruction is bit-symmetric in its operands
//     (tools/mfma_swap_probe.hip, profiles/r03/mfma_swap_probe.log) and the k order is that of every other GEMM kernel here, so
//     results stay bit-identical to theirs (tests/test_gpu_round3.py);
//   * epilogue: accumulators -> 16-bit -> a wave-private 4 KiB LDS patch (conflict-free both ways: 16-byte XOR swizzle + a
//     half swap on odd row octets) -> 16-byte stores of whole 128-byte output lines.  The patch does not alias the ring; no
//     workgroup barrier is involved except in the LayerNorm fold, whose row statistics the four waves of a row group stage
//     for each other (two barriers per tile, see stage_cols / stage_rows).
//
// K loop (validated as `variant 7` of the generic kernel before it moved here).  Tile = 8 waves as 2 (rows) x 4 (columns), 128 x 64
// per wave = 4 x 2 MFMA tiles.  Two 64 KiB buffers (K-steps t, t+1), each four 16 KiB half-tiles: A rows 0-127 / 128-255, W rows
// 0-127 / 128-255; every wave moves two 1 KiB pieces of a half-tile.  Waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD
// one wave is in its 8-MFMA segment while the other reads fragments and issues DMA.  Per iteration (two K-steps), phase p =
// [reads + one half-tile of DMA] barrier [8 MFMAs] barrier:
//   p   reads (buffer)        LDS-DMA issued (2 per wave)       wait            MFMAs (quadrant of the wave's 128 x 64)
//   1   W q0, A q0   (0)      buffer 1, W rows   0-127 (t+1)                    Q00
//   2   A q1         (0)      buffer 1, W rows 128-255 (t+1)                    Q10
//   3   W q1         (0)      buffer 0, A rows   0-127 (t+2)                    Q11
//   4   --                    buffer 0, A rows 128-255 (t+2)    vmcnt(4)        Q01   -> buffer 1 (t+1) landed
//   5   W q0, A q0   (1)      buffer 0, W rows   0-127 (t+2)                    Q00
//   6   A q1         (1)      buffer 0, W rows 128-255 (t+2)                    Q10
//   7   W q1         (1)      buffer 1, A rows   0-127 (t+3)                    Q11
//   8   --                    buffer 1, A rows 128-255 (t+3)    vmcnt(4)        Q01   -> buffer 0 (t+2) landed
// (t+2, t+3 = K-steps 0, 1 of the NEXT tile in the last iteration of a tile.)  Write-after-read: A rows 0-127 are read by waves
// 0-3 only, A rows 128-255 by waves 4-7 one barrier later, W by everybody; each half is re-staged >= 2 barriers after the
// lgkmcnt(0) that retired its last read.  Read-after-write: the counted vmcnt sits before the first barrier of phases 4 / 8, the
// first read of that buffer two (waves 0-3) or three (waves 4-7) barriers later.  vmcnt also counts the epilogue's stores: they
// are older than the DMA a later vmcnt(4) leaves in flight, so a count can only wait for more than it needs, never for less.
#include "gemm16.h"

namespace bg {

constexpr int P256_BUF = 65536, P256_HALF = 16384, P256_RING = 2 * P256_BUF, P256_PATCH = 4096;

// A value the compiler must treat as unknown at this point: everything derived from it is (re)computed here instead of being kept
// in a register across the K loop -- the loop runs at the 256-VGPR limit, and a spilled value costs more than its register: hipcc
// follows every scratch reload with s_waitcnt vmcnt(0), which drains the LDS-DMA pipeline.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <bool F16, int MODE, bool STATS = false, int ACT = BG_ACT_NONE>   // ACT: the activation, a compile-time constant (plain / fold epilogues)
__global__ __launch_bounds__(512) void gemm16_p256_kernel(GemmArgs g) {
    constexpr bool FOLD = MODE == P_FOLD16, SPLIT = MODE == P_SPLIT;
    static_assert(MODE == P_PLAIN16 || MODE == P_FOLD16 || MODE == P_SPLIT, "16-bit output epilogues only");
    static_assert(SPLIT || !STATS, "row statistics are an output of the split-residual epilogue");
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[P256_RING + 8 * P256_PATCH];     // 160 KiB: one workgroup per CU

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int Mv = g.m_dev ? *g.m_dev : g.M;                      // rows present (compacted batch: device-side count)
    const int nt_n = g.N_pad >> 8;
    // (hybrid launches: only the row panels that fill complete rounds -- a 128 x 128 kernel runs the rest; the partition comes from
    //  the host or from the table the compaction kernel wrote, bg_common.h p256_rows)
    const int T_all = (g.hybrid ? ((g.rows256_dev ? *g.rows256_dev : g.rows256_host) >> 8) : ((Mv + 255) >> 8)) * nt_n;
    const int G = gridDim.x;
    // XCD-aware walk: workgroup b (on XCD b % 8) owns tiles first, first + G, ... of the row-major tile list, `first` being
    // consecutive for the workgroups of one XCD -- the ~32 tiles an XCD runs at a time cover 3-4 row panels x all column tiles, so
    // an A panel is fetched into that L2 once and the W tiles stay resident.
    int L = xcd_remap(blockIdx.x, G);
    if (L >= T_all) return;                                       // uniform per workgroup, before any barrier
    if constexpr (SPLIT) {
        // Two phase groups (long launches only: the launcher sets p256_stagger from 5 rounds of tiles on).  Workgroups that start
        // together stay together -- every CU in its K loop, then every CU in its 512 KiB of residual traffic.  The workgroups of
        // every second row panel (the grid is a multiple of the column tiles: the nt_n workgroups that share a panel's A rows stay in
        // one group, round after round) start `p256_stagger` x 1024 cycles late, about one K loop.  The measured effect is small
        // (+1-3 % at 6-8 rounds, a loss at 1-3: profiles/r04/gemm_split_bench_sweep.log): the epilogue is a per-wave latency chain
        // (three vmcnt(0) drains per tile), not a fabric burst -- which is what the experiment was for.
        if (g.p256_stagger > 0 && (((unsigned)L / (unsigned)nt_n) & 1u))
            for (int r = g.p256_stagger; r > 0; --r) __builtin_amdgcn_s_sleep(16);
    }
    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.a);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.w);
    const unsigned lda_b = (unsigned)g.lda * 2u, ldw_b = (unsigned)g.K * 2u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    const unsigned patch_lds = lds0 + (unsigned)(P256_RING + wave * P256_PATCH);
    unsigned char* patch = lds + P256_RING + wave * P256_PATCH;
    // the four patches of a wave group (waves wm * 4 .. + 3: the same 128 rows) as one 16 KiB region (LayerNorm-fold staging)
    const unsigned grp_lds = lds0 + (unsigned)(P256_RING + wm * 4 * P256_PATCH);
    const unsigned char* grp = lds + P256_RING + wm * 4 * P256_PATCH;
    const bool has_bias = g.bias != nullptr;
    // (a run-time test costs the plain / fold epilogues a v_max + v_cndmask per accumulator element; the split epilogue keeps one --
    //  always false there, p256_eligible -- because with the constant hipcc's register allocation moves and three dwords spill)
    const bool relu = SPLIT ? g.act == BG_ACT_RELU : ACT == BG_ACT_RELU;

    // ---- LDS-DMA: wave w moves pieces w and w + 8 (8 rows x 128 B each) of every 128-row half-tile ----
    // Source = wave-uniform base (SGPR pair: tile origin + k offset + half offset) + 32-bit per-lane byte offset; the 16-byte chunk
    // index is XOR-swizzled on the source side (the DMA writes lane-linear), the fragment reads apply the same involution.
    auto dma = [&](unsigned dst, const unsigned char* src, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };
    auto stage = [&](int buf, int h4, const unsigned char* src, const unsigned (&off)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            dma(lds0 + (unsigned)(buf * P256_BUF + h4 * P256_HALF + (wave + 8 * r) * 1024), src, off[r]);
    };
    unsigned hw[2], ha[2][2];                                     // per-lane source offsets: W rows (either half), A rows per half
    auto piece_row = [&](int ln, int r) { return (wave + 8 * r) * 8 + (ln >> 3); };            // row inside the half-tile
    auto piece_chunk = [&](int ln, int r) { return (unsigned)(((ln & 7) ^ ((piece_row(ln, r) >> 1) & 7)) * 16); };
    auto a_offsets = [&](int m0t) {                               // (re)computed at every tile seam from an opaque lane id
        const int ln = opaque(threadIdx.x & 63);
        const int last = Mv - 1 - m0t;                            // rows >= Mv are clamped (never stored)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int row = hh * 128 + piece_row(ln, r);
                row = row < last ? row : last;
                ha[hh][r] = (unsigned)row * lda_b + piece_chunk(ln, r);
            }
    };
    {
        const int ln = threadIdx.x & 63;
#pragma unroll
        for (int r = 0; r < 2; ++r) hw[r] = (unsigned)piece_row(ln, r) * ldw_b + piece_chunk(ln, r);
    }
    const unsigned w_half = 128u * ldw_b;                         // W rows 128-255: the same lane offsets on a shifted base

    // Epilogue operands wait in LDS while the K loop runs (no registers): one 1 KiB LDS-DMA piece each, issued at the END of the
    // previous tile's epilogue (here: in the prologue) and read into registers before the first pass of this tile's epilogue
    // overwrites the patches.  An L2 round trip per tile would otherwise sit in front of every epilogue, and a plain load there would
    // make hipcc wait for vmcnt(0), i.e. for the next tile's DMA as well.
    //   plain / split: the tile's 256 bias values in [0, 1K) of the wave's own patch.
    //   LayerNorm fold: the four waves of a group (same 128 rows) share one copy in the group's 16 KiB -- [0, 12K) the twelve
    //   (sum, sum of squares) partials of the 128 rows (1 KiB per 64-column part; wave wn brings parts 3 wn .. 3 wn + 2),
    //   [12K, 13K) bias (wave 0), [13K, 14K) column sums (wave 1).  What a sibling staged is visible after the K loop's barriers
    //   (each wave's counted vmcnt wait precedes them); two extra workgroup barriers per tile keep the region from being
    //   overwritten by a pass while a sibling still reads it, and by the next staging while a sibling's pass still uses it.
    //   (A 16-byte DMA element = the pairs of two rows: the launcher sends odd M / unaligned statistics to the 128 x 128 kernel.)
    auto stage_cols = [&](int n0c) {
        const unsigned voff = (unsigned)opaque(threadIdx.x & 63) * 16u;
        if constexpr (FOLD) {
            if (wn == 0) dma(grp_lds + 12288u, reinterpret_cast<const unsigned char*>(g.bias + n0c), voff);
            if (wn == 1) dma(grp_lds + 13312u, reinterpret_cast<const unsigned char*>(g.colsum + n0c), voff);
        } else {
            if (has_bias) dma(patch_lds, reinterpret_cast<const unsigned char*>(g.bias + n0c), voff);
        }
    };
    auto stage_rows = [&](int m0t) {
        if constexpr (FOLD) {
            const int ln = opaque(threadIdx.x & 63);
            int pair = ((m0t + wm * 128) >> 1) + ln;
            const int last = (Mv - 1) >> 1;                       // (rows past the end: any valid address, the result is never stored)
            pair = pair < last ? pair : last;
            const unsigned char* sb = reinterpret_cast<const unsigned char*>(g.stats_in);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int part = wn * 3 + r;
                dma(grp_lds + (unsigned)part * 1024u, sb + (size_t)part * (size_t)g.M * 8, (unsigned)pair * 16u);
            }
        }
    };

    // ---- fragment read addresses inside a buffer (A rows first, W rows at +32 KiB) ----
    unsigned a_rd, b_rd, xk[4];
    {
        const int ln = threadIdx.x & 63, l31 = ln & 31, hq = ln >> 5, sw = (l31 >> 1) & 7;
        a_rd = (unsigned)(wm * 128 + l31) * 128u;                 // + i * 4096
        b_rd = 32768u + (unsigned)(wn * 64 + l31) * 128u;         // + j * 4096
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xk[ks] = (unsigned)(((ks * 2 + hq) ^ sw) << 4);
    }
    f32x16 acc[4][2];
    V8 fa[2][2][4], fb[4];                                        // A sub-tiles q = 0, 1 (2 row tiles x 4 k-slices), one W sub-tile
    auto read_a = [&](const unsigned char* st, int qm) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[qm][i][ks] = *reinterpret_cast<const V8*>(st + a_rd + (2 * qm + i) * 4096 + xk[ks]);
    };
    auto read_b = [&](const unsigned char* st, int qn) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const V8*>(st + b_rd + qn * 4096 + xk[ks]);
    };
    auto quadrant = [&](int qm, int qn) {                         // 8 MFMAs: rows qm*64..+63 x columns qn*32..+31 of the wave's block
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)                           // transposed product: lane = output row, registers = columns
                acc[2 * qm + i][qn] = E::mfma(fb[ks], fa[qm][i][ks], acc[2 * qm + i][qn]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bar = [&]() {                                            // raw barrier; nothing -- at IR or machine level -- moves across it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };

    const int KT = g.K / G_BK;                                    // even, >= 2 (launcher)
    const unsigned char* b0 = lds;
    const unsigned char* b1 = lds + P256_BUF;

    int m0 = (L / nt_n) << 8, n0 = (L % nt_n) << 8;
    const unsigned char* a_cur = Ab + (size_t)m0 * lda_b;
    const unsigned char* w_cur = Wb + (size_t)n0 * ldw_b;
    a_offsets(m0);
    stage_cols(n0);
    stage_rows(m0);

    // ---- prologue: buffer 0 complete, the two A halves of buffer 1 in flight (as if issued in phases 7, 8) ----
    stage(0, 0, a_cur, ha[0]); stage(0, 1, a_cur, ha[1]); stage(0, 2, w_cur, hw); stage(0, 3, w_cur + w_half, hw);
    stage(1, 0, a_cur + 2 * G_BK, ha[0]); stage(1, 1, a_cur + 2 * G_BK, ha[1]);
    wait_vmcnt<4>();
    bar();
    if (wm == 1) bar();                                           // waves 4-7 run one barrier behind

    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // the next tile of this workgroup (its first two K-steps are staged by the last iteration below)
        const int Ln = L + G;
        const bool has_next = Ln < T_all;
        const int m0n = (Ln / nt_n) << 8, n0n = (Ln % nt_n) << 8;
        const unsigned char* a_nxt = Ab + (size_t)m0n * lda_b;
        const unsigned char* w_nxt = Wb + (size_t)n0n * ldw_b;

        // one iteration = K-steps t (buffer 0) and t + 1 (buffer 1).  w1: K-step t + 1 of W; (a2, w2): where K-step t + 2 comes
        // from, a3: K-step t + 3's A rows (`ha` holds the lane offsets of whichever tile a2 / a3 belong to); more = those exist.
        auto iteration = [&](const unsigned char* w1, bool more, const unsigned char* a2, const unsigned char* w2,
                             const unsigned char* a3) {
            // phase 1
            read_b(b0, 0); __builtin_amdgcn_sched_barrier(0); read_a(b0, 0);
            stage(1, 2, w1, hw);
            bar(); quadrant(0, 0); bar();
            // phase 2
            read_a(b0, 1);
            stage(1, 3, w1 + w_half, hw);
            bar(); quadrant(1, 0); bar();
            // phase 3
            read_b(b0, 1);
            if (more) stage(0, 0, a2, ha[0]);
            bar(); quadrant(1, 1); bar();
            // phase 4
            if (more) { stage(0, 1, a2, ha[1]); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
            bar(); quadrant(0, 1); bar();
            // phase 5
            read_b(b1, 0); __builtin_amdgcn_sched_barrier(0); read_a(b1, 0);
            if (more) stage(0, 2, w2, hw);
            bar(); quadrant(0, 0); bar();
            // phase 6
            read_a(b1, 1);
            if (more) stage(0, 3, w2 + w_half, hw);
            bar(); quadrant(1, 0); bar();
            // phase 7
            read_b(b1, 1);
            if (more) stage(1, 0, a3, ha[0]);
            bar(); quadrant(1, 1); bar();
            // phase 8
            if (more) { stage(1, 1, a3, ha[1]); wait_vmcnt<4>(); }
            bar(); quadrant(0, 1); bar();
        };
        for (int t = 0; t + 2 < KT; t += 2) {
            const unsigned kb = (unsigned)t * (2 * G_BK);         // byte offset of K-step t inside a row
            iteration(w_cur + kb + 2 * G_BK, true, a_cur + kb + 4 * G_BK, w_cur + kb + 4 * G_BK, a_cur + kb + 6 * G_BK);
        }
        if (has_next) a_offsets(m0n);                             // the last iteration stages A rows of the next tile only
        iteration(w_cur + (unsigned)(KT - 1) * (2 * G_BK), has_next, a_nxt, w_nxt, a_nxt + 2 * G_BK);

        // ---------------- epilogue ----------------
        if (wm == 0) bar();           // both wave groups run their epilogues together (one barrier apart they serialise: measured slower)
        {
            const int ln = opaque(threadIdx.x & 63), l31 = ln & 31, hq = ln >> 5;
            const int rbase = m0 + wm * 128, cbase = n0 + wn * 64;
            T* out = reinterpret_cast<T*>(g.out);
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            auto store16 = [&](int grow, int col, uint4 v) {
                if (grow < Mv) {
                    u32x4* dst = reinterpret_cast<u32x4*>(out + (size_t)grow * g.ldc + col);
                    const u32x4 vv = {v.x, v.y, v.z, v.w};
                    *dst = vv;
                }
            };
            // one accumulator quad (row l31 of row tile i, columns j*32 + 8 q + 4 hq .. +3) -> 4 x 16 bit
            auto quad = [&](int i, int j, int q, float4 b, float4 c, float2 cf) -> uint2 {
                const float b4[4] = {b.x, b.y, b.z, b.w}, c4[4] = {c.x, c.y, c.z, c.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = acc[i][j][4 * q + e];
                    v[e] = FOLD ? ln_fold_apply(x, cf.x, cf.y, c4[e], b4[e]) : x + b4[e];
                    if (relu) v[e] = fmaxf(v[e], 0.f);
                }
                union { V4 v; uint2 u; } pk;
                pk.v = E::pack4(v[0], v[1], v[2], v[3]);
                return pk.u;
            };
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (SPLIT) {
                // Split residual stream (out-proj / FFN2): x_new = acc + bias + (hi + lo) -> (hi, lo) planes + per-64-column row
                // statistics, in place.  The wave's 128 x 64 block goes through the 4 KiB patch in eight slabs of 16 rows x 64 fp32
                // columns (row tile t >> 1, rows 16 (t & 1) ..): the 32 lanes that own those rows write their eight quads; on the
                // way back a lane owns 8 consecutive columns of a row, so the residual arrives and hi / lo leave as 16-byte
                // accesses of whole 128-byte lines, and one 8-lane butterfly gives the row's (sum, sum of squares) -- the
                // arithmetic, its order and the statistics layout of the 128 x 128 kernel (gemm_16bit.hip), bit for bit.
                // The vector-memory counter is shared by loads and stores and the two classes retire out of order with respect to each
                // other (a counted wait that allowed younger STORES to stay in flight passed early under load: measured,
                // profiles/r03/gemm_p256_split_*.log).  Until round 5 the loads were therefore waited for with vmcnt(0) in three
                // batches (slabs 0-2, 3-5, 6-7); see the rotating set below for what replaced that.
                const int k8 = ln & 7, r8 = ln >> 3;
                float4 bias0 = zero4, bias1 = zero4;
                if (has_bias) {
                    bias0 = *reinterpret_cast<const float4*>(patch + (unsigned)(wn * 64 + k8 * 8) * 4u);
                    bias1 = *reinterpret_cast<const float4*>(patch + (unsigned)(wn * 64 + k8 * 8 + 4) * 4u);
                }
                const unsigned char* rh = reinterpret_cast<const unsigned char*>(g.res_hi);
                const unsigned char* rl = reinterpret_cast<const unsigned char*>(g.res_lo);
                const unsigned col_b = (unsigned)(cbase + k8 * 8) * 2u;
                // Round 5: the residual octets arrive through inline-asm loads into a ROTATING set of three slabs (48 registers, as
                // before), and the waits are counted in LOADS: slab t has landed when at most the loads issued after it are
                // outstanding (vmcnt(4 x slabs requested behind it)).  Loads retire in order among themselves; the stores of
                // earlier slabs share the counter, but a pending store can only make such a wait longer, never let it pass early
                // (round 3's wrong variant counted the stores as "may stay in flight").  The request for slab t + 3 goes out right
                // behind slab t's stores, so two slabs' loads travel while a third is finished and stored -- no vmcnt(0) drain inside
                // a tile's epilogue any more (there were three).  hipcc does not see these loads: nothing it inserts can drain them.
                u32x4 rbuf[3][2][2];                                  // [slab % 3][it][hi / lo]
                auto load16 = [&](u32x4& dst, const unsigned char* base, unsigned off) {
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(off), "s"(base) : "memory");
                };
                auto res_issue = [&](int t) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        int grow = rbase + t * 16 + it * 8 + r8;
                        grow = grow < Mv ? grow : Mv - 1;
                        const unsigned voff = (unsigned)grow * (unsigned)g.ld_res * 2u + col_b;
                        load16(rbuf[t % 3][it][0], rh, voff);
                        load16(rbuf[t % 3][it][1], rl, voff);
                    }
                };
                auto res_wait = [&](int t, auto younger_c) {          // slab t has landed; `younger` slabs were requested after it
                    constexpr int YOUNGER = decltype(younger_c)::value;
                    wait_vmcnt<4 * YOUNGER>();
                    asm volatile("" : "+v"(rbuf[t % 3][0][0]), "+v"(rbuf[t % 3][0][1]), "+v"(rbuf[t % 3][1][0]), "+v"(rbuf[t % 3][1][1]) :: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                };
                T* out_lo = reinterpret_cast<T*>(g.out_lo);
                // patch rows of 256 B; 16-byte chunk c XOR-swizzled by the row: ds_write_b128 and ds_read_b128 conflict-free
                auto slab = [&](int t) {
                    const int i = t >> 1, half = t & 1;
                    if ((l31 >> 4) == half) {
                        const int prow = l31 & 15;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int c16 = j * 8 + 2 * q + hq;
                                *reinterpret_cast<float4*>(patch + prow * 256 + ((c16 ^ prow) << 4)) =
                                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                            }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int prow = it * 8 + r8;
                        const float4 p0 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8) ^ prow) << 4));
                        const float4 p1 = *reinterpret_cast<const float4*>(patch + prow * 256 + (((2 * k8 + 1) ^ prow) << 4));
                        float v[8] = {p0.x + bias0.x, p0.y + bias0.y, p0.z + bias0.z, p0.w + bias0.w,
                                      p1.x + bias1.x, p1.y + bias1.y, p1.z + bias1.z, p1.w + bias1.w};
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        const u32x4 h4 = rbuf[t % 3][it][0], l4 = rbuf[t % 3][it][1];
                        float fh[4], fl[4];
                        unpack4_16<F16>(make_uint2(h4[0], h4[1]), fh);
                        unpack4_16<F16>(make_uint2(l4[0], l4[1]), fl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += fh[e] + fl[e];
                        unpack4_16<F16>(make_uint2(h4[2], h4[3]), fh);
                        unpack4_16<F16>(make_uint2(l4[2], l4[3]), fl);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[4 + e] += fh[e] + fl[e];
                        const int grow = rbase + t * 16 + prow;
                        const bool row_ok = grow < Mv;                // (in place: a clamped duplicate row must not be written)
                        if (STATS) {
                            const float s8 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                            const float q8 = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) +
                                             ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                            const float S = group8_sum(s8), Q = group8_sum(q8);
                            if (row_ok && k8 == 0)
                                reinterpret_cast<float2*>(g.stats_out)[(size_t)(cbase / 64) * g.M + grow] = make_float2(S, Q);
                        }
                        if (row_ok) {
                            const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
                            uint2 ha, la, hb, lb;
                            split4_16<F16>(va, ha, la);
                            split4_16<F16>(vb, hb, lb);
                            const size_t o = (size_t)grow * g.ldc + cbase + k8 * 8;
                            *reinterpret_cast<uint4*>(out + o) = make_uint4(ha.x, ha.y, hb.x, hb.y);
                            *reinterpret_cast<uint4*>(out_lo + o) = make_uint4(la.x, la.y, lb.x, lb.y);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                };
                using Y0 = std::integral_constant<int, 0>;
                using Y1 = std::integral_constant<int, 1>;
                using Y2 = std::integral_constant<int, 2>;
                res_issue(0); res_issue(1); res_issue(2);
                res_wait(0, Y2{}); slab(0); res_issue(3);
                res_wait(1, Y2{}); slab(1); res_issue(4);
                res_wait(2, Y2{}); slab(2); res_issue(5);
                res_wait(3, Y2{}); slab(3); res_issue(6);
                res_wait(4, Y2{}); slab(4); res_issue(7);
                res_wait(5, Y2{}); slab(5);
                res_wait(6, Y1{}); slab(6);
                res_wait(7, Y0{}); slab(7);
            } else {
                // a lane's 32 columns are cbase + j*32 + 8*q + 4*hq + e (q, e = 0..3), i.e. 8 float4 of bias (LayerNorm fold: and of
                // column sums, plus the (rstd, -mean rstd) pairs of its four rows) -- read before the first pass overwrites the patch
                float4 bz[2][4], cs[2][4];
                float2 cf[4];
                if constexpr (FOLD) {
                    // (rstd, -mean rstd) of the lane's four rows from the staged partials, in the association order of every other
                    // kernel (tree16 + ln_fold_coeffs); then the column vectors; then every wave of the workgroup must be done
                    // reading before any wave's first pass overwrites its patch
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float ps[16], pq[16];
#pragma unroll
                        for (int pp = 0; pp < 16; ++pp) {
                            const float2 v = pp < FOLD_PARTS ? reinterpret_cast<const float2*>(grp + pp * 1024)[i * 32 + l31] : make_float2(0.f, 0.f);
                            ps[pp] = v.x; pq[pp] = v.y;
                        }
                        cf[i] = ln_fold_coeffs(tree16(ps), tree16(pq), g.K, g.ln_eps);
                        // one row's 24 partials in registers at a time (no room for more): the pair is pinned here, or hipcc sinks the
                        // arithmetic to the pass that uses it and keeps the 24 inputs alive (spilled) until then
                        asm volatile("" : "+v"(cf[i].x), "+v"(cf[i].y) :: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned off = (unsigned)(wn * 64 + j * 32 + 8 * q + 4 * hq) * 4u;
                            bz[j][q] = *reinterpret_cast<const float4*>(grp + 12288 + off);
                            cs[j][q] = *reinterpret_cast<const float4*>(grp + 13312 + off);
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    bar();
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bz[j][q] = has_bias ? *reinterpret_cast<const float4*>(patch + (unsigned)(wn * 64 + j * 32 + 8 * q + 4 * hq) * 4u) : zero4;
                            cs[j][q] = zero4;
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i) cf[i] = make_float2(1.f, 0.f);
                }
                // patch rows of 128 B; 16-byte chunk (j*4 + q) XOR-swizzled by the row; rows 8-15 / 24-31 swap the two 8-byte
                // halves, so the 16 lanes of a ds_write_b64 service group hit 32 distinct banks
                const unsigned wr_base = (unsigned)l31 * 128u + (unsigned)((hq ^ ((l31 >> 3) & 1)) * 8);
                const unsigned rd_off = (unsigned)(ln >> 3) * 128u + (unsigned)(((ln & 7) ^ (ln >> 3)) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<uint2*>(patch + wr_base + (unsigned)(((j * 4 + q) ^ (l31 & 7)) * 16)) =
                                quad(i, j, q, bz[j][q], cs[j][q], cf[i]);
                    __builtin_amdgcn_wave_barrier();              // LDS executes a wave's accesses in order: no wait needed
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        uint4 v = *reinterpret_cast<const uint4*>(patch + it * 1024 + rd_off);
                        if (it & 1) v = make_uint4(v.z, v.w, v.x, v.y);
                        store16(rbase + i * 32 + it * 8 + (ln >> 3), cbase + (ln & 7) * 8, v);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (has_next) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the patch accesses above are complete
                if constexpr (FOLD) bar();                         // ... those of the sibling waves too: the staging below writes into their patches
                stage_cols(n0n);
                stage_rows(m0n);
            }
        }
        if (!has_next) break;
        if (wm == 1) bar();                                       // restore the one-barrier lag of waves 4-7
        L = Ln; m0 = m0n; n0 = n0n; a_cur = a_nxt; w_cur = w_nxt;
    }
}

bool p256_eligible(const GemmArgs& g) {
    const bool fold = g.stats_in != nullptr, split = g.out_lo != nullptr;
    if (!(g.N_pad % 256 == 0 && g.N == g.N_pad && g.K % (2 * G_BK) == 0 && g.K >= 2 * G_BK && g.ldc % 8 == 0 &&
          g.out_dtype != BG_F32 && g.add == nullptr && g.add2 == nullptr && g.row_map == nullptr && g.cv_C == 0 &&
          (size_t)255 * g.lda * 2 + 128 < 0xffffffffull && (size_t)255 * g.K * 2 + 128 < 0xffffffffull))
        return false;
    if (g.act != BG_ACT_NONE && (split || g.act != BG_ACT_RELU)) return false;     // activations compiled in: none / ReLU (split: none)
    if (split)          // split residual stream in place (out-proj / FFN2): residual planes required, no LayerNorm fold on top
        return !fold && g.res_hi != nullptr && g.res_lo != nullptr && g.ld_res % 8 == 0 && (size_t)g.M * g.ld_res * 2 < 0xffffffffull;
    return g.res_hi == nullptr && g.stats_out == nullptr &&
           (!fold || (g.K == FOLD_PARTS * G_BK && g.colsum != nullptr && g.bias != nullptr && (g.M & 1) == 0 && ((size_t)g.stats_in & 15) == 0));
}

template <bool F16>
int launch_p256(const GemmArgs& g, hipStream_t s) {
    // tiles this launch may own: host-side row count -> the hybrid rule; device-side row count -> every tile of the bound (the rule
    // is evaluated on the device, surplus workgroups exit at once)
    const int tiles = ((g.hybrid && g.m_dev == nullptr ? p256_rows(g.M, g.N_pad / 256, g.out_lo != nullptr, g.hybrid == 2) : g.M + 255) / 256) * (g.N_pad / 256);
    if (tiles == 0) return 0;
    int grid = tiles < 256 ? tiles : 256;
    if (g.out_lo) {                                               // split-residual launches: whole row panels per round (phase groups)
        const int nt_n = g.N_pad / 256;
        grid = grid / nt_n * nt_n;
    }
    const bool relu = g.act == BG_ACT_RELU;                       // (p256_eligible: act is none or ReLU; split launches: none)
    if (g.out_lo && g.stats_out) hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_SPLIT, true>), dim3(grid), dim3(512), 0, s, g);
    else if (g.out_lo) hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_SPLIT, false>), dim3(grid), dim3(512), 0, s, g);
    else if (g.stats_in && relu) hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_FOLD16, false, BG_ACT_RELU>), dim3(grid), dim3(512), 0, s, g);
    else if (g.stats_in) hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_FOLD16, false, BG_ACT_NONE>), dim3(grid), dim3(512), 0, s, g);
    else if (relu) hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_PLAIN16, false, BG_ACT_RELU>), dim3(grid), dim3(512), 0, s, g);
    else hipLaunchKernelGGL((gemm16_p256_kernel<F16, P_PLAIN16, false, BG_ACT_NONE>), dim3(grid), dim3(512), 0, s, g);
    return launch_status("gemm16_p256");
}
template int launch_p256<false>(const GemmArgs&, hipStream_t);
template int launch_p256<true>(const GemmArgs&, hipStream_t);

}  // namespace bg
