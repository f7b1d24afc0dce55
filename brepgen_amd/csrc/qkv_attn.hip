// QKV projection (LayerNorm folded) + multi-head self-attention of one encoder layer in ONE launch, for batches of short,
// equally long sequences, with or without a key-padding mask (the face LDM's SurfPosNet: 30 / 60 tokens per sample; the dense
// execution of SurfZNet; network.py:1076-1078 ->
// torch/nn/modules/transformer.py + MultiheadAttention):
//
//   q|k|v[m, :] = T(rstd_m * (x_m . W'^T) - mean_m rstd_m colsum + b')        (the P_FOLD16 epilogue of gemm_p256.hip)
//   out[b, :, head] = softmax(q k^T) v  per sample b and head                  (attn16_kernel, attn.hip; 1/8 folded into W'_q)
//
// As two launches the 256 x 256 GEMM writes q|k|v (4.6 KB per token) and the attention kernel reads it back: 283 MB per layer at
// the face LDM's 30 720 tokens, all of it inside the GEMM's store burst (every CU reaches its epilogue at the same time) and the
// attention kernel's read -- neither overlaps MFMA work.  Here q|k|v of one (sample group, head) never leave the CU: the launch
// takes what the GEMM alone took (profiles/r04/qkv_attn_check_*.log: 131 vs 127 + 37 us at 512 x 60 tokens).
//
// Tile = 256 token slots x 192 columns (q, k, v of ONE head, 64 each) x K-steps of 64; the 256 slots are 256 / S samples of S = 64
// (or 32) slots each, a sample's N <= S tokens followed by copies of its last token (never stored, masked as keys) -- so the rows
// of a tile are whole samples and a wave's 32 queries belong to one of them.  8 waves as 4 (rows) x 2 (columns), 64 x 96 per wave
// = 2 x 3 MFMA tiles, transposed product (weights as the A operand: a lane owns one token); one persistent workgroup per CU walks
// an XCD-aware list of (sample group, head) tiles.  LDS (160 KiB): buffer 0 | buffer 1 (one K-step each: A rows 0-127 | A rows
// 128-255 | W rows q, k, v = 56 KiB) | 24 KiB statistics partials | ... | bias + column sums (2 KiB at the end).
//
// K loop: the phase discipline of gemm_p256.hip on this geometry -- waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one
// wave is in an 8-MFMA segment while the other reads fragments and issues LDS-DMA.  Six phases per iteration (two K-steps), phase =
// [reads + DMA] barrier [8 MFMAs: both row tiles x 4 k-slices of one column tile] barrier:
//   p   reads (buffer)            LDS-DMA issued per wave              wait         MFMAs
//   1   W j0, A          (0)      buffer 1, W (t+1): 3 pieces                       column tile 0
//   2   W j1, W j2       (0)                                                        column tile 1
//   3   --                        buffer 0, A (t+2): 4 pieces          vmcnt(4)     column tile 2     -> buffer 1 (t+1) landed
//   4   W j0, A          (1)      buffer 0, W (t+2): 3 pieces                       column tile 0
//   5   W j1, W j2       (1)                                                        column tile 1
//   6   --                        buffer 1, A (t+3): 4 pieces          vmcnt(4)     column tile 2     -> buffer 0 (t+2) landed
// Write-after-read: a region last read in phase R (by any wave, the late group included) is re-staged from phase R + 2 on -- A is
// last read in phases 1 / 4, W in 2 / 5.  Read-after-write: the counted wait sits before the first barrier of phases 3 / 6, the
// first read of that buffer one phase later (gemm_p256.hip: the same argument).  The read-free phases 3 / 6 of two late iterations
// also carry the LayerNorm-fold coefficients of the lane's two tokens (sums of the twelve staged partials, then rstd and
// -mean rstd: two light steps each, behind the partner wave's MFMA segment).  In the last iteration of a tile (t+2) is K-step 0 of
// the NEXT tile and nothing is staged into buffer 1: the seam costs no prologue (measured: 2.5-3.4 k of 36 k cycles before).
//
// Epilogue (all eight waves together): fold + bias + 16-bit rounding exactly as P_FOLD16, then q, k and v go to LDS row-major (one
// 128-byte row per token slot, 16-byte chunks XOR-swizzled by the row: eight-byte stores straight from the accumulator quads) --
// buffer 1 and the place of the statistics, consumed by then; buffer 0 already holds the next tile's first K-step -- one barrier;
// every wave runs the attention of 32 queries of one sample -- attn16_kernel's arithmetic (the 0 / -inf key bias only where a
// sub-tile holds dead keys, no rescaling of the empty accumulator: neither changes a bit), the V^T fragments of P V read with
// ds_read_b64_tr_b16 (round 5; until then v was stored transposed with 64 two-byte LDS stores per wave) -- and stores 32 x 64
// outputs, 16 bytes per lane after one v_permlane32_swap per register pair; one barrier.
// Per tile (s_memtime, profiles/r04/qkv_attn_stamps_*.log): K loop 21 k cycles for 18.4 k cycles of MFMA issue per SIMD, fold +
// images 3.5-4.8 k, attention 4.4-7 k (VALU-issue-bound: two waves per SIMD), barriers 1.5 + 2.6 k -- round 4's stamps; round 5's
// epilogue (row-major v, transpose reads, 16-byte stores) took 4.5 % off the launch (profiles/r05/qkv_attn_v_row_major_*.log).
// Results are bit-identical to gemm (P_FOLD16) + attention (tests/test_gpu_round4.py).
//
// Ragged batches (PAIR; SurfZNet's variable-length execution): the rows arrive SLOT-PACKED (compact.hip: compact_rows_paired) --
// every 64-row slot holds one or two whole samples and clones of its first row, *m_dev rows exist, slot_desc gives (n_a, n_b) per
// slot.  Tile rows are then consecutive global rows, and a wave walks the keys of each of its (at most two) samples from that
// sample's own first key (K and V rows at any offset inside the slot: both images are row-major, rows clamped to the slot's 64);
// a lane of the other sample sees -inf scores, which makes its online-softmax update an exact no-op -- so every valid token ends
// with the bits attn16_kernel produces on the dense packing.  The K loop is the same.
#include "gemm16.h"
#include <math.h>

namespace bg {

constexpr int QA_BUF = 57344, QA_AHALF = 16384, QA_WOFF = 32768, QA_RING = 2 * QA_BUF;      // 2 x 56 KiB
constexpr int QA_LDS = 163840;
constexpr int QA_AUX = QA_RING, QA_BIAS = QA_LDS - QA_AUX - 2048, QA_CSUM = QA_BIAS + 1024; // aux: [0, 24K) statistics ... bias, column sums at the end
// epilogue images: buffer 1 and the statistics' place (consumed inside the K loop) -- buffer 0 already holds K-step 0 of the next tile
constexpr int QA_QIMG = QA_BUF, QA_KIMG = QA_QIMG + 32768, QA_VIMG = QA_KIMG + 32768;

constexpr int QA_MB = QA_AUX + QA_BIAS - 1024;             // 256 floats: additive key bias (0 / -inf) of the tile's slots (key-padding mask given)
static_assert(QA_VIMG + 32768 <= QA_MB && QA_VIMG + 32768 <= QA_AUX + QA_BIAS, "the V image ends below the key bias / bias / column sums");

struct QkvAttnArgs {
    const void* a;            // [M, 768] raw 16-bit residual rows (hi plane)
    const void* w;            // [2304, 768] = T(gamma * W_qkv), q rows pre-scaled by 1/8
    const float* bias;        // [2304] b + W beta
    const float* colsum;      // [2304]
    const float* stats_in;    // [12][M][2] (sum, sum of squares) per 64-column part
    void* out;                // [M, 768] attention output (16-bit)
    void* dbg;                // optional [M, 2304]: the q|k|v a two-launch run would have written (tests)
    const uint8_t* key_pad;   // optional [B, N]: 1 = padded key (dense execution of a masked net); queries are computed at every position
    int B, N, M;
    float ln_eps;
    // slot-packed ragged batch (compact.hip: compact_rows_paired): rows = 64-row slots of one or two whole samples, *m_dev of them
    // exist (a multiple of 64), slot_desc[2 k] / [2 k + 1] = the lengths of slot k's samples; B is then the slot bound, N = 64
    const int* m_dev = nullptr;
    const int* slot_desc = nullptr;
    int pin = 0;              // 1: XCD-pinned tile walk (see the kernel)
};

typedef __attribute__((ext_vector_type(4))) short qa_v4s;

__device__ __forceinline__ int qa_opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <bool F16, int S, bool MASK, bool DBG = false, bool PAIR = false>
__global__ __launch_bounds__(512) void qkv_attn_kernel(QkvAttnArgs g) {
    using E = Elem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    constexpr int SPT = 256 / S;              // samples per tile
    constexpr int WPS = S / 32;               // waves (query blocks) per sample
    constexpr int LD = BG_D_MODEL;            // 768
    __shared__ __attribute__((aligned(16))) unsigned char lds[QA_LDS];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (wm, wn): rows 64 wm .., columns 96 wn ..; waves w and w + 4 share a SIMD: one of them gets wn = 0 (q + half of k), the other
    // wn = 1 (half of k + v)
    const int wm = wave >> 1, wn = (wave ^ (wave >> 2)) & 1;
    const int late = wave >> 2;               // waves 4-7 (rows 128-255) run one barrier behind
    static_assert(!PAIR || (S == 64 && !MASK), "slot-packed batches: 64-row slots, dead keys come from the slot descriptors");
    const int Mv = PAIR ? *g.m_dev : 0;                           // slot-packed batch: rows present (device-side count, a multiple of 64)
    const int n_groups = PAIR ? (Mv + 255) >> 8 : (g.B + SPT - 1) / SPT;
    const int T_all = n_groups * BG_N_HEAD;
    const int G = gridDim.x;
    int L = xcd_remap(blockIdx.x, G);
    // XCD-pinned walk (large launches): XCD x = blockIdx.x & 7 keeps ONE half of the heads for the whole launch (x & 1) and a quarter
    // of the sample groups (x >> 1), so its L2 holds 6 heads' weights (1.8 MB) while the A panels stream through -- with the plain
    // walk every XCD runs all 12 heads, the 3.5 MB of W' plus the round's A panels overflow the 4 MiB L2 and W' is re-fetched from
    // the fabric in every round of tiles (PMC: 181 -> 139 MB of reads per launch at 512 x 60, launch -3...-9 %, step -1...-2 %:
    // profiles/r05/qkv_attn_tile_walk_*.log).  L then counts the XCD's own tiles: tile j -> group 4 (j / 6) + (x >> 1), head
    // 6 (x & 1) + j % 6, j advancing by the XCD's workgroup count.
    const bool pin = g.pin != 0;
    const int pin_q = (blockIdx.x & 7) >> 1, pin_h0 = (blockIdx.x & 1) * 6, pin_step = G >> 3;
    auto tile_of = [&](int l, int& tg, int& th) -> bool {
        if (pin) { tg = 4 * (l / 6) + pin_q; th = pin_h0 + l % 6; return tg < n_groups; }
        tg = l / BG_N_HEAD; th = l % BG_N_HEAD;
        return l < T_all;
    };
    if (pin) L = blockIdx.x >> 3;
    {
        int tg, th;
        if (!tile_of(L, tg, th)) return;
    }

    const unsigned char* Ab = reinterpret_cast<const unsigned char*>(g.a);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(g.w);
    constexpr unsigned lda_b = LD * 2u, ldw_b = LD * 2u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    const unsigned aux_lds = lds0 + QA_AUX;
    const unsigned char* aux = lds + QA_AUX;

    auto dma = [&](unsigned dst, const unsigned char* src, unsigned voff) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
    };

    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    auto lds_done_bar = [&]() {               // every LDS access of every wave issued so far is complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
    };

    // ---- LDS-DMA pieces (1 KiB = 8 rows x 128 B): wave w moves pieces w, w + 8 of each A half and piece w of each of W's q, k, v ----
    unsigned ha[2][2], hw;
    auto piece_row = [&](int ln, int r) { return (wave + 8 * r) * 8 + (ln >> 3); };
    auto piece_chunk = [&](int ln, int r) { return (unsigned)(((ln & 7) ^ ((piece_row(ln, r) >> 1) & 7)) * 16); };
    auto slot_row = [&](int grp, int tile_row) {                  // global token of a tile row (slots past a sample's end: its last token)
        if (PAIR) {                                               // slot-packed batch: the tile's rows are consecutive global rows
            const int row = grp * 256 + tile_row;
            return row < Mv ? row : Mv - 1;
        }
        int smp = grp * SPT + tile_row / S;
        smp = smp < g.B ? smp : g.B - 1;
        int tok = tile_row % S;
        tok = tok < g.N ? tok : g.N - 1;
        return smp * g.N + tok;
    };
    auto a_offsets = [&](int grp) {
        const int ln = qa_opaque(threadIdx.x & 63);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                ha[hh][r] = (unsigned)slot_row(grp, hh * 128 + piece_row(ln, r)) * lda_b + piece_chunk(ln, r);
    };
    {
        const int ln = threadIdx.x & 63;
        hw = (unsigned)piece_row(ln, 0) * ldw_b + piece_chunk(ln, 0);
    }
    auto stage_a = [&](int buf, const unsigned char* src) {      // src = A + byte offset of the K-step
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                dma(lds0 + (unsigned)(buf * QA_BUF + hh * QA_AHALF + (wave + 8 * r) * 1024), src, ha[hh][r]);
    };
    auto stage_w = [&](int buf, const unsigned char* src) {      // src = W + head rows + byte offset of the K-step
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
            dma(lds0 + (unsigned)(buf * QA_BUF + QA_WOFF + (wave + 8 * c3) * 1024), src + (size_t)c3 * LD * ldw_b, hw);
    };
    // epilogue operands: statistics partials of the tile's 256 token slots ([row half][part][128 x (sum, sumsq)], 3 pieces per wave:
    // staged at the top of a tile, consumed inside its K loop), bias and column sums of the head's 192 columns (waves 0 / 1)
    auto stage_stats = [&](int grp) {
        const int ln = qa_opaque(threadIdx.x & 63);
        const int gh = wave >> 2;
        int row = slot_row(grp, gh * 128 + 2 * ln);
        row = (row & 1) ? row - 1 : row;                          // (N even: a slot pair never straddles samples; past the end: the last pair)
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(g.stats_in);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int part = (wave & 3) * 3 + r;
            dma(aux_lds + (unsigned)(gh * 12288 + part * 1024), sb + (size_t)part * (size_t)g.M * 8, (unsigned)(row >> 1) * 16u);
        }
    };
    auto stage_cols = [&](int head) {
        const int ln = qa_opaque(threadIdx.x & 63);
        int c3 = ln >> 4;
        c3 = c3 < 3 ? c3 : 2;
        const unsigned voff = (unsigned)(c3 * LD + (ln & 15) * 4) * 4u;
        if (wave == 0) dma(aux_lds + QA_BIAS, reinterpret_cast<const unsigned char*>(g.bias + head * 64), voff);
        if (wave == 1) dma(aux_lds + QA_CSUM, reinterpret_cast<const unsigned char*>(g.colsum + head * 64), voff);
    };

    // ---- fragment reads ----
    unsigned a_rd, b_rd, xk[4];
    {
        const int ln = threadIdx.x & 63, l31 = ln & 31, hq = ln >> 5, sw = (l31 >> 1) & 7;
        a_rd = (unsigned)(wm * 64 + l31) * 128u;                  // + i * 4096
        b_rd = (unsigned)QA_WOFF + (unsigned)(wn * 96 + l31) * 128u;   // + j * 4096
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xk[ks] = (unsigned)(((ks * 2 + hq) ^ sw) << 4);
    }
    f32x16 acc[2][3];
    V8 fa[2][4], fb[3][4];
    auto read_a = [&](const unsigned char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[i][ks] = *reinterpret_cast<const V8*>(st + a_rd + i * 4096 + xk[ks]);
    };
    auto read_b = [&](const unsigned char* st, int j) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb[j][ks] = *reinterpret_cast<const V8*>(st + b_rd + j * 4096 + xk[ks]);
    };
    auto segment = [&](int j) {                                  // 8 MFMAs: both row tiles x column tile j
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = E::mfma(fb[j][ks], fa[i][ks], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    constexpr int KT = LD / G_BK;                                 // 12 K-steps
    const unsigned char* b0 = lds;
    const unsigned char* b1 = lds + QA_BUF;

    int grp, head;
    tile_of(L, grp, head);
    // LayerNorm-fold coefficients (rstd, -mean rstd) of the lane's token of row tile i, from the staged partials -- in two light steps
    // (sums of the twelve partials; the coefficients) that fit behind an 8-MFMA segment of the partner wave
    float2 cf[2];
    auto fold_sums = [&](int i) {
        const int l31 = qa_opaque(threadIdx.x & 31);
        const unsigned char* sg = aux + (wm >> 1) * 12288;
        const int rg = (wm & 1) * 64 + i * 32 + l31;
        float ps[16], pq[16];
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            const float2 v = pp < FOLD_PARTS ? reinterpret_cast<const float2*>(sg + pp * 1024)[rg] : make_float2(0.f, 0.f);
            ps[pp] = v.x; pq[pp] = v.y;
        }
        cf[i] = make_float2(tree16(ps), tree16(pq));
        asm volatile("" : "+v"(cf[i].x), "+v"(cf[i].y) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fold_coeffs = [&](int i) {
        cf[i] = ln_fold_coeffs(cf[i].x, cf[i].y, LD, g.ln_eps);
        asm volatile("" : "+v"(cf[i].x), "+v"(cf[i].y) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue of the first tile: buffer 0 complete, A of buffer 1 in flight (as if issued in phase 6) ----
    a_offsets(grp);
    stage_stats(grp); stage_cols(head);
    stage_a(0, Ab); stage_w(0, Wb + (size_t)head * 64 * ldw_b); stage_a(1, Ab + 2 * G_BK);
    wait_vmcnt<4>();
    bar();
    for (;;) {
        // here: buffer 0 = K-step 0 of this tile, landed and visible; A rows of K-step 1 in flight; every wave at the same barrier
        if (late) bar();
        const unsigned char* w_cur = Wb + (size_t)head * 64 * ldw_b;
        const int Ln = L + (pin ? pin_step : G);
        int grp_n, head_n;
        const bool has_next = tile_of(Ln, grp_n, head_n);
        const unsigned char* w_nxt = Wb + (size_t)head_n * 64 * ldw_b;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // one iteration = K-steps t (buffer 0), t + 1 (buffer 1).  w1: K-step t + 1 of W; more: (a2, w2) = K-step t + 2 exists
        // (the last iteration: K-step 0 of the NEXT tile, `ha` then holds that tile's lane offsets); more3: a3 = K-step t + 3 of A
        // (never across the tile seam: buffer 1 becomes the epilogue's); fold: this iteration computes the LayerNorm-fold
        // coefficients of row tile `fold` in its two read-free phases (the statistics landed before the first counted wait of the tile)
        auto iteration = [&](const unsigned char* w1, bool more, const unsigned char* a2, const unsigned char* w2, bool more3,
                             const unsigned char* a3, int fold) {
            // phase 1
            read_b(b0, 0); __builtin_amdgcn_sched_barrier(0); read_a(b0);
            stage_w(1, w1);
            bar(); segment(0); bar();
            // phase 2
            read_b(b0, 1); read_b(b0, 2);
            bar(); segment(1); bar();
            // phase 3
            if (more) { stage_a(0, a2); wait_vmcnt<4>(); } else { wait_vmcnt<0>(); }
            if (fold >= 0) fold_sums(fold);
            bar(); segment(2); bar();
            // phase 4
            read_b(b1, 0); __builtin_amdgcn_sched_barrier(0); read_a(b1);
            if (more) stage_w(0, w2);
            bar(); segment(0); bar();
            // phase 5
            read_b(b1, 1); read_b(b1, 2);
            bar(); segment(1); bar();
            // phase 6
            if (more3) { stage_a(1, a3); wait_vmcnt<4>(); } else if (more) { wait_vmcnt<0>(); }
            if (fold >= 0) fold_coeffs(fold);
            bar(); segment(2); bar();
        };
        for (int t = 0; t + 2 < KT; t += 2) {
            const unsigned kb = (unsigned)t * (2 * G_BK);
            iteration(w_cur + kb + 2 * G_BK, true, Ab + kb + 4 * G_BK, w_cur + kb + 4 * G_BK, true, Ab + kb + 6 * G_BK,
                      t == KT - 6 ? 0 : (t == KT - 4 ? 1 : -1));
        }
        if (has_next) a_offsets(grp_n);                           // the last iteration stages A rows of the next tile only
        iteration(w_cur + (unsigned)(KT - 1) * (2 * G_BK), has_next, Ab, w_nxt, false, Ab, -1);
        if (!late) bar();                                         // both wave groups enter the epilogue together

        // ---------------- epilogue ----------------
        const int ln = qa_opaque(threadIdx.x & 63), l31 = ln & 31, hq = ln >> 5;
        T* dbg = reinterpret_cast<T*>(g.dbg);
        // slot-packed batch: the lengths of the two samples of this wave's slot (rows 64 wm ..: the wave's GEMM rows AND its queries)
        int pair_na = 64, pair_nb = 0;
        if (PAIR) {
            const int gslot = grp * 4 + wm;
            if (gslot * 64 < Mv) { pair_na = g.slot_desc[2 * gslot]; pair_nb = g.slot_desc[2 * gslot + 1]; }
            pair_na = __builtin_amdgcn_readfirstlane(pair_na);
            pair_nb = __builtin_amdgcn_readfirstlane(pair_nb);
        }
        float key_bias = 0.f;                                    // (MASK) slot threadIdx.x of the tile: requested now, written to LDS after the images
        if (MASK && threadIdx.x < 256) {
            const int smp = grp * SPT + (int)threadIdx.x / S, key = (int)threadIdx.x % S;
            const bool inside = smp < g.B && key < g.N;
            const uint8_t pad = inside ? g.key_pad[(size_t)smp * g.N + key] : (uint8_t)1;
            key_bias = pad ? -INFINITY : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tc0 = wn * 96 + j * 32;                     // wave-uniform: first tile column of this column tile
            const int c3 = tc0 >> 6, d0 = tc0 & 63;               // q / k / v, first head dimension
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = d0 + 8 * q + 4 * hq;
                const float4 bz = *reinterpret_cast<const float4*>(aux + QA_BIAS + (unsigned)(c3 * 64 + d) * 4u);
                const float4 cs = *reinterpret_cast<const float4*>(aux + QA_CSUM + (unsigned)(c3 * 64 + d) * 4u);
                const float b4[4] = {bz.x, bz.y, bz.z, bz.w}, c4[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ln_fold_apply(acc[i][j][4 * q + e], cf[i].x, cf[i].y, c4[e], b4[e]);
                    // (fp16: hipcc would fuse the outer fma with the conversion -- v_fma_mixlo_f16 rounds ONCE, the other kernels
                    //  round to fp32 and then to fp16; the fp32 value is pinned so that the two launches and this one agree bit for bit)
                    if (F16) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                    union { V4 v; uint2 u; T t[4]; } pk;
                    pk.v = E::pack4(v[0], v[1], v[2], v[3]);
                    const int R = wm * 64 + i * 32 + l31;         // token slot of the tile
                    // q, k and v all row-major, one 128-byte row per token slot, 16-byte chunks XOR-swizzled by the row (v was a
                    // transposed image until round 5: 64 ds_write_b16 per wave against these 16 ds_write_b64; the attention now
                    // reads it with ds_read_b64_tr_b16, as attn16_long_kernel does)
                    *reinterpret_cast<uint2*>(lds + QA_QIMG + c3 * 32768 + R * 128 + ((((d >> 3) ^ ((R >> 1) & 7))) << 4) + (d & 7) * 2) = pk.u;
                    if (DBG) {
                        const int smp = grp * SPT + R / S, tok = R % S;
                        if (PAIR ? grp * 256 + R < Mv : (smp < g.B && tok < g.N))
                            *reinterpret_cast<uint2*>(dbg + (size_t)(PAIR ? grp * 256 + R : smp * g.N + tok) * (3 * LD) + c3 * LD + head * 64 + d) = pk.u;
                    }
                }
            }
        }
        if (MASK && threadIdx.x < 256) reinterpret_cast<float*>(lds + QA_MB)[threadIdx.x] = key_bias;
        lds_done_bar();                                           // images complete; nobody reads the aux region any more

        if (has_next) stage_cols(head_n);                        // bias / column sums of the next tile travel while the attention runs

        // ---- attention: 32 queries of one sample per wave (attn16_kernel, attn.hip) ----
        // Slot-packed batch: the slot holds two samples; each is walked in sub-tiles of 32 keys FROM ITS OWN first key, exactly as
        // attn16_kernel walks a compacted sample.  A lane that does not belong to the block's sample sees -inf scores: its running
        // maximum stays, alpha = exp(0) = 1, p = 0 -- the update is an exact no-op, so every lane ends with attn16_kernel's bits.
        {
            const int wq = wm * 2 + wn;                            // (a bijection of the waves; wm = the sample's row block)
            const int slot = wq / WPS, qb = wq % WPS;
            const int smp = grp * SPT + slot;
            if (PAIR ? (grp * 256 + slot * 64 < Mv) : (smp < g.B)) {
                const int h = hq;
                const unsigned char* qimg = lds + QA_QIMG + (slot * S + qb * 32) * 128;
                const unsigned char* ktile = lds + QA_KIMG + slot * S * 128;
                // V^T fragments straight from the row-major image by transpose reads: inside a 16-lane group, lane i supplies the address
                // of key (i >> 2), head dimensions 4 (i & 3) .. + 3 and receives 4 consecutive keys of dimension i; group g covers
                // dimensions 16 (g & 1) .. + 15 of the 32-wide tile and the keys of k-chunk g >> 1 = hq (attn.hip: attn16_long_kernel)
                const unsigned char* vimg = lds + QA_VIMG + slot * S * 128;
                const int gi = ln & 15, gg = ln >> 4;
                const int v_key0 = 4 * (gg >> 1) + (gi >> 2);     // key inside a 16-key slice (second read: + 8)
                int v_c16[2], v_b8;                               // 16-byte chunk of the lane's dimensions per 32-wide tile, byte inside it
                v_b8 = ((gi & 3) * 4 & 7) * 2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) v_c16[dt] = (dt * 32 + (gg & 1) * 16 + (gi & 3) * 4) >> 3;
                const int sw = (l31 >> 1) & 7;
                V8 qf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qimg + l31 * 128 + (((ks * 2 + h) ^ sw) << 4));
                f32x16 o[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
                float m_run = -INFINITY, l_run = 0.f;
                const int tq = qb * 32 + l31;                     // the lane's query row inside its slot
                const bool in_b = PAIR && tq >= pair_na && tq < pair_na + pair_nb;   // (the clones behind both samples belong to a)
                bool first = true;                                // uniform: no score block has been executed yet
#pragma unroll
                for (int si = 0; si < (PAIR ? 2 : 1); ++si) {
                    const int cnt = PAIR ? (si == 0 ? pair_na : pair_nb) : g.N;
                    const int row0 = (PAIR && si == 1) ? pair_na : 0;        // the sample's first key row inside the slot ...
                    const bool member = !PAIR || (in_b == (si == 1));
                    if (PAIR && __ballot(member) == 0ull) continue;          // uniform: none of the wave's queries is in this sample
#pragma unroll
                    for (int sub = 0; sub < WPS; ++sub) {
                        if (sub * 32 >= cnt) break;               // uniform: sub-tile entirely past the last key
                        f32x16 s;
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] = 0.f;
                        int kr = row0 + sub * 32 + l31;           // key row inside the slot (slot-packed: never past the slot)
                        if (PAIR) kr = kr < 63 ? kr : 63;
                        const int ksw = PAIR ? (kr >> 1) & 7 : sw;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const V8 kf = *reinterpret_cast<const V8*>(ktile + kr * 128 + (((ks * 2 + h) ^ ksw) << 4));
                            s = E::mfma(kf, qf[ks], s);
                        }
                        // register r <-> key sub*32 + (r&3) + 8*(r>>2) + 4*h of query l31
                        // (attn16_kernel adds a 0 / -inf bias to every score; adding 0 changes nothing the softmax can see, so only a
                        //  sub-tile that holds keys past the sample's end pays for it)
                        if (MASK) {                               // the sample's padded keys (and the slots past its end): attn16_kernel's mb
                            const float* mb = reinterpret_cast<const float*>(lds + QA_MB) + slot * S + sub * 32 + 4 * h;
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const float4 mbv = *reinterpret_cast<const float4*>(mb + 8 * g4);
                                s[4 * g4 + 0] += mbv.x; s[4 * g4 + 1] += mbv.y; s[4 * g4 + 2] += mbv.z; s[4 * g4 + 3] += mbv.w;
                            }
                        } else if (PAIR) {
                            const int thr = member ? cnt - sub * 32 - 4 * h : -64;   // dead: keys past the sample's end; every key for a lane of the other sample
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[r] += ((r & 3) + 8 * (r >> 2)) >= thr ? -INFINITY : 0.f;
                        } else if (sub * 32 + 32 > g.N) {
                            const int thr = g.N - sub * 32 - 4 * h;   // register r is dead iff (r&3) + 8 (r>>2) >= thr
#pragma unroll
                            for (int r = 0; r < 16; ++r) s[r] += ((r & 3) + 8 * (r >> 2)) >= thr ? -INFINITY : 0.f;
                        }
                        float mloc = -INFINITY;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
                            mloc = fmaxf(mloc, fmaxf(fmaxf(s[4 * g4 + 0], s[4 * g4 + 1]), fmaxf(s[4 * g4 + 2], s[4 * g4 + 3])));
                        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                        const float m_new = fmaxf(m_run, mloc);
                        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                        float psum = 0.f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            s[r] = __expf(s[r] - m_use);
                            psum += s[r];
                        }
                        if (first) {                              // (running sum and O are 0: their rescaling by alpha = 0 leaves 0)
                            l_run = psum;
                        } else {
                            const float alpha = __expf(m_run - m_use);
                            l_run = l_run * alpha + psum;
#pragma unroll
                            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                        }
                        first = false;
                        m_run = m_new;
#pragma unroll
                        for (int sl = 0; sl < 2; ++sl) {
                            V8 pb;
#pragma unroll
                            for (int e = 0; e < 8; ++e) pb[e] = (T)s[8 * sl + e];
                            // the 4 + 4 keys of the slice this lane's P values belong to (rows clamped to the slot: a dead key, p = 0)
                            int r_lo = row0 + sub * 32 + 16 * sl + v_key0, r_hi = r_lo + 8;
                            if (PAIR) { r_lo = r_lo < 63 ? r_lo : 63; r_hi = r_hi < 63 ? r_hi : 63; }
#pragma unroll
                            for (int dt = 0; dt < 2; ++dt) {
                                const unsigned char* pl = vimg + r_lo * 128 + ((v_c16[dt] ^ ((r_lo >> 1) & 7)) << 4) + v_b8;
                                const unsigned char* ph = vimg + r_hi * 128 + ((v_c16[dt] ^ ((r_hi >> 1) & 7)) << 4) + v_b8;
                                union { qa_v4s s4[2]; V8 v; } vau;
                                vau.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) qa_v4s*)pl);
                                vau.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) qa_v4s*)ph);
                                const V8 va = vau.v;
                                o[dt] = E::mfma(va, pb, o[dt]);
                            }
                        }
                    }
                }
                const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
                const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
                if (PAIR || tq < g.N) {
                    T* op = reinterpret_cast<T*>(g.out) + (size_t)(PAIR ? grp * 256 + slot * 64 + tq : smp * g.N + tq) * LD + head * 64;
                    // a lane holds dimensions 8 g4 + 4 h .. + 3 of its query per quad g4: lanes l and l + 32 exchange one quad of every
                    // pair (v_permlane32_swap) so that each owns 8 consecutive dimensions -> 4 stores of 16 bytes instead of 8 of 8
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int k2 = 0; k2 < 2; ++k2) {
                            union { V4 v; unsigned u[2]; } q0, q1;
                            q0.v[0] = cvt16<T>(o[dt][8 * k2 + 0] * inv); q0.v[1] = cvt16<T>(o[dt][8 * k2 + 1] * inv);
                            q0.v[2] = cvt16<T>(o[dt][8 * k2 + 2] * inv); q0.v[3] = cvt16<T>(o[dt][8 * k2 + 3] * inv);
                            q1.v[0] = cvt16<T>(o[dt][8 * k2 + 4] * inv); q1.v[1] = cvt16<T>(o[dt][8 * k2 + 5] * inv);
                            q1.v[2] = cvt16<T>(o[dt][8 * k2 + 6] * inv); q1.v[3] = cvt16<T>(o[dt][8 * k2 + 7] * inv);
                            // swap: (q0 of the upper half) <-> (q1 of the lower half): lanes < 32 end with quads (2 k2, h = 0 | 1), lanes >= 32
                            // with quads (2 k2 + 1, h = 0 | 1)
                            const auto w0 = __builtin_amdgcn_permlane32_swap(q0.u[0], q1.u[0], false, false);
                            const auto w1 = __builtin_amdgcn_permlane32_swap(q0.u[1], q1.u[1], false, false);
                            const uint4 st16 = make_uint4(w0[0], w1[0], w0[1], w1[1]);
                            *reinterpret_cast<uint4*>(op + dt * 32 + 16 * k2 + 8 * h) = st16;
                        }
                }
            }
        }
        if (!has_next) break;
        lds_done_bar();                                           // the images are consumed: buffer 1 and the statistics' place are free
        L = Ln; grp = grp_n; head = head_n;
        stage_a(1, Ab + 2 * G_BK);                                // K-step 1 of the new tile (`ha` is already this tile's)
        stage_stats(grp);
    }
}

// uniform short sequences (a key-padding mask over them is fine), LayerNorm fold operands present
bool qkv_attn_eligible(int B, int N, int dtype, const void* stats_in, const void* colsum, const void* bias) {
    return (dtype == BG_BF16 || dtype == BG_F16) && B > 0 && N >= 2 && N <= 64 && (N & 1) == 0 && stats_in && colsum && bias &&
           (((size_t)stats_in) & 15) == 0 && (size_t)B * N * BG_D_MODEL * 2 < 0xffffffffull;
}

// One 256-slot tile per CU and round.  Measured against GEMM + attention (profiles/r04/qkv_attn_small_batches.log): the fused launch
// wins from 24 tiles (16 samples x 30 tokens: 16 vs 25 us) through one full round (0.56-0.72) and again from a round and a half on
// (0.68-0.83); only a second round that is less than about a third full is a wash (288 / 336 tiles: 1.03) -- left to the two launches.
bool qkv_attn_worthwhile(int B, int N) {
    const int spt = N <= 32 ? 8 : 4;
    const int tiles = (B + spt - 1) / spt * BG_N_HEAD;
    return tiles <= 256 || tiles >= 352;
}

// the XCD-pinned walk needs whole octets of workgroups (block b runs on XCD b % 8) and pays from two rounds of tiles on: below that the
// XCDs' shares (a quarter of the sample groups x 6 heads each) are too uneven (37 samples x 60 tokens: 20.7 -> 34.6 us).  bg_tune key
// 14: 1 = never, 2 = wherever the grid allows (the bit-equality tests).
static int qkv_attn_pinned_walk(int tiles, int grid) {
    const int t = g_tune[TUNE_QKV_WALK];
    if (t == 1 || (grid & 7) != 0) return 0;
    return (t == 2 || tiles >= 512) ? 1 : 0;
}

int qkv_attention_launch(const QkvAttnArgs& g, int dtype, hipStream_t s) {
    const int S = g.N <= 32 ? 32 : 64, spt = 256 / S;
    const int tiles = (g.B + spt - 1) / spt * BG_N_HEAD;
    if (tiles <= 0) return 0;
    const double rows = (double)g.B * g.N;
    ProfScope prof(PK_QKV_ATTN, 2.0 * rows * BG_D_MODEL * 3 * BG_D_MODEL + 4.0 * BG_N_HEAD * (double)g.B * g.N * g.N * BG_D_HEAD,
                   rows * (2.0 * BG_D_MODEL * 2 + FOLD_PARTS * 8.0) + 2.0 * 3 * BG_D_MODEL * BG_D_MODEL + 2 * 4.0 * 3 * BG_D_MODEL, s);
    const int grid = tiles < 256 ? tiles : 256;
    const bool f16 = dtype == BG_F16;
    QkvAttnArgs ge = g;
    ge.pin = qkv_attn_pinned_walk(tiles, grid);
#define QA_LAUNCH(F, SS, MK, D) hipLaunchKernelGGL((qkv_attn_kernel<F, SS, MK, D>), dim3(grid), dim3(512), 0, s, ge)
#define QA_PICK(MK, D)                                                                                     \
    do {                                                                                                   \
        if (S == 64) { if (f16) QA_LAUNCH(true, 64, MK, D); else QA_LAUNCH(false, 64, MK, D); }           \
        else { if (f16) QA_LAUNCH(true, 32, MK, D); else QA_LAUNCH(false, 32, MK, D); }                   \
    } while (0)
    if (g.dbg) {                                                   // (tests: the q|k|v image is written out as well)
        if (g.key_pad) QA_PICK(true, true); else QA_PICK(false, true);
    } else {
        if (g.key_pad) QA_PICK(true, false); else QA_PICK(false, false);
    }
#undef QA_PICK
#undef QA_LAUNCH
    return launch_status("qkv_attn");
}

// slot-packed ragged batch (compact.hip: compact_rows_paired): *m_dev rows in 64-row slots of one or two samples, at most
// `slot_bound` slots; m_stats = the row stride of the statistics partials (the workspace's row capacity)
int qkv_attention_paired(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in, void* out,
                         void* qkv_dbg, const int* m_dev, const int* slot_desc, int slot_bound, int m_stats, int dtype, float ln_eps,
                         hipStream_t s, double rows_hint, double pairs_hint) {
    if (slot_bound <= 0) return 0;
    QkvAttnArgs g{x_hi, w_qkv, bias, colsum, stats_in, out, qkv_dbg, nullptr, slot_bound, 64, m_stats, ln_eps};
    g.m_dev = m_dev;
    g.slot_desc = slot_desc;
    const int tiles = (slot_bound + 3) / 4 * BG_N_HEAD;           // upper bound: the kernel reads the row count on the device
    g.pin = qkv_attn_pinned_walk(tiles, tiles < 256 ? tiles : 256);
    // (opt-in profiler only: executed rows and attention pairs = sum over the samples of n^2, from the caller's host-side estimates;
    //  without them the slot bound -- every slot full, one sample of 64 tokens each)
    const double rows = rows_hint > 0 ? rows_hint : 64.0 * slot_bound;
    const double pairs = pairs_hint > 0 ? pairs_hint : 64.0 * rows;
    ProfScope prof(PK_QKV_ATTN, 2.0 * rows * BG_D_MODEL * 3 * BG_D_MODEL + 4.0 * BG_N_HEAD * pairs * BG_D_HEAD,
                   rows * (2.0 * BG_D_MODEL * 2 + FOLD_PARTS * 8.0) + 2.0 * 3 * BG_D_MODEL * BG_D_MODEL + 2 * 4.0 * 3 * BG_D_MODEL, s);
    const int grid = tiles < 256 ? tiles : 256;
    if (dtype == BG_F16) {
        if (qkv_dbg) hipLaunchKernelGGL((qkv_attn_kernel<true, 64, false, true, true>), dim3(grid), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((qkv_attn_kernel<true, 64, false, false, true>), dim3(grid), dim3(512), 0, s, g);
    } else {
        if (qkv_dbg) hipLaunchKernelGGL((qkv_attn_kernel<false, 64, false, true, true>), dim3(grid), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((qkv_attn_kernel<false, 64, false, false, true>), dim3(grid), dim3(512), 0, s, g);
    }
    return launch_status("qkv_attn_paired");
}

int qkv_attention(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in, void* out,
                  const uint8_t* key_pad, int B, int N, int dtype, float ln_eps, hipStream_t s) {
    const QkvAttnArgs g{x_hi, w_qkv, bias, colsum, stats_in, out, nullptr, key_pad, B, N, B * N, ln_eps};
    return qkv_attention_launch(g, dtype, s);
}

}  // namespace bg

extern "C" int bg_qkv_attn_fwd(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in,
                               const uint8_t* key_pad, void* out, void* qkv_dbg, int B, int N, int dtype, float ln_eps,
                               bg_stream_t stream) {
    BG_REQUIRE(x_hi && w_qkv && out && B >= 0 && N >= 0, BG_E_ARG, "bg_qkv_attn_fwd: null pointer or negative size");
    BG_REQUIRE(bg::qkv_attn_eligible(B, N, dtype, stats_in, colsum, bias), BG_E_SHAPE,
               "bg_qkv_attn_fwd: needs 16-bit operands, an even N in [2, 64], LayerNorm-fold statistics / column sums / bias (16-byte aligned)");
    BG_REQUIRE(((uintptr_t)x_hi & 15) == 0 && ((uintptr_t)w_qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
               ((uintptr_t)bias & 15) == 0 && ((uintptr_t)colsum & 15) == 0, BG_E_ALIGN, "bg_qkv_attn_fwd: 16-byte alignment");
    bg::QkvAttnArgs g{x_hi, w_qkv, bias, colsum, stats_in, out, qkv_dbg, key_pad, B, N, B * N, ln_eps};
    return bg::qkv_attention_launch(g, dtype, (hipStream_t)stream);
}

extern "C" int bg_qkv_attn_paired_fwd(const void* x_hi, const void* w_qkv, const float* bias, const float* colsum, const float* stats_in,
                                      void* out, void* qkv_dbg, const int* m_dev, const int* slot_desc, int slot_bound, int m_stats,
                                      int dtype, float ln_eps, bg_stream_t stream) {
    BG_REQUIRE(x_hi && w_qkv && out && m_dev && slot_desc && slot_bound > 0 && m_stats >= 64 * slot_bound, BG_E_ARG,
               "bg_qkv_attn_paired_fwd: null pointer, or a statistics stride below 64 rows per slot");
    BG_REQUIRE((dtype == BG_BF16 || dtype == BG_F16) && stats_in && colsum && bias && (m_stats & 1) == 0, BG_E_SHAPE,
               "bg_qkv_attn_paired_fwd: 16-bit operands with LayerNorm-fold statistics / column sums / bias");
    BG_REQUIRE(((uintptr_t)x_hi & 15) == 0 && ((uintptr_t)w_qkv & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0 &&
               ((uintptr_t)colsum & 15) == 0 && ((uintptr_t)stats_in & 15) == 0, BG_E_ALIGN, "bg_qkv_attn_paired_fwd: 16-byte alignment");
    // (the kernel forms 32-bit byte offsets row * 1536 over the slot rows)
    BG_REQUIRE((size_t)64 * slot_bound * BG_D_MODEL * 2 < 0xffffffffull, BG_E_SHAPE, "bg_qkv_attn_paired_fwd: more than 43 690 slots");
    return bg::qkv_attention_paired(x_hi, w_qkv, bias, colsum, stats_in, out, qkv_dbg, m_dev, slot_desc, slot_bound, m_stats, dtype, ln_eps,
                                    (hipStream_t)stream, 0.0, 0.0);
}
