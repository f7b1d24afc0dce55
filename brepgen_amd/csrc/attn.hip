// Masked multi-head self-attention of the denoisers (12 heads x 64, scale 1/8 folded into q by the weight
// packer, additive -inf key-padding mask; network.py:1076-1078 -> torch MHA slow path -> SDPA).
//
// bf16 kernel (MFMA, flash-style online softmax; any N):
//   * one workgroup = 4 waves = one sample b, 4/WPH heads, WPH*32 queries per head; each wave owns 32 queries
//     of one head and walks the keys in tiles of 64;
//   * scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16 with A = K rows from LDS, B = Q rows
//     held in registers), so every lane holds 16 keys of ONE query: the softmax row reductions are in-lane plus a
//     single lane <-> lane+32 exchange, and the running max / sum / rescale factors are lane-local to the O^T
//     accumulator columns as well;
//   * P^T is fed straight back as the B operand of O^T = V^T P^T: the k-slot order of the MFMA is a free
//     permutation as long as both operands agree, so the A operand (V^T rows from LDS) is simply read in the key
//     order the lane's score registers already have (two 8-byte reads) -- no cross-lane shuffle of P at all;
//   * K tile in LDS: 64 keys x 128 B, LDS-DMA + 16-byte XOR swizzle (same scheme as the GEMM);
//     V tile in LDS: transposed [64 d][64 keys], row stride 68 bf16 so the 32 d-rows a half-wave reads with
//     ds_read_b64 fall on 32 distinct bank pairs.
// fp32 kernel: exact VALU restatement for the fp32 parity mode (not a performance path).
#include "bg_common.h"
#include <math.h>

namespace bg {

constexpr int QKV_LD = 3 * BG_D_MODEL;     // 2304
constexpr int VS = 68;                     // V^T row stride (bf16 elements)
constexpr int HEAD_LDS = 64 * 128 + 64 * VS * 2;   // K tile + V^T tile bytes per head slot = 16896

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <bool F16> struct AElem;
template <> struct AElem<false> {
    using T = __bf16; using V8 = bf16x8; using V4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AElem<true> {
    using T = _Float16; using V8 = f16x8; using V4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <int WPH, bool F16>
__global__ __launch_bounds__(256, WPH == 1 ? 2 : 4) void attn16_kernel(const void* __restrict__ qkv_, const uint8_t* __restrict__ key_pad,
                                                        void* __restrict__ out_, int B, int N,
                                                        const int* __restrict__ offsets) {
    using E = AElem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    const T* __restrict__ qkv = reinterpret_cast<const T*>(qkv_);
    T* __restrict__ out = reinterpret_cast<T*>(out_);
    constexpr int HPW = 4 / WPH;               // heads per workgroup
    constexpr int KPI = 8 / WPH;               // K-tile DMA instructions per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[HPW * HEAD_LDS + 256];
    float* mb = reinterpret_cast<float*>(lds + HPW * HEAD_LDS);   // additive mask bias of the 64 keys of a tile

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = wave / WPH, qt = wave % WPH;
    const int b = blockIdx.z;
    const int head = blockIdx.y * HPW + hl;
    const int q0 = (blockIdx.x * WPH + qt) * 32;
    unsigned char* ktile = lds + hl * HEAD_LDS;
    T* vt = reinterpret_cast<T*>(ktile + 64 * 128);
    // compacted batch: sample b owns rows offsets[b] .. offsets[b+1]-1, every one a valid key (no mask)
    size_t row_base = (size_t)b * N;
    if (offsets) {
        row_base = (size_t)offsets[b];
        N = offsets[b + 1] - offsets[b];
        key_pad = nullptr;
        if ((int)(blockIdx.x * WPH * 32) >= N) return;            // uniform per workgroup (also N == 0), before any barrier
    }
    const T* base = qkv + row_base * QKV_LD + head * 64;

    // Q fragments (B operand of S^T = K Q^T): lane = (query l&31, k-chunk h) for each 16-wide slice of d
    V8 qf[4];
    {
        int qrow = q0 + (lane & 31);
        qrow = qrow < N ? qrow : N - 1;
        const T* qp = base + (size_t)qrow * QKV_LD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + (ks * 2 + h) * 8);
    }
    // K fragment read offsets (A operand): key row l&31 (+32 for the second sub-tile)
    int k_off[2], k_sw[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int row = sub * 32 + (lane & 31);
        k_off[sub] = row * 128;
        k_sw[sub] = (row >> 1) & 7;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nkt = (N + 63) / 64;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                                   // previous tile fully consumed
        // ---- stage K (LDS-DMA, swizzled source) ----
#pragma unroll
        for (int j = 0; j < KPI; ++j) {
            const int q = qt * KPI + j;
            const int row = q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            int key = kt * 64 + row;
            key = key < N ? key : N - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(base + (size_t)key * QKV_LD + BG_D_MODEL + c * 8),
                (__attribute__((address_space(3))) void*)(ktile + q * 1024), 16, 0, 0);
        }
        // ---- stage V transposed ----
#pragma unroll
        for (int j = 0; j < KPI; ++j) {
            const int idx = (qt * KPI + j) * 64 + lane;
            const int row = idx >> 3, dc = idx & 7;
            int key = kt * 64 + row;
            key = key < N ? key : N - 1;
            const V8 v = *reinterpret_cast<const V8*>(base + (size_t)key * QKV_LD + 2 * BG_D_MODEL + dc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(dc * 8 + e) * VS + row] = v[e];
        }
        if (tid < 64) {
            const int key = kt * 64 + tid;
            const bool dead = key >= N || (key_pad != nullptr && key_pad[row_base + key] != 0);
            mb[tid] = dead ? -INFINITY : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (kt * 64 + sub * 32 >= N) break;            // uniform: sub-tile entirely past the last key
            // ---- S^T = K Q^T over d = 64 ----
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const V8 kf = *reinterpret_cast<const V8*>(ktile + k_off[sub] + (((ks * 2 + h) ^ k_sw[sub]) << 4));
                s = E::mfma(kf, qf[ks], s);
            }
            // register r <-> key sub*32 + (r&3) + 8*(r>>2) + 4*h of query (lane & 31)
            float mloc = -INFINITY;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 mbv = *reinterpret_cast<const float4*>(&mb[sub * 32 + 8 * g4 + 4 * h]);
                s[4 * g4 + 0] += mbv.x; s[4 * g4 + 1] += mbv.y; s[4 * g4 + 2] += mbv.z; s[4 * g4 + 3] += mbv.w;
                mloc = fmaxf(mloc, fmaxf(fmaxf(s[4 * g4 + 0], s[4 * g4 + 1]), fmaxf(s[4 * g4 + 2], s[4 * g4 + 3])));
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            const float m_new = fmaxf(m_run, mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __expf(m_run - m_use);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __expf(s[r] - m_use);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            // ---- O^T += V^T P^T : two 16-key slices, two 32-row d tiles ----
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                V8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (T)s[8 * sl + e];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const T* vrow = vt + (dt * 32 + (lane & 31)) * VS + sub * 32 + 16 * sl + 4 * h;
                    const V4 lo = *reinterpret_cast<const V4*>(vrow);
                    const V4 hi = *reinterpret_cast<const V4*>(vrow + 8);
                    V8 va;
                    va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
                    va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
                    o[dt] = E::mfma(va, pb, o[dt]);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int q = q0 + (lane & 31);
    if (q < N) {
        T* op = out + (row_base + q) * BG_D_MODEL + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                V4 pk;
                pk[0] = cvt16<T>(o[dt][4 * g4 + 0] * inv); pk[1] = cvt16<T>(o[dt][4 * g4 + 1] * inv);
                pk[2] = cvt16<T>(o[dt][4 * g4 + 2] * inv); pk[3] = cvt16<T>(o[dt][4 * g4 + 3] * inv);
                *reinterpret_cast<V4*>(op + d) = pk;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Long sequences (N > 64: the edge nets, 1800 / 2400 / 4000 tokens, where attention is 40-60 % of the FLOPs).
//
//   * one workgroup = 4 waves = 128 queries of one (sample, head); 1-D grid walked XCD-aware so the ~15-32 query blocks
//     of a (sample, head) run on ONE XCD and its K/V (0.5-1 MB) is fetched into that L2 once;
//   * K and V tiles (64 keys x 128 B each) arrive by LDS-DMA into a 2-stage ring: tile t+1 is in flight while tile t is
//     computed, counted by vmcnt, ONE barrier per tile; both images are row-major with a 16-byte XOR swizzle applied on
//     the DMA source address (K: chunk ^= (key >> 1) & 7 for the ds_read_b128 fragment reads; V: chunk ^= ((key >> 1) & 1)
//     << 2, i.e. 32-byte granularity, for the transpose reads);
//   * V is consumed straight from its row-major image with ds_read_b64_tr_b16 (the hardware transpose read): a lane
//     supplies the address of 4 consecutive d of one key and receives 4 consecutive KEYS of one d -- the V^T fragment
//     of O^T = V^T P^T -- so the scalar transposing LDS writes of the short-sequence kernel are gone;
//   * scores transposed as in the short kernel (S^T = K Q^T: a lane owns 2 x 16 keys of ONE query), both 32-key
//     sub-tiles of a tile share one softmax pass: max by v_max3 chains + one v_permlane32_swap, exp as a single
//     v_fma + v_exp_f32 per score (base-2, log2(e) folded into the fma), the O / l rescale only when the running max
//     of some query of the wave grew by more than 3 (deferred maximum: P <= e^3 in the meantime, harmless in 16-bit
//     floating point whose relative precision does not depend on scale);
//   * key tiles beyond the last valid key of the sample are never visited; the mask bias is applied only in tiles that
//     contain a masked key (a per-tile flag), so compacted (variable-length) batches pay for masking in their tail tile only.
// MFMA-bound in principle; with d_head = 64 the softmax VALU work per MFMA is twice that of a d = 128 head, which is
// what bounds the achievable fraction of the matrix peak here.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short v4s_t;

// LDS-DMA issued from inline asm (1 KiB per wave-instruction: lane l's 16 bytes land at lds_wave_base + 16 l).  hipcc makes
// the first ds_read_b64_tr_b16 after a *builtin* LDS-DMA wait for vmcnt(0) -- it cannot tell that the transpose read and the
// in-flight DMA touch different ring stages -- which would serialise the next tile's fetch with this tile's P V product.
// Hidden in asm, the DMA is ordered by hand: counted vmcnt at the loop head, then the workgroup barrier, then the reads.
__device__ __forceinline__ void dma16_asm(const void* gsrc, void* lds_wave_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// value of `v` in lane ^ 32, by one v_permlane32_swap (pure VALU, no LDS crossbar)
__device__ __forceinline__ float other_half(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b = a;
    // hipcc (ROCm 7.2) folds the two results of permlane32_swap(x, x) as if they were equal -- max(r0, r1) became r0 and
    // half of every query's keys dropped out of the running maximum; an empty asm keeps the operands distinct values
    asm volatile("" : "+v"(b));
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    // r[0] = (a.lo, b.lo), r[1] = (a.hi, b.hi): lanes < 32 find the other half's value in r[1], lanes >= 32 in r[0]
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

constexpr int AL_MAX_N = 4096;       // longest sequence the long kernel stages a mask for (ABC edge nets: 4000)

template <bool F16, bool MASKED>      // MASKED: a key-padding mask is given (dense execution); otherwise only keys >= N are dead
__global__ __launch_bounds__(256, 3) void attn16_long_kernel(const void* __restrict__ qkv_, const uint8_t* __restrict__ key_pad,
                                                             void* __restrict__ out_, int B, int N, int nqb,
                                                             const int* __restrict__ offsets, int eighths) {
    using E = AElem<F16>;
    using T = typename E::T;
    using V8 = typename E::V8;
    using V4 = typename E::V4;
    constexpr int STAGE = 2 * 64 * 128;                  // K tile + V tile
    // ONE LDS object (a second one makes hipcc drain the DMA queue before every LDS read): ring | key flags | tile flags
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE + AL_MAX_N + 64 + 16];
    unsigned char* kp = lds + 2 * STAGE;                 // [AL_MAX_N] 1 = this key is masked / beyond the sequence
    unsigned char* tflag = kp + AL_MAX_N;                // [64] tile t contains such a key
    int* nk_sh = reinterpret_cast<int*>(tflag + 64);     // [4] per-wave "last valid key + 1"
    const T* __restrict__ qkv = reinterpret_cast<const T*>(qkv_);
    T* __restrict__ out = reinterpret_cast<T*>(out_);

    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware walk (block id % 8 = XCD): the query blocks of one (sample, head) UNIT run back to back on one XCD -- they share
    // that unit's K / V in its L2 -- and the units go round-robin over the XCDs, so that every XCD gets 1.5 heads of EVERY sample.
    // (Until round 6 each XCD walked a contiguous eighth of the batch: with ragged samples the XCDs' shares of sum n_b^2 differed
    // by +- 15 % at the bench's edge batches and the launch lasted as long as the heaviest eighth.)
    int unit = ((int)(blockIdx.x >> 3) / nqb) * 8 + (int)(blockIdx.x & 7);
    const int qb = (int)(blockIdx.x >> 3) % nqb;
    if (eighths) {                                       // (bg_tune key 20 = 1: the walk of rounds 2-5, for the in-process A/B)
        const int upx = (int)(gridDim.x >> 3) / nqb;     // units per XCD of the padded grid
        unit = (int)(blockIdx.x & 7) * upx + (int)(blockIdx.x >> 3) / nqb;
    }
    if (unit >= B * BG_N_HEAD) return;                   // (grid padded to whole rounds of 8 units)
    const int head = unit % BG_N_HEAD, b = unit / BG_N_HEAD;
    size_t row_base = (size_t)b * N;
    if (offsets) {                                       // compacted batch: rows offsets[b] .. offsets[b+1]-1, all valid keys
        row_base = (size_t)offsets[b];
        N = offsets[b + 1] - offsets[b];
        key_pad = nullptr;
    }
    if (qb * 128 >= N) return;                           // uniform per workgroup, before any barrier
    const T* base = qkv + row_base * QKV_LD + head * 64;
    const int q0 = qb * 128 + wave * 32;

    // ---- DMA: wave w moves K pieces 2w, 2w+1 and V pieces 2w, 2w+1 of a tile (a piece = 8 keys x 128 B) ----
    const int prow = lane >> 3, pch = lane & 7;
    auto issue_tile = [&](int t, int st) {
        unsigned char* kbase = lds + st * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int piece = wave * 2 + j;
            const int row = piece * 8 + prow;            // key row inside the tile
            int key = t * 64 + row;
            key = key < N ? key : N - 1;
            const T* src = base + (size_t)key * QKV_LD;
            const int kc = pch ^ ((row >> 1) & 7);
            const int vc = pch ^ (((row >> 1) & 1) << 2);
            dma16_asm(src + BG_D_MODEL + kc * 8, kbase + piece * 1024);
            dma16_asm(src + 2 * BG_D_MODEL + vc * 8, kbase + 64 * 128 + piece * 1024);
        }
    };
    // the first tile's fetch goes out before anything else, so its latency overlaps the rest of the prologue (key 0 is
    // fetched even if the whole sample turns out to be masked: harmless)
    issue_tile(0, 0);

    // ---- masked execution: key flags of the whole sample into LDS (once), last valid key, per-tile flags ----
    int n_keys = N;
    if (MASKED) {
        const int nkt_all = (N + 63) / 64;
        int last = 0;
        for (int k = tid; k < nkt_all * 64; k += 256) {
            const bool dead = k >= N || key_pad[row_base + k] != 0;
            kp[k] = dead ? 1 : 0;
            last = dead ? last : k + 1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
        if (lane == 0) nk_sh[wave] = last;
        __syncthreads();
        n_keys = max(max(nk_sh[0], nk_sh[1]), max(nk_sh[2], nk_sh[3]));
    }
    const int nkt = (n_keys + 63) / 64;                  // key tiles beyond the last valid key are never visited
    if (MASKED) {
        if (tid < nkt) {
            const uint4* w = reinterpret_cast<const uint4*>(kp + tid * 64);
            const uint4 a = w[0], b4 = w[1], c = w[2], d = w[3];
            tflag[tid] = ((a.x | a.y | a.z | a.w | b4.x | b4.y | b4.z | b4.w | c.x | c.y | c.z | c.w | d.x | d.y | d.z | d.w) != 0u) ? 1 : 0;
        }
        __syncthreads();                                 // tflag visible
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane = (query l & 31, k-chunk h) per 16-wide slice of d ----
    V8 qf[4];
    {
        int qrow = q0 + (lane & 31);
        qrow = qrow < N ? qrow : N - 1;
        const T* qp = base + (size_t)qrow * QKV_LD;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const V8*>(qp + (ks * 2 + h) * 8);
        // have hipcc wait for these loads HERE: a register load still pending at the loop head would make it drain the
        // whole VMEM queue -- including the next tile's DMA -- in every iteration of the tile loop
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // K fragment offsets (A operand of S^T): key row l & 31 of each 32-key sub-tile, 16-byte chunk ks*2+h, swizzled
    int k_off[2], k_sw[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int row = sub * 32 + (lane & 31);
        k_off[sub] = row * 128;
        k_sw[sub] = (row >> 1) & 7;
    }
    // V^T fragment addresses for the transpose read: 16-lane group g = lane >> 4 covers d columns (g & 1) * 16 .. +15 of
    // the 32-row d tile and the keys of k-chunk h = g >> 1; lane i of the group supplies the address of key (i >> 2),
    // d sub-chunk (i & 3) * 4 and receives 4 consecutive keys of d column i.
    const int gi = lane & 15, gg = lane >> 4;
    const int v_key0 = 4 * (gg >> 1) + (gi >> 2);        // + 32*sub + 16*sl (+ 8 for the second read)
    const int v_col = (gg & 1) * 16 + (gi & 3) * 4;      // + 32*dt   (16-bit elements)
    // byte offset of (key0 + 32*sub + 16*sl [+ 8], col + 32*dt) in the swizzled V image.  (key >> 1) & 1 depends only on
    // bit 1 of v_key0 for every such key, so the swizzle term is a per-lane constant
    const int v_swz = ((v_key0 >> 1) & 1) << 2;
    int v_off[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int col = dt * 32 + v_col;
        v_off[dt] = v_key0 * 128 + (((col >> 3) ^ v_swz) << 4) + (col & 7) * 2;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;

    for (int t = 0; t < nkt; ++t) {
        const int st = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my pieces of tile t (issued one tile ago) landed
        __builtin_amdgcn_s_barrier();                             // everybody's did; stage st^1 is free again
        if (t + 1 < nkt) issue_tile(t + 1, st ^ 1);
        const unsigned char* kt_ = lds + st * STAGE;
        const unsigned char* vt_ = kt_ + 64 * 128;
        if (q0 >= N) continue;        // a wave without a valid query (last, partly filled query block) only helps move the tiles

        // ---- S^T = K Q^T for both 32-key sub-tiles ----
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const V8 kf = *reinterpret_cast<const V8*>(kt_ + k_off[sub] + (((ks * 2 + h) ^ k_sw[sub]) << 4));
                s[sub] = E::mfma(kf, qf[ks], s[sub]);
            }
        }
        // register r of sub-tile sub <-> key sub*32 + (r&3) + 8*(r>>2) + 4*h of query (lane & 31)
        if (MASKED) {
            if (tflag[t]) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const unsigned w = *reinterpret_cast<const unsigned*>(kp + t * 64 + sub * 32 + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            s[sub][4 * g4 + e] = ((w >> (8 * e)) & 0xffu) ? -INFINITY : s[sub][4 * g4 + e];
                    }
            }
        } else if ((t + 1) * 64 > N) {                            // only the last tile of an unmasked sequence has dead keys
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    s[sub][r] = key >= N ? -INFINITY : s[sub][r];
                }
        }
        float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(fmaxf(mloc, s[0][r]), s[1][r]);
        mloc = fmaxf(mloc, other_half(mloc));                    // the query's other 32 keys live in lane ^ 32
        // deferred maximum: rescale only when some query's maximum grew by more than 3 (or at its first valid tile)
        if (!__all(mloc <= m_run + 3.0f)) {
            const float m_new = fmaxf(m_run, mloc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * LOG2E);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            m_run = m_new;
        }
        const float mc = ((m_run == -INFINITY) ? 0.f : m_run) * LOG2E;
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[sub][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][r], LOG2E, -mc));
                psum += s[sub][r];
            }
        l_run += psum;

        // ---- O^T += V^T P^T: per 16-key slice the P^T fragment is the lane's own score registers, the V^T fragment two
        // transpose reads in the same key order (keys 4h..4h+3 and 8+4h..8+4h+3 of the slice) ----
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                V8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (T)s[sub][8 * sl + e];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const unsigned char* pa = vt_ + v_off[dt] + (sub * 32 + sl * 16) * 128;
                    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)pa);
                    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(pa + 8 * 128));
                    union { v4s_t s4[2]; V8 v; } va;
                    va.s4[0] = lo; va.s4[1] = hi;
                    o[dt] = E::mfma(va.v, pb, o[dt]);
                }
            }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may outlive the workgroup (nkt == 0: tile 0's fetch)
    l_run += other_half(l_run);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    const int q = q0 + (lane & 31);
    if (q < N) {
        T* op = out + (row_base + q) * BG_D_MODEL + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                V4 pk;
                pk[0] = cvt16<T>(o[dt][4 * g4 + 0] * inv); pk[1] = cvt16<T>(o[dt][4 * g4 + 1] * inv);
                pk[2] = cvt16<T>(o[dt][4 * g4 + 2] * inv); pk[3] = cvt16<T>(o[dt][4 * g4 + 3] * inv);
                *reinterpret_cast<V4*>(op + d) = pk;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 attention: one wave per (b, head, query).  Scores for keys lane, lane+64, ... in registers/LDS,
// accurate expf, then lane = d for the P V product.  Parity mode only.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qkv,
                                                       const uint8_t* __restrict__ key_pad,
                                                       float* __restrict__ out, int B, int N,
                                                       const int* __restrict__ offsets) {
    extern __shared__ float dyn[];                     // 4 waves x N probabilities
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const int head = blockIdx.y, b = blockIdx.z;
    float* p = dyn + (size_t)wave * N;
    size_t row_base = (size_t)b * N;
    if (offsets) {                                     // compacted batch (see attn16_kernel)
        row_base = (size_t)offsets[b];
        N = offsets[b + 1] - offsets[b];
        key_pad = nullptr;
    }
    if (q >= N) return;                                // no block-level barriers below
    const float* base = qkv + row_base * QKV_LD + head * 64;
    const float* qp = base + (size_t)q * QKV_LD;
    float qv[64];
#pragma unroll
    for (int d = 0; d < 64; ++d) qv[d] = qp[d];
    float mloc = -INFINITY;
    for (int j = lane; j < N; j += 64) {
        float sc;
        if (key_pad != nullptr && key_pad[row_base + j] != 0) {
            sc = -INFINITY;
        } else {
            const float* kp = base + (size_t)j * QKV_LD + BG_D_MODEL;
            sc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) sc = fmaf(qv[d], kp[d], sc);
        }
        p[j] = sc;
        mloc = fmaxf(mloc, sc);
    }
    const float m = wave_max(mloc);
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float lsum = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float e = expf(p[j] - m_use);
        p[j] = e;
        lsum += e;
    }
    const float l = wave_sum(lsum);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float acc = 0.f;                                   // lane = d
    for (int j = 0; j < N; ++j) acc = fmaf(p[j], base[(size_t)j * QKV_LD + 2 * BG_D_MODEL + lane], acc);
    out[(row_base + q) * BG_D_MODEL + head * 64 + lane] = l > 0.f ? acc / l : 0.f;
}

int attention(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype, hipStream_t s,
              const int* offsets, double pairs_hint, double rows_hint) {
    if (B <= 0 || N <= 0) return 0;
    const double es = dtype == BG_F32 ? 4.0 : 2.0;
    // algorithmic: QK^T + PV over all heads, no mask discount; bytes: qkv read once, out written once
    // (compacted batch: pairs_hint = expected sum over samples of n_b^2, for the opt-in profiler only)
    const double pairs = (offsets && pairs_hint > 0) ? pairs_hint : (double)B * N * N;
    ProfScope prof(dtype == BG_F32 ? PK_ATTN_F32 : PK_ATTN_BF16, 4.0 * BG_N_HEAD * pairs * BG_D_HEAD,
                   es * ((offsets && rows_hint > 0) ? rows_hint : (double)B * N) * (QKV_LD + BG_D_MODEL), s);
    if (dtype == BG_BF16 || dtype == BG_F16) {
        const bool f16 = dtype == BG_F16;
        if (N <= 32) {
            const dim3 grid(1, BG_N_HEAD / 4, B);
            if (f16) hipLaunchKernelGGL((attn16_kernel<1, true>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
            else hipLaunchKernelGGL((attn16_kernel<1, false>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
        } else if (N <= 64) {
            const dim3 grid(1, BG_N_HEAD / 2, B);
            if (f16) hipLaunchKernelGGL((attn16_kernel<2, true>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
            else hipLaunchKernelGGL((attn16_kernel<2, false>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
        } else if (N > AL_MAX_N) {                                  // sequences beyond the long kernel's mask staging: the single-buffered kernel
            const dim3 grid((N + 127) / 128, BG_N_HEAD, B);
            if (f16) hipLaunchKernelGGL((attn16_kernel<4, true>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
            else hipLaunchKernelGGL((attn16_kernel<4, false>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, offsets);
        } else {
            const int nqb = (N + 127) / 128;
            const dim3 grid((unsigned)(nqb * 8 * ((B * BG_N_HEAD + 7) / 8)));
            const bool masked = key_pad != nullptr && offsets == nullptr;
            const int eighths = g_tune[TUNE_ATTN_WALK] == 1;
            if (f16 && masked) hipLaunchKernelGGL((attn16_long_kernel<true, true>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, nqb, offsets, eighths);
            else if (f16) hipLaunchKernelGGL((attn16_long_kernel<true, false>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, nqb, offsets, eighths);
            else if (masked) hipLaunchKernelGGL((attn16_long_kernel<false, true>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, nqb, offsets, eighths);
            else hipLaunchKernelGGL((attn16_long_kernel<false, false>), grid, dim3(256), 0, s, qkv, key_pad, out, B, N, nqb, offsets, eighths);
        }
        return launch_status("attn16");
    }
    if (dtype == BG_F32) {
        const size_t shm = (size_t)4 * N * sizeof(float);
        if (shm > 160 * 1024) {
            set_error("attn_f32: N=%d too long for the parity kernel", N);
            return BG_E_SHAPE;
        }
        if (shm > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipLaunchKernelGGL(attn_f32_kernel, dim3((N + 3) / 4, BG_N_HEAD, B), dim3(256), shm, s,
                           reinterpret_cast<const float*>(qkv), key_pad, reinterpret_cast<float*>(out), B, N, offsets);
        return launch_status("attn_f32");
    }
    set_error("attention: unsupported dtype %d", dtype);
    return BG_E_DTYPE;
}

}  // namespace bg

extern "C" int bg_attn_fwd(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype,
                           bg_stream_t stream) {
    BG_REQUIRE(qkv && out && B >= 0 && N >= 0, BG_E_ARG, "bg_attn_fwd: null pointer or negative size");
    BG_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, BG_E_ALIGN, "bg_attn_fwd: 16-byte alignment");
    return bg::attention(qkv, key_pad, out, B, N, dtype, (hipStream_t)stream, nullptr, 0.0);
}

extern "C" int bg_attn_varlen_fwd(const void* qkv, const uint8_t* key_pad, void* out, int B, int N, int dtype,
                                  const int* offsets, bg_stream_t stream) {
    BG_REQUIRE(qkv && out && B >= 0 && N >= 0, BG_E_ARG, "bg_attn_varlen_fwd: null pointer or negative size");
    BG_REQUIRE(!(offsets && key_pad), BG_E_ARG, "bg_attn_varlen_fwd: a compacted batch has no padded keys (key_pad must be NULL)");
    BG_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, BG_E_ALIGN, "bg_attn_varlen_fwd: 16-byte alignment");
    return bg::attention(qkv, key_pad, out, B, N, dtype, (hipStream_t)stream, offsets, 0.0);
}
