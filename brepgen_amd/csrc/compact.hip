// Valid-token compaction of a padded batch (variable-length execution of the denoisers).
//
// The reference pads every sample to max_face (x max_edge) tokens and masks the padding as attention KEYS only
// (network.py:1196, 1283, 1390 -> src_key_padding_mask); the padded tokens still run through all 12 layers and their
// outputs are thrown away (sample.py:284, 307-314 read valid rows only).  No op of the path mixes tokens except
// attention, and a padded key never reaches a valid query -- so the valid tokens of each sample can be packed into
// consecutive rows, every GEMM / LayerNorm runs on sum(valid) rows, attention runs per sample over its own rows, and
// the result is scattered back into the padded layout (padded positions: 0).
//
// mask [B, n_mask] uint8, 1 = padded; each entry covers `rep` consecutive tokens (EdgePosNet: one entry per face,
// rep = E).  Three tiny launches, all integer work, no host round trip: the row count stays on the device (offsets[B])
// and every consumer kernel reads it from there.
#include "bg_common.h"

namespace bg {

__global__ __launch_bounds__(256) void count_valid_kernel(const uint8_t* __restrict__ mask, int n_mask, int rep,
                                                          int* __restrict__ counts) {
    __shared__ int part[4];
    const uint8_t* m = mask + (size_t)blockIdx.x * n_mask;
    int c = 0;
    for (int i = threadIdx.x; i < n_mask; i += 256) c += m[i] == 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (part[0] + part[1] + part[2] + part[3]) * rep;
}

// exclusive scan of counts[0..B) -> offsets[0..B]; one block, B <= a few thousand
// rule (optional, P256_RULE_ENTRIES ints): how the GEMM launches of this evaluation split their rows between the 256 x 256 kernel
// and a 128 x 128 kernel, evaluated HERE once per row count -- the GEMM kernels then read one int instead of each carrying the rule
__global__ __launch_bounds__(1024) void scan_offsets_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets,
                                                            int* __restrict__ rule) {
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < B ? counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                       // Hillis-Steele inclusive scan
            const int t = threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < B) offsets[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[B] = carry;
    if (rule != nullptr && threadIdx.x < P256_RULE_ENTRIES) {
        const int nt = threadIdx.x >> 2 == 0 ? 3 : (threadIdx.x >> 2 == 1 ? 4 : 9);
        rule[threadIdx.x] = p256_rows(carry, nt, (threadIdx.x & 2) != 0, (threadIdx.x & 1) != 0);
    }
}

// src_row[offsets[b] + j] = padded-layout index of the j-th valid token of sample b (order preserved)
__global__ __launch_bounds__(256) void fill_rows_kernel(const uint8_t* __restrict__ mask, int n_mask, int rep,
                                                        const int* __restrict__ offsets, int* __restrict__ src_row) {
    __shared__ int wsum[4];
    __shared__ int running;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* m = mask + (size_t)b * n_mask;
    if (threadIdx.x == 0) running = offsets[b];
    __syncthreads();
    for (int base = 0; base < n_mask; base += 256) {
        const int i = base + threadIdx.x;
        const bool valid = i < n_mask && m[i] == 0;
        const unsigned long long bal = __ballot(valid);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        if (valid) {
            const int dst = running + (wbase + before) * rep;
            for (int e = 0; e < rep; ++e) src_row[dst + e] = (b * n_mask + i) * rep + e;
        }
        __syncthreads();
        if (threadIdx.x == 0) running += (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * rep;
        __syncthreads();
    }
}

// ---- slot-packed compaction (nets with <= 64 tokens per sample: the fused QKV + attention launch on ragged batches) ----------
// Every 64-row slot holds one or two WHOLE samples: rows [0, n_a) sample a, [n_a, n_a + n_b) sample b, the rest clones of a's first
// token (same src_row: every row-wise kernel computes the clone's bits exactly as the original's, the final scatter writes
// identical values to the same place).  Lengths are sorted (rank = number of samples that are shorter, or equally long with a
// smaller index: deterministic), then the shortest unpaired sample joins the longest one while their sum fits -- 0.96 of the slots
// filled for lengths ~ U{8..60} against 0.69 with one sample per 32 / 64-slot.
//   counts[b]          valid tokens of sample b (count_valid_kernel)
//   offsets[b]         first row of sample b in the slot-packed layout (samples without a valid token: 0, never used)
//   offsets[B]         = 64 * n_slots, the device-side row count every kernel reads
//   slot_desc[2 k]     n_a, slot_desc[2 k + 1] n_b of slot k;   slot_a[k] = sample a of slot k
// LDS of pair_slots_kernel: [B] shorts sorted | [B] bytes len | [B] bytes slen | a region that holds the per-wave length histograms
// of the ranking and afterwards pair_i [B] shorts | lim [B] shorts
static inline size_t pair_slots_lds_bytes(int B) {
    const size_t nw = (size_t)(B + 63) / 64, hist = nw * 66 * sizeof(short), walk = (size_t)4 * B;
    return (size_t)4 * B + (hist > walk ? hist : walk);
}

__global__ __launch_bounds__(1024) void pair_slots_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets,
                                                          int* __restrict__ slot_desc, int* __restrict__ slot_a, int* __restrict__ rule) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sh[];
    short* sorted = reinterpret_cast<short*>(sh);                 // [B] sample ids in ascending (length, index)
    unsigned char* len = reinterpret_cast<unsigned char*>(sorted + B);
    unsigned char* slen = len + B;                                // lengths in ascending order
    short* region = reinterpret_cast<short*>(sh + (size_t)4 * B);
    short* wh = region;                                           // [waves of 64 samples][66] length histogram -> exclusive prefix over the waves
    short* pair_i = region;                                       // (after the ranking) slot -> rank of its second sample, or -1
    short* lim = region + B;                                      // per rank: how many ranks may still pair with it (below)
    __shared__ int cum[66];                                       // cum[v + 1] = samples of length <= v
    __shared__ int n_slots_sh;
    const int lane = threadIdx.x & 63;
    const int nw = (B + 63) >> 6;
    if (threadIdx.x < 66) cum[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < nw * 66; i += 1024) wh[i] = 0;
    __syncthreads();
    // ---- stable counting rank: rank = #shorter + #equally long with a smaller index.  Inside a wave of 64 consecutive samples the
    // equally long ones are found with seven ballots (one per bit of the length); across waves by a prefix over per-wave histograms.
    // (Until round 5 every thread compared its sample with all B others: O(B^2) dependent LDS reads -- 40 us at B = 512, 4 ms at 8192.)
    auto same_length_before = [&](int i, int& li) -> int {
        li = i < B ? (int)len[i] : 127;                           // (no sample is 127 long: the lanes past B form their own group)
        unsigned long long match = ~0ull;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const unsigned long long bal = __ballot((li >> b) & 1);
            match &= ((li >> b) & 1) ? bal : ~bal;
        }
        const int before = __popcll(match & ((1ull << lane) - 1ull));
        if (before == 0 && i < B) wh[(i >> 6) * 66 + li] = (short)__popcll(match);       // (first of its group: the wave's count of that length)
        return before;
    };
    for (int base = 0; base < B; base += 1024) {
        const int i = base + threadIdx.x;
        if (i < B) {
            const int c = counts[i];
            len[i] = (unsigned char)c;
            atomicAdd(&cum[c + 1], 1);
        }
    }
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        int li;
        (void)same_length_before(base + threadIdx.x, li);
    }
    if (threadIdx.x == 0)
        for (int v = 1; v < 66; ++v) cum[v] += cum[v - 1];
    __syncthreads();
    if (threadIdx.x < 65) {                                       // exclusive prefix over the waves, per length
        int run = 0;
        for (int w = 0; w < nw; ++w) {
            const int t = wh[w * 66 + threadIdx.x];
            wh[w * 66 + threadIdx.x] = (short)run;
            run += t;
        }
    }
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int i = base + threadIdx.x;
        const int li = i < B ? (int)len[i] : 127;                 // (the same ballots again, without the histogram write)
        unsigned long long match = ~0ull;
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            const unsigned long long bal = __ballot((li >> b) & 1);
            match &= ((li >> b) & 1) ? bal : ~bal;
        }
        if (i < B) {
            const int rank = cum[li] + wh[(i >> 6) * 66 + li] + __popcll(match & ((1ull << lane) - 1ull));
            sorted[rank] = (short)i;
            slen[rank] = (unsigned char)li;
        }
    }
    __syncthreads();                                              // (the histograms are consumed: their place becomes pair_i / lim)
    // The walk: the longest unpaired sample (rank j, descending) takes the shortest unpaired one (rank i, ascending) while their sum
    // fits 64 -- i < j and slen[i] <= 64 - slen[j], i.e. (the ranks are sorted) i < lim[j] = min(j, #samples of length <= 64 - slen[j]).
    // lim[] does not depend on the walk, so it is computed by all threads, and the walk itself -- i += (i < lim[j]) -- is pure register
    // arithmetic on batches of 64 preloaded limits.  (Round 4 walked it with global stores and dependent LDS reads per iteration.)
    for (int j = threadIdx.x; j < B; j += 1024) {
        const int c = cum[64 - slen[j] + 1];
        lim[j] = (short)(j < c ? j : c);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // one wave, state in SCALAR registers: every lane fetches one limit of the next 64 ranks, the walk reads them back with
        // v_readlane and records its pairings in a 64-bit scalar mask, from which every lane derives the answer of its slot (a
        // single-lane VALU walk costs ~230 cycles per slot)
        int i = __builtin_amdgcn_readfirstlane(cum[1]), j = B - 1, k = 0;   // samples without a valid token (the first cum[1] ranks) own no rows
        while (i <= j) {
            const int jj = j - lane;
            const int lv = jj >= 0 ? (int)lim[jj] : 0;
            const int i0 = i;
            int n = 0;
            unsigned long long pm = 0ull;                         // bit u: slot k + u got a second sample
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const int m = __builtin_amdgcn_readlane(lv, u);
                const bool live = i <= j;                         // (once false it stays false: i never decreases, j only does)
                const bool paired = live && i < m;
                pm |= (unsigned long long)(paired ? 1 : 0) << u;
                i += paired ? 1 : 0;
                j -= live ? 1 : 0;
                n += live ? 1 : 0;
            }
            // slot k + lane: paired -> the rank it took = i0 + the number of pairings before it in this batch
            const int res = ((pm >> lane) & 1ull) ? i0 + __popcll(pm & ((1ull << lane) - 1ull)) : -1;
            if (lane < n) pair_i[k + lane] = (short)res;
            k += n;
        }
        if (lane == 0) {
            n_slots_sh = k;
            offsets[B] = 64 * k;
        }
    }
    __syncthreads();
    const int n_slots = n_slots_sh;
    for (int k = threadIdx.x; k < n_slots; k += 1024) {
        const int a = sorted[B - 1 - k], na = len[a];
        int nb = 0;
        if (pair_i[k] >= 0) {
            const int b = sorted[pair_i[k]];
            nb = len[b];
            offsets[b] = 64 * k + na;
        }
        offsets[a] = 64 * k;
        slot_desc[2 * k] = na; slot_desc[2 * k + 1] = nb; slot_a[k] = a;
    }
    for (int i = threadIdx.x; i < B; i += 1024)
        if (len[i] == 0) offsets[i] = 0;
    if (rule != nullptr && threadIdx.x < P256_RULE_ENTRIES) {
        const int nt = threadIdx.x >> 2 == 0 ? 3 : (threadIdx.x >> 2 == 1 ? 4 : 9);
        rule[threadIdx.x] = p256_rows(64 * n_slots, nt, (threadIdx.x & 2) != 0, (threadIdx.x & 1) != 0);
    }
}

// the rows of a slot behind its two samples: clones of the slot's first row (src_row[64 k] is sample a's first valid token)
__global__ __launch_bounds__(64) void fill_clones_kernel(const int* __restrict__ offsets_B, const int* __restrict__ slot_desc,
                                                         int* __restrict__ src_row) {
    const int k = blockIdx.x;
    if (64 * k >= *offsets_B) return;
    const int used = slot_desc[2 * k] + slot_desc[2 * k + 1];
    const int t = threadIdx.x;
    if (t >= used) src_row[64 * k + t] = src_row[64 * k];
}

// (rep == 1, n_mask <= 64.)  offsets [B + 1], src_row [64 B], slot_desc [2 B], slot_a [B]
int compact_rows_paired(const uint8_t* mask, int B, int n_mask, int* offsets, int* src_row, int* slot_desc, int* slot_a,
                        int* counts, hipStream_t s, int* rule) {
    // (pair_slots_kernel keeps sample indices in 16 bits and lengths in 8, and its dynamic LDS grows with B: guarded HERE, not only
    //  at the extern "C" entry, because the denoiser calls this function directly)
    BG_REQUIRE(B > 0 && B <= 8192 && n_mask > 0 && n_mask <= 64, BG_E_SHAPE, "compact_rows_paired: 1 .. 8192 samples of 1 .. 64 tokens");
    ProfScope prof(PK_MISC, 0.0, (double)B * n_mask * 6.0, s);
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, s, mask, n_mask, 1, counts);
    hipLaunchKernelGGL(pair_slots_kernel, dim3(1), dim3(1024), pair_slots_lds_bytes(B), s, counts, B, offsets, slot_desc, slot_a, rule);
    hipLaunchKernelGGL(fill_rows_kernel, dim3(B), dim3(256), 0, s, mask, n_mask, 1, offsets, src_row);
    hipLaunchKernelGGL(fill_clones_kernel, dim3(B), dim3(64), 0, s, offsets + B, slot_desc, src_row);
    return launch_status("compact_rows_paired");
}

int compact_rows(const uint8_t* mask, int B, int n_mask, int rep, int* offsets, int* src_row, hipStream_t s, int* rule) {
    ProfScope prof(PK_MISC, 0.0, (double)B * n_mask * (2.0 + 4.0 * rep), s);
    // counts live in src_row's tail?  no: keep it simple -- offsets[1..B] doubles as the count buffer before the scan
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, s, mask, n_mask, rep, offsets + 1);
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(1024), 0, s, offsets + 1, B, offsets, rule);
    hipLaunchKernelGGL(fill_rows_kernel, dim3(B), dim3(256), 0, s, mask, n_mask, rep, offsets, src_row);
    return launch_status("compact_rows");
}

}  // namespace bg

extern "C" int bg_compact_rows(const uint8_t* mask, int B, int n_mask, int rep, int* offsets, int* src_row, bg_stream_t stream) {
    BG_REQUIRE(mask && offsets && src_row, BG_E_ARG, "bg_compact_rows: null pointer");
    BG_REQUIRE(B > 0 && n_mask > 0 && rep > 0, BG_E_SHAPE, "bg_compact_rows: empty shape");
    return bg::compact_rows(mask, B, n_mask, rep, offsets, src_row, (hipStream_t)stream, nullptr);
}

extern "C" int bg_compact_rows_paired(const uint8_t* mask, int B, int n_mask, int* offsets, int* src_row, int* slot_desc, int* slot_a,
                                      int* counts, bg_stream_t stream) {
    BG_REQUIRE(mask && offsets && src_row && slot_desc && slot_a && counts, BG_E_ARG, "bg_compact_rows_paired: null pointer");
    BG_REQUIRE(B > 0 && B <= 8192 && n_mask > 0 && n_mask <= 64, BG_E_SHAPE, "bg_compact_rows_paired: 1 .. 8192 samples of 1 .. 64 tokens");
    return bg::compact_rows_paired(mask, B, n_mask, offsets, src_row, slot_desc, slot_a, counts, (hipStream_t)stream, nullptr);
}
