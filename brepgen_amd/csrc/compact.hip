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
__global__ __launch_bounds__(1024) void pair_slots_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets,
                                                          int* __restrict__ slot_desc, int* __restrict__ slot_a, int* __restrict__ rule) {
    // [B] sample ids in ascending length, [B] slot -> rank of its second sample (or -1), [B] lengths, [B] lengths in ascending order
    extern __shared__ unsigned char sh[];
    short* sorted = reinterpret_cast<short*>(sh);
    short* pair_i = sorted + B;
    unsigned char* len = reinterpret_cast<unsigned char*>(pair_i + B);
    unsigned char* slen = len + B;
    __shared__ int n_slots_sh;
    for (int i = threadIdx.x; i < B; i += 1024) len[i] = (unsigned char)counts[i];
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 1024) {
        const int li = len[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (len[j] < li) || (len[j] == li && j < i);
        sorted[rank] = (short)i;
        slen[rank] = (unsigned char)li;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // the two-pointer walk is inherently serial, so it touches nothing but LDS and keeps the next length of either pointer
        // loaded ahead of its use (slot k always takes rank B - 1 - k as its first sample); everything else is written out by all
        // threads below.  (Round 4 walked it with global stores and dependent LDS reads in every iteration: 63 us at B = 512.)
        int i = 0, j = B - 1, k = 0;
        while (i < B && slen[i] == 0) ++i;                        // samples without a valid token own no rows
        int li = i < B ? slen[i] : 0, li_next = i + 1 < B ? slen[i + 1] : 0;
        int lj = j >= 0 ? slen[j] : 0, lj_next = j >= 1 ? slen[j - 1] : 0;
        while (i <= j) {
            const bool paired = i < j && li + lj <= 64;
            pair_i[k] = paired ? (short)i : (short)-1;
            if (paired) {
                ++i;
                li = li_next;
                li_next = i + 1 < B ? slen[i + 1] : 0;
            }
            --j; ++k;
            lj = lj_next;
            lj_next = j >= 1 ? slen[j - 1] : 0;
        }
        n_slots_sh = k;
        offsets[B] = 64 * k;
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
    ProfScope prof(PK_MISC, 0.0, (double)B * n_mask * 6.0, s);
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, s, mask, n_mask, 1, counts);
    hipLaunchKernelGGL(pair_slots_kernel, dim3(1), dim3(1024), (size_t)6 * B, s, counts, B, offsets, slot_desc, slot_a, rule);
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
