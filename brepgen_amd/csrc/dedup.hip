// On-device bounding-box de-duplication between the stages of the cascade (sample.py:159-183 and 242-261): the only
// host round-trips inside the reference's denoising path (numpy loops, O(B*S*E^2)).  Same greedy, order-dependent
// algorithm, same float32 arithmetic, bit-identical decisions:
//   keep = [box_0]; for every box (box_0 included): drop it if max|kept - box| < thr for ANY kept box, comparing also
//   against the box with its two corners swapped; else append.
// One wave64 per sample (faces) or per (sample, face) (edges); lane l holds kept box l (and l+64), the candidate is
// broadcast, one ballot per candidate.  Integer/compare work, no FLOPs to speak of: latency-bound and tiny.
#include "bg_common.h"
#include <math.h>

namespace bg {

constexpr int DD_MAX = 128;                       // max boxes per group (ABC: 100 faces after doubling)

__device__ __forceinline__ bool same_box(const float* k, const float* c, float thr) {
    float d = 0.f, r = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        d = fmaxf(d, fabsf(k[i] - c[i]));
        r = fmaxf(r, fabsf(k[i] - c[(i + 3) % 6]));        // corners swapped: bbox[::-1]
    }
    return d < thr || r < thr;
}

// faces: in [B,S,6] -> out [B,S,6] = kept boxes (np.round(.,4)) left-aligned, zero padded; mask[b,s] = 1 for padding
__global__ __launch_bounds__(64) void dedup_surfaces_kernel(const float* __restrict__ in, float thr, float* __restrict__ out,
                                                            uint8_t* __restrict__ mask, int S) {
    __shared__ float keep[DD_MAX][6];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* src = in + (size_t)b * S * 6;
    int nkeep = 0;
    for (int i = 0; i < S; ++i) {
        float c[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) c[j] = rintf(src[i * 6 + j] * 10000.0f) / 10000.0f;   // np.round(x, 4) in float32
        if (i == 0) {                                   // non_repeat = bboxes[:1]
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) keep[0][j] = c[j];
            }
            nkeep = 1;
            __syncthreads();
        }
        bool hit = false;
        for (int k = lane; k < nkeep; k += 64) hit = hit || same_box(keep[k], c, thr);
        if (!__any(hit)) {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) keep[nkeep][j] = c[j];
            }
            ++nkeep;
            __syncthreads();
        }
    }
    for (int i = lane; i < S * 6; i += 64) out[(size_t)b * S * 6 + i] = (i / 6 < nkeep) ? keep[i / 6][i % 6] : 0.f;
    for (int i = lane; i < S; i += 64) mask[(size_t)b * S + i] = i >= nkeep;
}

// edges: edge_pos [B,S,E,6], surf_mask [B,S] (1 = padded face) -> edge_mask [B,S,E] (1 = padded face or duplicate edge).
// The reference indexes the output row by the POSITION of the face among the valid faces (sample.py:246-257).
__global__ __launch_bounds__(64) void dedup_edges_kernel(const float* __restrict__ ep, const uint8_t* __restrict__ smask, float thr,
                                                         uint8_t* __restrict__ emask, int S, int E) {
    __shared__ float keep[DD_MAX][6];
    const int b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
    const uint8_t* sm = smask + (size_t)b * S;
    // every row starts as the broadcast surface mask; rows are then overwritten for the valid faces, by position
    int n_valid = 0, pos = 0;
    for (int i = 0; i < S; ++i) {
        if (!sm[i]) { if (i < s) ++pos; ++n_valid; }
    }
    uint8_t* row_default = emask + ((size_t)b * S + s) * E;
    if (s >= n_valid)                                   // rows that no valid face writes keep the surface-mask value
        for (int e = lane; e < E; e += 64) row_default[e] = sm[s];
    if (sm[s]) return;                                  // padded face: contributes nothing
    uint8_t* row = emask + ((size_t)b * S + pos) * E;
    const float* src = ep + (((size_t)b * S + s) * E) * 6;
    int nkeep = 0;
    for (int i = 0; i < E; ++i) {
        float c[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) c[j] = src[i * 6 + j];
        if (i == 0) {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) keep[0][j] = c[j];
            }
            nkeep = 1;
            __syncthreads();
        }
        bool hit = false;
        for (int k = lane; k < nkeep; k += 64) hit = hit || same_box(keep[k], c, thr);
        const bool dup = __any(hit);
        if (!dup) {
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < 6; ++j) keep[nkeep][j] = c[j];
            }
            ++nkeep;
            __syncthreads();
        }
        // duplicates become True, the rest keeps the broadcast surface-mask value of that row; edgeM[.., 0] = False
        if (lane == 0) row[i] = (i == 0) ? 0 : (dup ? 1 : sm[pos]);
    }
}

}  // namespace bg

extern "C" int bg_dedup_surfaces(const float* surf_pos, float threshold, float* pos_out, uint8_t* mask_out, int B, int S,
                                 bg_stream_t stream) {
    BG_REQUIRE(surf_pos && pos_out && mask_out && B > 0 && S > 0, BG_E_ARG, "bg_dedup_surfaces: bad arguments");
    BG_REQUIRE(S <= bg::DD_MAX, BG_E_SHAPE, "bg_dedup_surfaces: at most %d faces per sample", bg::DD_MAX);
    hipLaunchKernelGGL(bg::dedup_surfaces_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, surf_pos, threshold, pos_out, mask_out, S);
    return bg::launch_status("dedup_surfaces");
}

extern "C" int bg_dedup_edges(const float* edge_pos, const uint8_t* surf_mask, float threshold, uint8_t* edge_mask, int B, int S,
                              int E, bg_stream_t stream) {
    BG_REQUIRE(edge_pos && surf_mask && edge_mask && B > 0 && S > 0 && E > 0, BG_E_ARG, "bg_dedup_edges: bad arguments");
    BG_REQUIRE(E <= bg::DD_MAX && B <= 65535, BG_E_SHAPE, "bg_dedup_edges: at most %d edges per face", bg::DD_MAX);
    hipLaunchKernelGGL(bg::dedup_edges_kernel, dim3(S, B), dim3(64), 0, (hipStream_t)stream, edge_pos, surf_mask, threshold,
                       edge_mask, S, E);
    return bg::launch_status("dedup_edges");
}
