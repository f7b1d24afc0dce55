// Device loop of the reference's joint_optimize (utils.py:746-772): for every face a 3-D
// offset that pulls the decoded surface point grid onto its boundary edge points, fitted by `iters` AdamW steps on the
// one-directional Chamfer loss  L = mean_f sum_{e in edges_f} min_{s in surf_f} |e - (s + off_f)|^2  (chamferdist's
// ChamferDistance(reverse=True) -- the only third-party CUDA kernel of the reference's inference pipeline).
//
// The faces are independent given the 1/F of the mean, so ONE launch runs the whole optimisation: a 256-thread block per
// face keeps its P <= 4096 surface points in LDS (SoA), every thread brute-forces the nearest surface point of its edge
// points, the block reduces the 3-component gradient, thread 0 does the AdamW update (torch.optim.AdamW arithmetic in
// fp32, bias corrections in double) and broadcasts the new offset.  No atomics, fixed reduction order: deterministic.
// Latency-bound by design (200 dependent iterations); the reference spends 200 x F kernel launches + autograd on it.
#include "bg_common.h"

namespace bg {

constexpr int CH_MAXP = 4096;

__global__ __launch_bounds__(256) void chamfer_offset_kernel(const float* __restrict__ surf, const float* __restrict__ edge_pts,
                                                             const int* __restrict__ edge_off, int F, int P, int iters,
                                                             double lr, double beta1, double beta2, float decay_mul,
                                                             float eps, float* __restrict__ offsets_out,
                                                             float* __restrict__ surf_out, float* __restrict__ loss_out) {
    __shared__ float sx[CH_MAXP], sy[CH_MAXP], sz[CH_MAXP];
    __shared__ float red[4][256];
    __shared__ float cur[3];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float* sp = surf + (size_t)f * P * 3;
    for (int i = tid; i < P; i += 256) { sx[i] = sp[3 * i]; sy[i] = sp[3 * i + 1]; sz[i] = sp[3 * i + 2]; }
    const int q0 = edge_off[f], q1 = edge_off[f + 1];
    float off[3] = {0.f, 0.f, 0.f}, m[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f}, used[3] = {0.f, 0.f, 0.f};
    double p1 = 1.0, p2 = 1.0;                                     // beta^it
    const float c1 = (float)(1.0 - beta1), c2 = (float)(1.0 - beta2), b2f = (float)beta2;
    float last_loss = 0.f;
    __syncthreads();
    for (int it = 1; it <= iters; ++it) {
        used[0] = off[0]; used[1] = off[1]; used[2] = off[2];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, ls = 0.f;
        for (int q = q0 + tid; q < q1; q += 256) {
            const float ex = edge_pts[3 * (size_t)q], ey = edge_pts[3 * (size_t)q + 1], ez = edge_pts[3 * (size_t)q + 2];
            float best = 3.4e38f, bx = 0.f, by = 0.f, bz = 0.f;
            for (int s = 0; s < P; ++s) {
                const float dx = ex - (sx[s] + off[0]), dy = ey - (sy[s] + off[1]), dz = ez - (sz[s] + off[2]);
                const float d = (dx * dx + dy * dy) + dz * dz;
                if (d < best) { best = d; bx = dx; by = dy; bz = dz; }   // first minimum wins, like np.argmin
            }
            g0 += -2.0f * bx; g1 += -2.0f * by; g2 += -2.0f * bz;
            ls += best;
        }
        red[0][tid] = g0; red[1][tid] = g1; red[2][tid] = g2; red[3][tid] = ls;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o];
                red[2][tid] += red[2][tid + o]; red[3][tid] += red[3][tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) {
            p1 *= beta1; p2 *= beta2;
            const float step = (float)(lr / (1.0 - p1));
            const float sq2 = (float)sqrt(1.0 - p2);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float g = red[k][0] / (float)F;
                float o = off[k] * decay_mul;                     // decoupled weight decay
                m[k] = m[k] + (g - m[k]) * c1;
                v[k] = v[k] * b2f + g * g * c2;
                const float denom = sqrtf(v[k]) / sq2 + eps;
                o = o - step * m[k] / denom;
                cur[k] = o;
            }
        }
        last_loss = red[3][0];
        __syncthreads();
        off[0] = cur[0]; off[1] = cur[1]; off[2] = cur[2];
        __syncthreads();                                           // red / cur are rewritten next iteration
    }
    // like the reference, the returned surface is the one of the LAST evaluated iteration (utils.py:751,770)
    if (tid < 3) offsets_out[3 * f + tid] = used[tid];
    if (tid == 0 && loss_out) loss_out[f] = last_loss;
    if (surf_out) {
        float* so = surf_out + (size_t)f * P * 3;
        for (int i = tid; i < P; i += 256) { so[3 * i] = sx[i] + used[0]; so[3 * i + 1] = sy[i] + used[1]; so[3 * i + 2] = sz[i] + used[2]; }
    }
}

}  // namespace bg

extern "C" int bg_chamfer_offset_fit(const float* surf, const float* edge_pts, const int* edge_off, int F, int P, int iters,
                                     double lr, double beta1, double beta2, double weight_decay, double eps,
                                     float* offsets_out, float* surf_out, float* loss_out, bg_stream_t stream) {
    BG_REQUIRE(surf && edge_pts && edge_off && offsets_out, BG_E_ARG, "bg_chamfer_offset_fit: null pointer");
    BG_REQUIRE(F > 0 && P > 0 && P <= bg::CH_MAXP && iters >= 0, BG_E_SHAPE, "bg_chamfer_offset_fit: need 0 < P <= 4096 (F=%d P=%d)", F, P);
    const float decay_mul = (float)(1.0 - lr * weight_decay);
    hipLaunchKernelGGL(bg::chamfer_offset_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, surf, edge_pts, edge_off, F, P,
                       iters, lr, beta1, beta2, decay_mul, (float)eps, offsets_out, surf_out, loss_out);
    return bg::launch_status("bg_chamfer_offset_fit");
}
