// Counter-based ancestral noise for the sharded DDPM loops (sample.py:144-153, 224-236 draw it with randn on the device).
//
// Philox4x32-10 keyed by (seed), counter = (element block, GLOBAL sample index, draw id, domain tag): the value of element
// e of sample b of draw d depends on nothing else -- not on the rank that owns the sample, not on the batch it is in --
// so an N-GPU run reproduces the 1-GPU run sample for sample without any rank drawing (or copying) noise it does not
// use.  4 x 32 random bits -> 2 Box-Muller pairs -> 4 N(0,1) values.  HBM-bound (4 B written per element).
#include "bg_common.h"

namespace bg {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// u32 -> uniform in (0, 1): the top 23 bits, centred.  k + 0.5 with k < 2^23 is exactly representable in fp32 (24
// significant bits), so the 2^23 grid points are equally spaced, none is 0 and none is 1.
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

template <bool RAW>
__global__ __launch_bounds__(256) void philox_randn_kernel(float* __restrict__ out, long long n_samples, int per,
                                                           uint32_t seed_lo, uint32_t seed_hi, uint32_t draw,
                                                           long long sample0) {
    const int blocks_per = (per + 3) / 4;
    const long long total = n_samples * blocks_per;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / blocks_per;
        const int blk = (int)(i - b * blocks_per);
        const unsigned long long gs = (unsigned long long)(sample0 + b);
        uint32_t c[4] = {(uint32_t)blk, (uint32_t)gs, draw, 0xB9E50000u | (uint32_t)((gs >> 32) & 0xFFFFu)};
        philox4x32_10(c, seed_lo, seed_hi);
        float v[4];
        if (RAW) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(c[j]);
        } else {
            const float r0 = sqrtf(-2.0f * logf(u01(c[0]))), r1 = sqrtf(-2.0f * logf(u01(c[2])));
            const float a0 = 6.28318530717958647692f * u01(c[1]), a1 = 6.28318530717958647692f * u01(c[3]);
            v[0] = r0 * cosf(a0); v[1] = r0 * sinf(a0); v[2] = r1 * cosf(a1); v[3] = r1 * sinf(a1);
        }
        float* o = out + b * per + blk * 4;
        const int left = per - blk * 4;
        if (left >= 4 && (per & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < left) o[j] = v[j];
    }
}

}  // namespace bg

extern "C" int bg_philox_randn(float* out, long long n_samples, int per_sample, unsigned long long seed,
                               unsigned draw_id, long long first_sample, int raw_bits, bg_stream_t stream) {
    using namespace bg;
    BG_REQUIRE(out != nullptr || n_samples == 0, BG_E_ARG, "bg_philox_randn: null output");
    BG_REQUIRE(n_samples >= 0 && per_sample > 0 && first_sample >= 0, BG_E_SHAPE, "bg_philox_randn: bad shape");
    BG_REQUIRE(((uintptr_t)out & 15) == 0, BG_E_ALIGN, "bg_philox_randn: out must be 16-byte aligned");
    if (n_samples == 0) return 0;
    const long long total = n_samples * ((per_sample + 3) / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    ProfScope prof(PK_MISC, 0.0, 4.0 * (double)n_samples * per_sample, (hipStream_t)stream);
    if (raw_bits)
        hipLaunchKernelGGL(philox_randn_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n_samples, per_sample,
                           (uint32_t)seed, (uint32_t)(seed >> 32), draw_id, first_sample);
    else
        hipLaunchKernelGGL(philox_randn_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n_samples, per_sample,
                           (uint32_t)seed, (uint32_t)(seed >> 32), draw_id, first_sample);
    return launch_status("philox_randn");
}
