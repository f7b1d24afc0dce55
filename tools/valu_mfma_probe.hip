// Stand-alone probe (diagnostic, not part of the library): what does a VALU instruction COST next to MFMAs on gfx950?
// The long-sequence attention tile (csrc/attn.hip) is 16 x v_mfma_f32_32x32x16 (512 matrix-pipe cycles per wave) next to ~33 v_exp_f32 and
// ~130 other VALU instructions; DESIGN.md priced the exponentials at 16 cycles each ("quarter rate") and derived a 0.49 ceiling from
// that.  This measures it: W waves per SIMD (1, 2, 4), each running ITER iterations of
//     16 MFMAs (two independent accumulator chains, like S^T / O^T)  +  NV filler instructions of one kind, interleaved evenly,
// timed with s_memtime on one workgroup per CU (all CUs busy, so the clock is the loaded one).  Output: cycles per iteration per wave
// and per SIMD for filler = none / v_fma_f32 / v_exp_f32 / v_max3_f32 / v_cvt_pk_bf16_f32, at NV = 32, 64, 128, 160.
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/valu_mfma_probe.hip -o gpurun_out/valu_mfma_probe && gpurun_out/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { F_NONE = 0, F_FMA = 1, F_EXP = 2, F_MAX3 = 3, F_CVT = 4, F_MIX = 5 };      // F_MIX: 2 v_exp_f32 + (PER - 2) v_fma_f32 per MFMA = the attention tile's mix at PER = 10

template <int KIND, int PER_MFMA>      // PER_MFMA fillers behind each of the 16 MFMAs
__global__ __launch_bounds__(1024) void probe(long long* out, int iters, float seed) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = seed * (e + 1) + threadIdx.x * 1e-3f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < PER_MFMA; ++f) {
                float& x = v[(m * PER_MFMA + f) & 7];              // eight independent chains: issue-bound, not latency-bound
                if (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed));
                if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == F_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed));
                if (KIND == F_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(seed));
                if (KIND == F_MIX && f < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == F_MIX && f >= 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(seed));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    for (int r = 0; r < 16; ++r) sink += acc0[r] + acc1[r];
    for (int e = 0; e < 8; ++e) sink += v[e];
    if (sink == 12345.678f) out[0] = 0;                            // keep everything alive
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[1 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int PER>
static void run(const char* name, long long* d_out, int waves_per_simd) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<KIND, PER>), dim3(256), dim3(threads), 0, 0, d_out, 10, 0.5f);          // warm
    hipLaunchKernelGGL((probe<KIND, PER>), dim3(256), dim3(threads), 0, 0, d_out, iters, 0.5f);
    hipDeviceSynchronize();
    std::vector<long long> h(20);
    hipMemcpy(h.data(), d_out, 20 * sizeof(long long), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int w = 0; w < threads / 64; ++w) worst = worst > (double)h[1 + w] ? worst : (double)h[1 + w];
    const double per_iter = worst / iters;          // cycles for ALL waves of the SIMD to finish one iteration each (they run concurrently)
    printf("%-22s x%2d per MFMA (%3d fillers)  %d wave(s)/SIMD: %7.1f cycles per iteration (512 = matrix pipe of ONE wave; SIMD needs %d) -> matrix pipe busy %.2f\n",
           name, PER, 16 * PER, waves_per_simd, per_iter, 512 * waves_per_simd, 512.0 * waves_per_simd / per_iter);
}

int main() {
    long long* d_out;
    hipMalloc(&d_out, 64 * sizeof(long long));
    for (int w : {1, 2, 4}) {
        run<F_NONE, 0>("no filler", d_out, w);
        run<F_FMA, 2>("v_fma_f32", d_out, w);
        run<F_FMA, 4>("v_fma_f32", d_out, w);
        run<F_FMA, 8>("v_fma_f32", d_out, w);
        run<F_FMA, 10>("v_fma_f32", d_out, w);
        run<F_EXP, 2>("v_exp_f32", d_out, w);
        run<F_EXP, 4>("v_exp_f32", d_out, w);
        run<F_EXP, 8>("v_exp_f32", d_out, w);
        run<F_MAX3, 4>("v_max3_f32", d_out, w);
        run<F_MAX3, 8>("v_max3_f32", d_out, w);
        run<F_CVT, 4>("v_cvt_pk_bf16_f32", d_out, w);
        run<F_CVT, 8>("v_cvt_pk_bf16_f32", d_out, w);
        run<F_MIX, 10>("2 exp + 8 fma (tile mix)", d_out, w);
        run<F_MIX, 6>("2 exp + 4 fma", d_out, w);
    }
    return 0;
}
