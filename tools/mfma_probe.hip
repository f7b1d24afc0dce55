// Stand-alone layout probe (diagnostic, not part of the library):
//   1. v_mfma_f32_32x32x16_bf16 operand / result lane maps as assumed by gemm_bf16.hip and attn.hip
//   2. v_mfma_f32_32x32x2_f32 maps as assumed by gemm_f32.hip
//   3. global_load_lds_dwordx4: destination = wave-uniform base + lane * 16
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe ; prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe_bf16(const __bf16* A /*[32][16]*/, const __bf16* Bt /*[32 n][16 k]*/, float* D /*[32][32]*/) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[(l & 31) * 16 + (l >> 5) * 8 + e];
        b[e] = Bt[(l & 31) * 16 + (l >> 5) * 8 + e];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

__global__ void probe_f32(const float* A /*[32][2]*/, const float* Bt /*[32 n][2 k]*/, float* D) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], Bt[(l & 31) * 2 + (l >> 5)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

__global__ void probe_glds(const unsigned* src /*[2][256] dwords*/, unsigned* out /*[512]*/) {
    __shared__ __attribute__((aligned(16))) unsigned lds[512];
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wave w copies its 1 KiB piece; source permuted within the piece: lane l reads chunk (l ^ 5)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w * 256 + (l ^ 5) * 4),
                                     (__attribute__((address_space(3))) void*)(lds + w * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}

int main() {
    int fails = 0;
    {   // bf16
        std::vector<__bf16> A(32 * 16), Bt(32 * 16);
        std::vector<float> Af(32 * 16), Bf(32 * 16);
        for (int i = 0; i < 32 * 16; ++i) {
            Af[i] = (float)((i * 7 + 3) % 13 - 6);
            Bf[i] = (float)((i * 5 + 1) % 11 - 5);
            A[i] = (__bf16)Af[i];
            Bt[i] = (__bf16)Bf[i];
        }
        __bf16 *dA, *dB; float* dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, Bt.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        std::vector<float> D(32 * 32);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float ref = 0;
                for (int k = 0; k < 16; ++k) ref += Af[i * 16 + k] * Bf[j * 16 + k];
                if (ref != D[i * 32 + j]) ++bad;
            }
        printf("%s mfma_f32_32x32x16_bf16 layout (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        fails += bad != 0;
    }
    {   // f32
        std::vector<float> A(64), Bt(64), D(1024);
        for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 7 + 3) % 13 - 6); Bt[i] = (float)((i * 5 + 1) % 11 - 5); }
        float *dA, *dB, *dD;
        hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 4096);
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dB, Bt.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float ref = A[i * 2] * Bt[j * 2] + A[i * 2 + 1] * Bt[j * 2 + 1];
                if (ref != D[i * 32 + j]) ++bad;
            }
        printf("%s mfma_f32_32x32x2f32 layout (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        fails += bad != 0;
    }
    {   // LDS-DMA
        std::vector<unsigned> S(512), O(512);
        for (int i = 0; i < 512; ++i) S[i] = 1000 + i;
        unsigned *dS, *dO;
        hipMalloc(&dS, 2048); hipMalloc(&dO, 2048);
        hipMemcpy(dS, S.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_glds, dim3(1), dim3(128), 0, 0, dS, dO);
        hipMemcpy(O.data(), dO, 2048, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int w = 0; w < 2; ++w)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e)
                    if (O[w * 256 + l * 4 + e] != S[w * 256 + (l ^ 5) * 4 + e]) ++bad;
        printf("%s global_load_lds_dwordx4 lane-linear destination (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) for (int i = 0; i < 16; ++i) printf("  lds[%d]=%u\n", i * 4, O[i * 4]);
        fails += bad != 0;
    }
    hipError_t e = hipDeviceSynchronize();
    printf("device status: %s\n", hipGetErrorString(e));
    return fails;
}
