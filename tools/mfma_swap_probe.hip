// Is v_mfma_f32_32x32x16_{bf16,f16} symmetric in its operands, bit for bit?  D1 = mfma(a, b, c) gives D1[i][j] in lane (j, half h),
// D2 = mfma(b, a, c^T) gives D2[j][i]; the persistent 256x256 GEMM computes the transposed product (so that a lane owns one output
// ROW and packs 4 consecutive columns without any cross-lane exchange) and must stay bit-identical to the other GEMM kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_swap_probe.hip -o gpurun_out/mfma_swap_probe && gpurun_out/mfma_swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool F16>
__global__ void probe(const unsigned short* A, const unsigned short* B, float* D1, float* D2, int ksteps) {
    // A: [32][16 * ksteps] (row i), B: [32][16 * ksteps] (row j), both K-contiguous like the GEMM operands
    const int lane = threadIdx.x, r32 = lane & 31, h = lane >> 5;
    f32x16 c1, c2;
    for (int r = 0; r < 16; ++r) { c1[r] = 0.25f * (r + 1) + r32; c2[r] = 0.f; }
    // c2 must be the transpose of c1: c1 lane(j=r32,h) reg r = C[i = (r&3)+8(r>>2)+4h][j]; choose C[i][j] = 0.25 * (i + 1) + 3 j
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        c1[r] = 0.25f * (i + 1) + 3.f * r32;      // C[i][j = r32]
        c2[r] = 0.25f * (r32 + 1) + 3.f * i;      // C^T: lane owns i' = r32 (as the "j" of the swapped product), reg -> j' = i
    }
    for (int ks = 0; ks < ksteps; ++ks) {
        union { bf16x8 b; f16x8 f; uint4 u; } a, b;
        a.u = *reinterpret_cast<const uint4*>(A + r32 * 16 * ksteps + ks * 16 + h * 8);
        b.u = *reinterpret_cast<const uint4*>(B + r32 * 16 * ksteps + ks * 16 + h * 8);
        if (F16) {
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.f, b.f, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.f, a.f, c2, 0, 0, 0);
        } else {
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b, a.b, c2, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        D1[i * 32 + r32] = c1[r];                 // D1[i][j]
        D2[r32 * 32 + i] = c2[r];                 // swapped product: lane owns row r32 of D (as i), reg -> column
    }
}

int main() {
    const int ks = 48;                            // K = 768
    std::vector<unsigned short> A(32 * 16 * ks), B(32 * 16 * ks);
    int bad_total = 0;
    for (int f16 = 0; f16 < 2; ++f16) {
        srand(7 + f16);
        for (auto& v : A) { float x = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &x, 4); v = f16 ? (unsigned short)(0x3000 + (rand() & 0x0fff) + ((rand() & 1) << 15)) : (unsigned short)(u >> 16); }
        for (auto& v : B) { float x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &x, 4); v = f16 ? (unsigned short)(0x2800 + (rand() & 0x0fff) + ((rand() & 1) << 15)) : (unsigned short)(u >> 16); }
        unsigned short *dA, *dB; float *d1, *d2;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&d1, 4096); hipMalloc(&d2, 4096);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        if (f16) hipLaunchKernelGGL(probe<true>, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2, ks);
        else hipLaunchKernelGGL(probe<false>, dim3(1), dim3(64), 0, 0, dA, dB, d1, d2, ks);
        float h1[1024], h2[1024];
        hipMemcpy(h1, d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 4096, hipMemcpyDeviceToHost);
        int bad = 0; double s = 0;
        for (int i = 0; i < 1024; ++i) { bad += memcmp(&h1[i], &h2[i], 4) != 0; s += h1[i]; }
        printf("%s: mfma(a,b)[i][j] vs mfma(b,a)[j][i]: %d of 1024 elements differ bitwise (checksum %.6f, D[3][5] = %.6f / %.6f)\n",
               f16 ? "f16 " : "bf16", bad, s, h1[3 * 32 + 5], h2[3 * 32 + 5]);
        bad_total += bad;
    }
    return bad_total != 0;
}
