// Stand-alone probe (diagnostic, not part of the library).  Round 6 found that bg_im2col / norm_act (GroupNorm + activation + cast: a pure
// elementwise kernel) returned WRONG values -- computed with another group's (mean, rstd), always in lanes 48-63 of a wave -- whenever a
// 16-bit GEMM of this library ran on a second stream; torch.mm beside it did not disturb it, the library's fp32 GEMM did not, and the
// same elementwise kernel built with -fno-slp-vectorize was immune.  This program isolates the ingredients:
//   victim    : out = (x - mean[g]) * rstd[g] * gamma + beta on float4 per thread, written in plain C++ (hipcc -O3 turns it into
//               v_pk_add_f32 / v_pk_mul_f32 with op_sel broadcast modifiers); run alone for the reference, then beside a disturber
//   disturbers: (1) bf16 MFMA loop, (2) LDS-DMA loop (global_load_lds_dwordx4 from an L2-resident buffer, nothing else),
//               (3) both in one wave (the shape of the library's GEMM K loops), (4) plain v_fma_f32 loop (control)
// and reports the number of mismatching output elements per lane quarter.  Build the victim twice (-DNOSLP adds
// __attribute__((optnone))-free scalar code via volatile-free manual scalarisation is NOT needed: compile the whole file with and without
// -fno-slp-vectorize).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/pk_f32_mfma_hazard_probe.hip -o gpurun_out/pk_probe && gpurun_out/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct VArgs { const float* x; float* out; const float* stats; const float* gamma; const float* beta; unsigned total4, n4s, c4n, cpg4, G; };

__global__ __launch_bounds__(256) void victim(VArgs a) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x);
    const unsigned i0 = blockIdx.x * 1024u + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = i0 + k * 256u;
        if (i < a.total4) v[k] = x4[i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = i0 + k * 256u;
        if (i >= a.total4) continue;
        float4 w = v[k];
        const unsigned c4 = i % a.c4n, smp = i / a.n4s;
        const float mean = a.stats[((size_t)smp * a.G + c4 / a.cpg4) * 2], rstd = a.stats[((size_t)smp * a.G + c4 / a.cpg4) * 2 + 1];
        const float4 ga = *reinterpret_cast<const float4*>(a.gamma + c4 * 4);
        const float4 be = *reinterpret_cast<const float4*>(a.beta + c4 * 4);
        w.x = (w.x - mean) * rstd * ga.x + be.x;
        w.y = (w.y - mean) * rstd * ga.y + be.y;
        w.z = (w.z - mean) * rstd * ga.z + be.z;
        w.w = (w.w - mean) * rstd * ga.w + be.w;
        reinterpret_cast<float4*>(a.out)[i] = w;
    }
}

// the same arithmetic as ONE instruction pair in inline asm on register operands (no memory traffic in the victim): which instruction form
// is the vulnerable one?  OPSEL: v_pk_add_f32 / v_pk_mul_f32 with the op_sel broadcast modifiers; otherwise plain packed ops on
// pre-broadcast operands.  Compared in-kernel with scalar v_sub_f32 / v_mul_f32; mismatches counted per 16-lane quarter.
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <bool OPSEL>
__global__ __launch_bounds__(256) void victim_asm(unsigned* bad_per_quarter, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    float seed = 0.37f * (float)(blockIdx.x * 256 + threadIdx.x) + 1.0f;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1.000173f + 0.013f;
        f32x2 x = {seed, seed * 0.5f - 3.0f};
        f32x2 p = {seed * 0.25f + (float)(lane >> 2), 1.0f + 0.001f * (float)lane};
        f32x2 d, y;
        if (OPSEL) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(p));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(y) : "v"(d), "v"(p));
        } else {
            f32x2 pm = {p.x, p.x}, pr = {p.y, p.y};
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(pm));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(d), "v"(pr));
        }
        float r0, r1;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x.x), "v"(p.x));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x.y), "v"(p.x));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r0) : "v"(p.y));
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r1) : "v"(p.y));
        bad += (__float_as_uint(y.x) != __float_as_uint(r0)) + (__float_as_uint(y.y) != __float_as_uint(r1));
    }
    if (bad) atomicAdd(&bad_per_quarter[lane >> 4], bad);
}

template <int KIND>      // 1 MFMA, 2 LDS-DMA, 3 both, 4 v_fma
__global__ __launch_bounds__(256) void disturber(const unsigned char* src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.5f + e); b[e] = (__bf16)(0.25f - e); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    float v = threadIdx.x * 1e-3f;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* my = src + ((size_t)(blockIdx.x & 255) * 4 + wave) * 65536 + lane * 16;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2 || KIND == 3) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(my + ((it & 7) * 8 + p) * 1024),
                                                 (__attribute__((address_space(3))) void*)(lds + wave * 8192 + p * 1024), 16, 0, 0);
        }
        if (KIND == 1 || KIND == 3) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
        }
        if (KIND == 4) {
#pragma unroll
            for (int m = 0; m < 64; ++m) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(0.999f));
        }
        if (KIND == 2 || KIND == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = v + (float)lds[threadIdx.x];
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 1234.5f) sink[0] = s;
}

int main() {
    const unsigned S = 64, P = 64, C = 512, G = 32;
    const unsigned total4 = S * P * C / 4;
    std::vector<float> hx(S * P * C), hs(S * G * 2), hg(C, 1.0f), hb(C, 0.0f);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 8192.0f - 4.0f;
    for (size_t i = 0; i < hs.size(); i += 2) { hs[i] = 0.01f * (float)(i % 97) - 0.4f; hs[i + 1] = 1.0f + 0.003f * (float)(i % 53); }
    float *dx, *dout, *dref, *ds, *dg, *db, *dsink;
    unsigned char* dsrc;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dout, hx.size() * 4); hipMalloc(&dref, hx.size() * 4);
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&dg, C * 4); hipMalloc(&db, C * 4); hipMalloc(&dsink, 4);
    hipMalloc(&dsrc, (size_t)256 * 4 * 65536 + 65536);
    hipMemset(dsrc, 1, (size_t)256 * 4 * 65536 + 65536);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg, hg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice);
    VArgs a{dx, dref, ds, dg, db, total4, P * C / 4, C / 4, (C / G) / 4, G};
    hipStream_t s0, s1;
    hipStreamCreate(&s0); hipStreamCreate(&s1);
    hipLaunchKernelGGL(victim, dim3((total4 + 1023) / 1024), dim3(256), 0, s0, a);
    hipDeviceSynchronize();
    std::vector<float> ref(hx.size()), got(hx.size());
    hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost);
    a.out = dout;
    const char* names[5] = {"nothing", "bf16 MFMA loop", "LDS-DMA loop (global_load_lds)", "MFMA + LDS-DMA", "v_fma_f32 loop"};
    for (int kind = 0; kind <= 4; ++kind) {
        unsigned bad[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 10; ++rep) {
            hipMemset(dout, 0, hx.size() * 4);
            hipDeviceSynchronize();
            if (kind == 1) hipLaunchKernelGGL((disturber<1>), dim3(1024), dim3(256), 0, s1, dsrc, dsink, 3000);
            if (kind == 2) hipLaunchKernelGGL((disturber<2>), dim3(1024), dim3(256), 0, s1, dsrc, dsink, 3000);
            if (kind == 3) hipLaunchKernelGGL((disturber<3>), dim3(1024), dim3(256), 0, s1, dsrc, dsink, 3000);
            if (kind == 4) hipLaunchKernelGGL((disturber<4>), dim3(1024), dim3(256), 0, s1, dsrc, dsink, 3000);
            for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(victim, dim3((total4 + 1023) / 1024), dim3(256), 0, s0, a);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < got.size(); ++i)
                if (got[i] != ref[i]) bad[((i / 4) % 64) / 16]++;       // lane of the float4's thread inside its wave
        }
        printf("victim beside %-34s: mismatching elements by lane quarter (0-15, 16-31, 32-47, 48-63): %u %u %u %u\n", names[kind], bad[0], bad[1], bad[2], bad[3]);
    }
    // register-only victims beside the MFMA + LDS-DMA disturber
    unsigned* d_bad;
    hipMalloc(&d_bad, 16);
    for (int form = 0; form < 2; ++form) {
        unsigned tot[4] = {0, 0, 0, 0};
        for (int rep = 0; rep < 5; ++rep) {
            hipMemset(d_bad, 0, 16);
            hipDeviceSynchronize();
            hipLaunchKernelGGL((disturber<3>), dim3(1024), dim3(256), 0, s1, dsrc, dsink, 3000);
            if (form == 0) hipLaunchKernelGGL((victim_asm<true>), dim3(2048), dim3(256), 0, s0, d_bad, 4000);
            else hipLaunchKernelGGL((victim_asm<false>), dim3(2048), dim3(256), 0, s0, d_bad, 4000);
            hipDeviceSynchronize();
            unsigned h[4];
            hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost);
            for (int q = 0; q < 4; ++q) tot[q] += h[q];
        }
        printf("register-only victim, %-28s beside MFMA + LDS-DMA: mismatches by lane quarter: %u %u %u %u\n",
               form == 0 ? "op_sel packed f32 (inline asm)" : "plain packed f32 (inline asm)", tot[0], tot[1], tot[2], tot[3]);
    }
    return 0;
}
