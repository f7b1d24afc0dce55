// Pieces shared by the 16-bit MFMA GEMM kernels (gemm_16bit.hip: 128 x 128 persistent + generic tiles; gemm_p256.hip: the
// 256 x 256 persistent kernel with the 8-phase K loop).
#pragma once
#include "bg_common.h"
#include <type_traits>

namespace bg {

constexpr int SMALL_LAUNCH_TILES = 160;   // launches of fewer 128 x 128 tiles run on 64 x 64 tiles (gemm_16bit.hip launch16)
constexpr int G_BK = 64;            // 16-bit elements per K-step = 128 bytes per tile row

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

template <bool F16> struct Elem;
template <> struct Elem<false> {
    using T = __bf16; using V8 = bf16x8; using V4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ V4 pack4(float a, float b, float c, float d) { return to_bf16x4(a, b, c, d); }
};
template <> struct Elem<true> {
    using T = _Float16; using V8 = f16x8; using V4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ V4 pack4(float a, float b, float c, float d) {
        // the fp32 values are made opaque first: where they come straight out of an fma, hipcc would otherwise fuse fma + conversion
        // into v_fma_mixlo_f16 (ONE rounding) in some kernels and not in others, and the GEMM kernels would differ by an ulp in ~10 ppm
        // of the fp16 values (seen between the 256 x 256 and the 128 x 128 kernel once the build flags changed in round 6)
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        V4 r; r[0] = (_Float16)a; r[1] = (_Float16)b; r[2] = (_Float16)c; r[3] = (_Float16)d; return r;
    }
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// epilogue selection of the persistent kernels (one instantiation each, so that none carries the registers of another):
//   P_PLAIN16  16-bit output, bias (+ReLU)                              -- QKV / FFN1 without the LayerNorm fold, VAE convs
//   P_FOLD16   same with the LayerNorm fold (stats_in / colsum)         -- QKV / FFN1 of the denoisers
//   P_GENERAL  fp32 or 16-bit output with fp32 addends (add / add2)     -- fp32 residual stream, embeds, VAE residuals
//   P_SPLIT    split (hi, lo) output, addend = split residual or fp32 broadcast rows, optional row statistics
//                                                                       -- out-proj / FFN2 / token embeds of the denoisers
enum { P_PLAIN16 = 0, P_FOLD16 = 1, P_GENERAL = 2, P_SPLIT = 3 };
constexpr int FOLD_PARTS = 12;      // the persistent kernels' LayerNorm fold is compiled for K = 768 (LN width of the denoisers)

// 256 x 256 persistent kernel (gemm_p256.hip).  p256_eligible: shape / argument checks only -- the caller decides whether the
// tile count makes it the faster choice.
bool p256_eligible(const GemmArgs& g);
template <bool F16> int launch_p256(const GemmArgs& g, hipStream_t s);

// software-pipelined split-residual kernel (gemm_split.hip): out-proj / FFN2 of the encoder layers
bool split_pipe_eligible(const GemmArgs& g);
template <bool F16> int launch_split_pipe(const GemmArgs& g, hipStream_t s);

}  // namespace bg
