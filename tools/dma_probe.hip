// How many bytes per second does ONE CU pull through its L2 -> LDS path (global_load_lds_dwordx4, the staging primitive of every GEMM
// here), as a function of the bytes it keeps in flight?  DESIGN.md section 4 (round 4) finds the 128 x 128 GEMM kernels at 46-50
// GB/s per CU with 64 KiB in flight and the 256 x 256 kernel at 40-45 with ~96 KiB; this probe measures the path alone, with the
// access pattern of a GEMM's operand loads (8 rows x 128 B per wave-instruction, 16-byte XOR-swizzled source, row stride = K):
//   * source "W": one 1.2 MB matrix (768 x 768 16-bit) re-read by every workgroup -- L2-resident after the first pass;
//   * source "A": a 47 MB matrix (30 720 x 768) streamed, every 256-row panel read by the six workgroups of one XCD that a GEMM's six
//     column tiles would be -- one L2 miss per line and XCD, served by the Infinity Cache (where the residual stream lives);
//   * S ring slots of 32 KiB (256 rows x 128 B), S - 1 in flight behind a counted vmcnt, 1 or 2 workgroups per CU.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void dma(unsigned dst, const unsigned char* src, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(dst) : "memory");
}

// One workgroup = 4 waves; a "stage" = 256 rows x 128 B = 32 KiB = 8 DMA instructions per wave.  SLOTS ring slots, SLOTS - 1 stages in
// flight.  steps stages per workgroup; stage k reads rows (base_row + (k % rows_per_wg_stages) * 256 ...) x bytes [kcol*128, +128).
template <int SLOTS>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ src, unsigned row_bytes, int rows_total, int share,
                                             int steps, int use_barrier, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    unsigned voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (wave * 8 + j) * 8 + (lane >> 3);
        voff[j] = (unsigned)row * row_bytes + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
    }
    const int kt = row_bytes / 128;                               // K-steps per row panel
    // panels of 256 rows; a workgroup reads steps / kt consecutive panels.  Groups of `share` workgroups of one XCD (block b runs on
    // XCD b % 8) read the SAME panels at the same time, like the column tiles of a GEMM that share an A panel.
    const int n_panels = rows_total / 256, ppw = steps / kt;
    const int xcd = blockIdx.x & 7, grp = (blockIdx.x >> 3) / share;
    const int first_panel = ((xcd + 8 * grp) * ppw) % n_panels;
    auto issue = [&](int k) {
        const int slot = k % SLOTS;
        const int panel = (first_panel + k / kt) % n_panels;
        const unsigned char* s = src + (size_t)panel * 256 * row_bytes + (size_t)(k % kt) * 128;
#pragma unroll
        for (int j = 0; j < 8; ++j) dma(lds0 + (unsigned)(slot * 32768 + (wave * 8 + j) * 1024), s, voff[j]);
    };
    for (int k = 0; k < SLOTS - 1 && k < steps; ++k) issue(k);
    unsigned acc = 0;
    for (int k = 0; k < steps; ++k) {
        if (k + SLOTS - 1 < steps) {
            wait_vmcnt<(SLOTS - 2) * 8>();                        // stage k landed; SLOTS - 2 younger stages stay in flight
        } else {
            wait_vmcnt<0>();
        }
        if (use_barrier) __builtin_amdgcn_s_barrier();
        acc += lds[(k % SLOTS) * 32768 + threadIdx.x * 4];      // touch the slot (keeps the DMA from being dead code for the reader)
        if (k + SLOTS - 1 < steps) issue(k + SLOTS - 1);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

// The residual read-modify-write of the split epilogue alone, in its access pattern: a workgroup walks 128 x 128 tiles of two 16-bit
// planes [M, 768] (hi, lo); a wave owns a 64 x 64 block = four slabs of 16 rows x 64 columns; a lane moves 16 bytes of two rows per slab
// and plane (8 rows x 128-byte segments per wave instruction, row stride 1536 B).  `linear` = 1: the same bytes per instruction, but as
// one contiguous KiB (what a tile-major plane would give).  All loads of a tile first, then all stores (the most memory-level
// parallelism the pattern allows).
__global__ __launch_bounds__(256, 2) void rmw_probe(unsigned char* __restrict__ hi, unsigned char* __restrict__ lo, int M, int linear) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int k8 = lane & 7, r8 = lane >> 3;
    const int nt_n = 6, tiles = (M / 128) * nt_n;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int m0 = (t / nt_n) * 128 + wm * 64, n0 = (t % nt_n) * 128 + wn * 64;
        u4 vh[4][2], vl[4][2];
        size_t off[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = m0 + c * 16 + it * 8 + r8;
                off[c][it] = linear ? ((size_t)t * 16384 + (size_t)(wave * 8 + c * 2 + it) * 1024 + lane * 16) * 1   // 32 KiB per tile and plane
                                    : ((size_t)row * 768 + n0 + k8 * 8) * 2;
                vh[c][it] = *reinterpret_cast<const u4*>(hi + off[c][it]);
                vl[c][it] = *reinterpret_cast<const u4*>(lo + off[c][it]);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                *reinterpret_cast<u4*>(hi + off[c][it]) = vh[c][it] + 1u;
                *reinterpret_cast<u4*>(lo + off[c][it]) = vl[c][it] + 1u;
            }
    }
}

int main() {
    const int K = 768, M = 30720;
    unsigned char *dW, *dA;
    unsigned* sink;
    CHECK(hipMalloc(&dW, (size_t)768 * K * 2));
    CHECK(hipMalloc(&dA, (size_t)M * K * 2));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(dW, 1, (size_t)768 * K * 2));
    CHECK(hipMemset(dA, 1, (size_t)M * K * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto run = [&](int slots, int wg_per_cu, bool resident, int barrier) {
        const int grid = 256 * wg_per_cu;
        const size_t lds = (size_t)slots * 32768;
        // streaming: 120 panels of 256 rows x 768 cols (12 K-steps each): every workgroup reads whole panels, `steps` stages
        const int steps = resident ? 36 * 8 : 12 * 6;             // resident: 8 passes over W (3 panels x 12); streaming: 6 panels per workgroup
        const unsigned char* src = resident ? dW : dA;
        const int rows_total = resident ? 768 : M, share = 6;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
#define LAUNCH(S) do { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                       hipLaunchKernelGGL(probe<S>, dim3(grid), dim3(256), lds, 0, src, (unsigned)(K * 2), rows_total, share, steps, barrier, sink); } while (0)
            if (slots == 2) LAUNCH(2); else if (slots == 3) LAUNCH(3); else if (slots == 4) LAUNCH(4); else LAUNCH(5);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double bytes = (double)grid * steps * 32768.0;
        printf("%-9s slots %d (%3d KiB in flight per workgroup) x %d workgroup(s)/CU%s: %7.1f us  %6.1f GB/s per CU  %5.2f TB/s chip\n",
               resident ? "L2 (W)" : "stream(A)", slots, (slots - 1) * 32, wg_per_cu, barrier ? " + barrier" : "          ", best * 1e3,
               bytes / 256 / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    };
    for (int resident = 1; resident >= 0; --resident)
        for (int barrier = 0; barrier <= 1; ++barrier)
            for (int wg = 1; wg <= 2; ++wg)
                for (int slots = 2; slots <= (wg == 1 ? 5 : 2); ++slots) run(slots, wg, resident, barrier);
    // two workgroups per CU with 3 slots do not fit 160 KiB (2 x 96); 2 x 64 KiB (2 slots) is the GEMM kernels' configuration
    // ---- the residual read-modify-write in the epilogue's pattern vs as contiguous KiB ----
    for (int rows : {30720, 138752}) {
        unsigned char *ph, *pl;
        const size_t bytes = (size_t)rows * 768 * 2;
        CHECK(hipMalloc(&ph, bytes));
        CHECK(hipMalloc(&pl, bytes));
        CHECK(hipMemset(ph, 0, bytes));
        CHECK(hipMemset(pl, 0, bytes));
        for (int linear = 0; linear <= 1; ++linear)
            for (int grid : {512, 1024, 2048}) {
                float best = 1e30f;
                for (int rep = 0; rep < 6; ++rep) {
                    CHECK(hipEventRecord(e0));
                    hipLaunchKernelGGL(rmw_probe, dim3(grid), dim3(256), 0, 0, ph, pl, rows, linear);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("residual RMW, %6d rows (%5.1f MB per plane), %s, grid %4d: %7.1f us  %5.2f TB/s (read + write)\n", rows, bytes / 1e6,
                       linear ? "contiguous KiB per instruction" : "8 rows x 128 B per instruction", grid, best * 1e3, 4.0 * bytes / (best * 1e-3) / 1e12);
            }
        CHECK(hipFree(ph));
        CHECK(hipFree(pl));
    }
    return 0;
}
