// Which workgroups of a 512-block, 80-KiB-LDS, 256-thread launch share a CU?  (Speed-only knowledge for the persistent
// GEMM's tile walk: co-resident workgroups that work on tiles of the same A row panel can hit each other's lines in the
// CU's vector L1.)  Build + run:  hipcc --offload-arch=gfx950 -O2 tools/cu_census.hip -o /tmp/cu_census && /tmp/cu_census
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void census(unsigned* out, unsigned long long hold) {
    __shared__ unsigned char pad[80 * 1024];
    pad[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < hold) __builtin_amdgcn_s_sleep(8);   // keep every block resident together
    if (pad[(threadIdx.x * 7) & 255] == 255 && hold == 0) out[0] = 0;
}

int main() {
    const int G = 512;
    unsigned* d;
    hipMalloc(&d, G * 2 * sizeof(unsigned));
    std::vector<unsigned> h(G * 2);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(census, dim3(G), dim3(256), 0, 0, d, 4000000ull);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, G * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
    }
    std::map<unsigned, std::vector<int>> by_cu;
    int xcd_ok = 0;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
        xcd_ok += (int)(xcc == (unsigned)(b & 7));
    }
    printf("blocks on XCD b%%8: %d / %d; distinct CUs seen: %zu\n", xcd_ok, G, by_cu.size());
    std::map<int, int> delta_hist;
    for (auto& kv : by_cu) {
        if (kv.second.size() == 2) delta_hist[kv.second[1] - kv.second[0]]++;
        else delta_hist[-(int)kv.second.size()]++;
    }
    for (auto& kv : delta_hist) printf("  co-resident pair block-id delta %d : %d CUs\n", kv.first, kv.second);
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 12) break;
        printf("  xcc %u se %u sh %u cu %u :", kv.first >> 16, (kv.first >> 8) & 0xff, (kv.first >> 4) & 0xf, kv.first & 0xf);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    return 0;
}
