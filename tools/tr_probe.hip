// Semantics probe for ds_read_b64_tr_b16 (the LDS transpose read the long-sequence attention kernel relies on).
// Assumed (csrc/attn.hip): within each 16-lane group, lane i receives element (i & 3) of the 8 bytes addressed by lane
// j*4 + (i >> 2), for j = 0..3 -- i.e. column i of the 4 x 16 matrix whose row r, columns 4c..4c+3 lane 4r + c addresses.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short v4s;
__global__ void probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    for (int e = 0; e < 4; ++e) lds[threadIdx.x * 4 + e] = (short)(threadIdx.x * 4 + e);   // element id = lane*4 + e
    __syncthreads();
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
    short* d; short h[256];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l >> 4, i = l & 15;
            const int want = (g * 16 + j * 4 + (i >> 2)) * 4 + (i & 3);
            bad += h[l * 4 + j] != want;
        }
    printf("tr_probe: %s (%d mismatches)\n", bad ? "ASSUMPTION WRONG" : "assumed semantics confirmed", bad);
    if (bad) for (int l = 0; l < 64; ++l) printf("lane %2d: src lane/elem %d.%d %d.%d %d.%d %d.%d\n", l, h[l*4]/4, h[l*4]%4, h[l*4+1]/4, h[l*4+1]%4, h[l*4+2]/4, h[l*4+2]%4, h[l*4+3]/4, h[l*4+3]%4);
    return 0;
}
