// Checks the DPP encodings common.h uses for x[l ^ 8], x[l ^ 4], x[l ^ 2], x[l ^ 1] (row_xor<O>): with x = lane id every
// lane must read lane ^ O.  Build: hipcc --offload-arch=gfx950 -O3 -I unimedvl_amd/csrc tools/dpp_probe.hip -o tools/bin/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.h"
__global__ void k(int* out) {
    const float x = (float)threadIdx.x;
    out[threadIdx.x * 4 + 0] = (int)row_xor<8>(x);
    out[threadIdx.x * 4 + 1] = (int)row_xor<4>(x);
    out[threadIdx.x * 4 + 2] = (int)row_xor<2>(x);
    out[threadIdx.x * 4 + 3] = (int)row_xor<1>(x);
}
__global__ void s(float* out) {   // wave_sum / wave_max of distinct values: exact in fp32 (small integers)
    const float x = (float)(threadIdx.x * 3 + 1);
    out[threadIdx.x] = wave_sum(x);
    out[64 + threadIdx.x] = wave_max(x);
}
int main() {
    int* d; int h[256];
    float* f; float g[128];
    hipMalloc(&d, sizeof(h)); hipMalloc(&f, sizeof(g));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(s, dim3(1), dim3(64), 0, 0, f);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(g, f, sizeof(g), hipMemcpyDeviceToHost);
    int ok = 1;
    const int O[4] = {8, 4, 2, 1};
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) ok &= (h[l * 4 + i] == (l ^ O[i]));
    float tot = 0; for (int l = 0; l < 64; ++l) tot += (float)(l * 3 + 1);
    for (int l = 0; l < 64; ++l) ok &= (g[l] == tot) && (g[64 + l] == (float)(63 * 3 + 1));
    printf("row_xor<8,4,2,1> read lane ^ O, wave_sum / wave_max agree on every lane: %s\n", ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
