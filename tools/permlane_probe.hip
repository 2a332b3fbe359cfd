// Lane mapping of the gfx950 row-swap instructions (v_permlane16_swap_b32 / v_permlane32_swap_b32) with both operands = x:
// prints, per lane, which source lanes end up in result[0] / result[1].  Build: hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/bin/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned x = threadIdx.x;
    auto r16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    auto r32 = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[threadIdx.x * 4 + 0] = r16[0]; out[threadIdx.x * 4 + 1] = r16[1];
    out[threadIdx.x * 4 + 2] = r32[0]; out[threadIdx.x * 4 + 3] = r32[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: swap16 -> (%2u, %2u)   swap32 -> (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        unsigned a = h[l * 4], b = h[l * 4 + 1], c = h[l * 4 + 2], e = h[l * 4 + 3];
        ok &= ((a == (unsigned)l && b == (unsigned)(l ^ 16)) || (b == (unsigned)l && a == (unsigned)(l ^ 16)));
        ok &= ((c == (unsigned)l && e == (unsigned)(l ^ 32)) || (e == (unsigned)l && c == (unsigned)(l ^ 32)));
    }
    printf("each lane holds {x[l], x[l^16]} resp. {x[l], x[l^32]} in its two results: %s\n", ok ? "yes" : "NO");
    return 0;
}
