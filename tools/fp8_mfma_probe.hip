// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950: (1) operand layout check against a host fp32 product,
// (2) issue-rate microbenchmark.  Build: hipcc --offload-arch=gfx950 -O3 tools/fp8_mfma_probe.hip -o tools/bin/fp8_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static float e4m3_to_float(uint8_t v) {   // OCP e4m3fn
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

// hypothesis: lane (r = l & 15, g = l >> 4) holds A[r][g*32 .. g*32+31] and B[n = r][g*32 .. +31]; D[g*4+reg][l&15]
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a = *reinterpret_cast<const i32x8*>(A + r * 128 + g * 32);
    i32x8 b = *reinterpret_cast<const i32x8*>(B + r * 128 + g * 32);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 4; ++i) D[(g * 4 + i) * 16 + r] = c[i];
}

__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    std::vector<uint8_t> hA(16 * 128), hB(16 * 128);
    srand(1);
    for (auto& v : hA) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }   // no NaN codes
    for (auto& v : hB) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x30; }
    uint8_t *dA, *dB; float* dD;
    CK(hipMalloc(&dA, hA.size())); CK(hipMalloc(&dB, hB.size())); CK(hipMalloc(&dD, 256 * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> hD(256);
    CK(hipMemcpy(hD.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0;
            for (int k = 0; k < 128; ++k) ref += (double)e4m3_to_float(hA[m * 128 + k]) * e4m3_to_float(hB[n * 128 + k]);
            maxerr = fmax(maxerr, fabs(ref - hD[m * 16 + n]));
            maxref = fmax(maxref, fabs(ref));
        }
    printf("layout check: max |err| %.4g vs max |ref| %.4g  -> %s\n", maxerr, maxref, maxerr <= 1e-4 * maxref ? "OK (A[r][g*32..], B[n][g*32..], D[g*4+i][l&15])" : "MISMATCH");
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount * 4, iters = 20000;
    float* dO; CK(hipMalloc(&dO, blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dO, 100);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, dO, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 * iters * 4 * 2.0 * 16 * 16 * 128;
    printf("rate: %.1f TFLOP/s dense e4m3 (16x16x128, 4 waves/SIMD x 4 chains, constant operands)\n", flops / (ms * 1e-3) / 1e12);
    return 0;
}
