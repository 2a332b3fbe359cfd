// EXPERIMENTAL (libunimedvl_hip_experimental.so): measured and not adopted - a weight that is already resident in the 256 MiB
// memory-side cache is not delivered faster than HBM delivers it, and a prefetch branch on a second stream made the decode
// graph slower (profiles/HISTORY.md section 5b).
#include "common.h"
#include "unimedvl_hip_experimental.h"

// ----------------------------------------------------------------------------- weight prefetch into L2 / Infinity Cache
// Decode is a chain of dependent kernels; during the latency-bound ones (norms, RoPE/KV append,
// attention, combine) HBM idles.  This kernel streams the NEXT GEMM's packed weights with ordinary
// (cache-allocating) loads on a parallel stream so that they are resident in the 256 MiB Infinity
// Cache when the GEMM starts.  It computes nothing: the XOR sink only keeps the loads alive.
__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ p, size_t n16, uint32_t* __restrict__ sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && sink) *sink = acc.x;   // practically never taken
}

extern "C" int umv_prefetch(const void* ptr, size_t bytes, int blocks, void* sink, umv_stream_t stream) {
    UMV_CHECK(ptr && blocks > 0, UMV_ERR_ARG, "prefetch: bad args");
    if (bytes < 16) return UMV_OK;
    hipLaunchKernelGGL(prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ptr, bytes / 16, (uint32_t*)sink);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
