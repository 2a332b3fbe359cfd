// Store-path microbenchmark for the GEMM epilogue (MI355X): what does the chip sustain when every CU stores its output tile at the
// same time?  One workgroup of NW waves per CU writes BYTES per workgroup as whole rows (a wave instruction = 4 rows x 256 B, the
// shape gemm_epilogue.h's phase 2 stores) into its own 256 x 256 bf16 tile of a row-major matrix, `rounds` tiles per workgroup
// back to back (rounds = 1: the burst at the end of a one-round GEMM).  Buffers rotate over > 256 MiB so that the Infinity
// Cache cannot absorb a repeat.  Built by tools/build_tools.sh into tools/bin/store_bench; prints TB/s per variant.
//   mode 0 plain global_store_dwordx4   1 nt   2 sc1   3 sc0 sc1   4 plain, with a 20k-cycle pause between the two halves of a tile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__device__ __forceinline__ void st16(char* p, u32x4 v) {
    if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// N = row length in bf16 of the output matrix; tile (mb, nb) = rows mb*256.., columns nb*256..
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(char* out, int N, int nblocks, int rounds, uint64_t* ticks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        const int tile = blockIdx.x + r * gridDim.x;
        const int mb = tile / nblocks, nb = tile % nblocks;
        // wave w owns the 128 x 128 quadrant (w >> 1, w & 1): 128 rows x 256 B; an instruction covers 4 rows
        char* base = out + ((size_t)(mb * 256 + (wave >> 1) * 128) * N + nb * 256 + (wave & 1) * 128) * 2;
        u32x4 v = {(uint32_t)tile, (uint32_t)lane, 0u, 0u};
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int row = i * 4 + (lane >> 4);
            st16<MODE>(base + (size_t)row * N * 2 + (lane & 15) * 16, v);
            if (MODE == 4 && i == 15) __builtin_amdgcn_s_sleep(127);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int M, int N, int grid, int rounds) {
    const size_t bytes = (size_t)M * N * 2;
    const int nbuf = (int)((600ull << 20) / bytes) + 1;
    char* buf;
    uint64_t* ticks;
    hipMalloc(&buf, bytes * nbuf);
    hipMalloc(&ticks, grid * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, buf + (size_t)(i % nbuf) * bytes, N, N / 256, rounds, ticks);
    hipDeviceSynchronize();
    const int reps = 40;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, buf + (size_t)(i % nbuf) * bytes, N, N / 256, rounds, ticks);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t* h = (uint64_t*)malloc(grid * 8);
    hipMemcpy(h, ticks, grid * 8, hipMemcpyDeviceToHost);
    uint64_t mx = 0, sum = 0;
    for (int i = 0; i < grid; ++i) { mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
    const double us = ms * 1e3 / reps, total = (double)grid * rounds * 131072.0;
    printf("%-22s M=%5d N=%5d grid=%4d rounds=%d  %8.2f us/launch  %6.2f TB/s  (in-kernel counter: mean %.0f max %llu shader cycles, s_memtime)\n", name, M, N, grid,
           rounds, us, total / us / 1e6, (double)sum / grid, (unsigned long long)mx);
    free(h);
    hipFree(buf); hipFree(ticks);
}

int main() {
    // one-round bursts: 256 tiles = 8192 x 2048 (32 MiB), the ViT q/k/v GEMM's first round
    run<0>("plain", 8192, 2048, 256, 1);
    run<1>("nt", 8192, 2048, 256, 1);
    run<2>("sc1", 8192, 2048, 256, 1);
    run<3>("sc0 sc1", 8192, 2048, 256, 1);
    run<0>("plain", 8192, 8192, 256, 4);          // sustained: 4 tiles per workgroup back to back (128 MiB)
    run<1>("nt", 8192, 8192, 256, 4);
    run<0>("plain", 8192, 8192, 1024, 1);         // the same bytes as 1024 workgroups
    run<0>("plain 128 WGs", 8192, 1024, 128, 1);  // half the CUs storing
    run<0>("plain 64 WGs", 8192, 512, 64, 1);
    run<0>("plain 32 WGs", 8192, 256, 32, 1);
    run<4>("plain + pause", 8192, 2048, 256, 1);
    return 0;
}
