// Staging-path microbenchmark for the tiled GEMM (MI355X): what does ONE CU sustain from L2 into LDS / VGPRs, by instruction
// form and access shape, with every CU doing the same (one workgroup per CU, NW waves, batches of 8 x 1 KiB pieces per wave,
// DEPTH batches in flight)?  Built by tools/build_tools.sh into tools/bin/dma_bench; prints one line per variant:
//   bytes / clk / CU (s_memtime) and GB/s / CU (wall).
//   mode 0  global_load_lds_dwordx4, contiguous KiB          (the W pieces of gemm.hip / gemm_w4.hip)
//        1  global_load_lds_dwordx4, 8 rows x 128 B           (the x pieces)
//        2  buffer_load_dwordx4 ... offen lds, contiguous      (SGPR base + one 32-bit offset VGPR: hipBLASLt's form)
//        3  buffer ... lds, 8 rows x 128 B
//        4  buffer ... lds contiguous, nt
//        5  buffer ... lds contiguous, sc0 sc1
//        6  global_load_dwordx4 -> VGPR, contiguous
//        7  global_load_dwordx4 -> VGPR, 8 rows x 128 B
//        8  global_load_dwordx4 -> VGPR -> ds_write_b128, contiguous
//        9  half the pieces as mode 2, half as mode 8 (are the two paths independent?)
//       10  mode 2 + 16 ds_read_b128 per batch (fragment-read traffic beside the DMA)
//       11  mode 0 with 16-row x 64-B pieces (the half-line x pieces of round 3)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define REGION (2u << 20)          // bytes every workgroup cycles through (L2-resident)
#define ROWSTRIDE 7168             // K = 3584 bf16

__device__ __forceinline__ void gload(u32x4& dst, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p)); }
__device__ __forceinline__ void touch(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void use(const u32x4& v) { asm volatile("" ::"v"(v)); }

template <int MODE>
__device__ __forceinline__ uint32_t piece_off(int lane, uint32_t lin) {   // byte offset of this lane's 16 bytes of piece `lin`
    if constexpr (MODE == 1 || MODE == 3 || MODE == 7) {
        const uint32_t rowblk = lin & 31, kblk = (lin >> 5) % (ROWSTRIDE / 128);       // 32 row blocks of 8 rows = 256 rows
        return (rowblk * 8 + (lane >> 3)) * ROWSTRIDE + kblk * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    } else if constexpr (MODE == 11) {
        const uint32_t rowblk = lin & 15, kblk = (lin >> 4) % (ROWSTRIDE / 64);
        return (rowblk * 16 + (lane & 15)) * ROWSTRIDE + kblk * 64 + ((lane >> 4) << 4);
    } else {
        return (lin * 1024u) % REGION + lane * 16;
    }
}

template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void dma_kernel(const char* __restrict__ src, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NP = 8;
    char* myl = smem + (wave * (DEPTH * NP * 1024)) % (128 * 1024 - DEPTH * NP * 1024 + 1024);   // (8 waves x depth 3 would need 192 KiB: regions may overlap, nobody reads the data)
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, REGION, 0x00020000);
    const uint32_t wgofs = ((blockIdx.x >> 3) & 3) * 977;         // the 32 workgroups of an XCD: groups of 8 share their lines
    u32x4 regs[2][NP];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < NP; ++p) regs[b][p] = (u32x4){0, 0, 0, 0};
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    auto issue = [&](int it, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const uint32_t lin = ((uint32_t)it * NW + wave) * NP + p + wgofs;
            const uint32_t off = piece_off<MODE>(lane, lin) % (REGION - 16);
            char* dst = myl + ((it % DEPTH) * NP + p) * 1024;
            constexpr bool vg = MODE == 6 || MODE == 7 || MODE == 8;
            if constexpr (MODE == 0 || MODE == 1 || MODE == 11) {
                __builtin_amdgcn_global_load_lds((const void*)(src + off), (lds_t)dst, 16, 0, 0);
            } else if constexpr (MODE == 2 || MODE == 3 || MODE == 10) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, off, 0, 0, 0);
            } else if constexpr (MODE == 4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, off, 0, 0, 2);
            } else if constexpr (MODE == 5) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, off, 0, 0, 17);
            } else if constexpr (vg) {
                gload(regs[set][p], src + off);
            } else if constexpr (MODE == 9) {
                if (p < NP / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)dst, 16, off, 0, 0, 0);
                else gload(regs[set][p], src + off);
            }
        }
    };
    auto retire = [&](int it, auto SET) {       // batch `it` has landed (caller waited)
        constexpr int set = decltype(SET)::value;
        if constexpr (MODE == 8 || MODE == 9) {
#pragma unroll
            for (int p = (MODE == 9 ? NP / 2 : 0); p < NP; ++p) {
                touch(regs[set][p]);
                *reinterpret_cast<u32x4*>(myl + ((it % DEPTH) * NP + p) * 1024 + lane * 16) = regs[set][p];
            }
        } else if constexpr (MODE == 6 || MODE == 7) {
#pragma unroll
            for (int p = 0; p < NP; ++p) touch(regs[set][p]);
        } else if constexpr (MODE == 10) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                u32x4 v = *reinterpret_cast<const u32x4*>(smem + ((q * 5 + wave) % 120) * 1024 + lane * 16);
                use(v);
            }
        }
    };
    static_assert(DEPTH == 2 || DEPTH == 3, "depth");
    // prologue: DEPTH - 1 batches in flight
    issue(0, std::integral_constant<int, 0>{});
    if constexpr (DEPTH == 3) issue(1, std::integral_constant<int, 1>{});
    for (int it = DEPTH - 1; it < iters; it += 2) {
        if constexpr (DEPTH == 2) {
            issue(it, std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            retire(it - 1, std::integral_constant<int, 0>{});
            issue(it + 1, std::integral_constant<int, 0>{});
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            retire(it, std::integral_constant<int, 1>{});
        } else {
            // (register-destination modes would need three register sets at depth 3: the host runs them at depth 2 only)
            issue(it, std::integral_constant<int, 0>{});
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            retire(it - 2, std::integral_constant<int, 0>{});
            issue(it + 1, std::integral_constant<int, 1>{});
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            retire(it - 1, std::integral_constant<int, 1>{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    acc += reinterpret_cast<float*>(smem)[tid];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, int NW, int DEPTH>
static void run(const char* src, unsigned long long* cyc, float* sink, int iters, const char* name) {
    const size_t lds = 128 * 1024;   // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<MODE, NW, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dma_kernel<MODE, NW, DEPTH>), dim3(grid), dim3(NW * 64), lds, 0, src, iters, cyc, sink);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((dma_kernel<MODE, NW, DEPTH>), dim3(grid), dim3(NW * 64), lds, 0, src, iters, cyc, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= grid;
    const double bytes = (double)iters * NW * 8 * 1024;          // per workgroup (= per CU)
    const double us = ms * 1e3 / reps;
    printf("mode %2d %-46s NW=%d depth=%d  %7.1f us  %6.2f GB/s/CU  %6.2f TB/s chip   counter: %.0f ticks -> %6.2f B/tick/CU\n", MODE, name, NW, DEPTH, us,
           bytes / us / 1e3, bytes * grid / us / 1e6, avg, bytes / avg);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    char* src;
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&src, REGION + 4096);
    hipMemset(src, 1, REGION + 4096);
    hipMalloc(&cyc, 256 * sizeof(unsigned long long));
    hipMalloc(&sink, 64);
#define RUN(M, NW, D, NAME) run<M, NW, D>(src, cyc, sink, iters, NAME)
    RUN(0, 4, 2, "global_load_lds contiguous KiB");
    RUN(0, 4, 3, "global_load_lds contiguous KiB");
    RUN(0, 8, 2, "global_load_lds contiguous KiB");
    RUN(0, 8, 3, "global_load_lds contiguous KiB");
    RUN(1, 4, 3, "global_load_lds 8 rows x 128 B");
    RUN(1, 8, 3, "global_load_lds 8 rows x 128 B");
    RUN(11, 4, 3, "global_load_lds 16 rows x 64 B");
    RUN(2, 4, 2, "buffer_load lds contiguous KiB");
    RUN(2, 4, 3, "buffer_load lds contiguous KiB");
    RUN(2, 8, 3, "buffer_load lds contiguous KiB");
    RUN(3, 4, 3, "buffer_load lds 8 rows x 128 B");
    RUN(3, 8, 3, "buffer_load lds 8 rows x 128 B");
    RUN(4, 4, 3, "buffer_load lds contiguous, nt");
    RUN(5, 4, 3, "buffer_load lds contiguous, sc0 sc1");
    RUN(6, 4, 2, "global_load_dwordx4 -> VGPR contiguous");
    RUN(6, 8, 2, "global_load_dwordx4 -> VGPR contiguous");
    RUN(7, 4, 2, "global_load_dwordx4 -> VGPR 8 rows x 128 B");
    RUN(7, 8, 2, "global_load_dwordx4 -> VGPR 8 rows x 128 B");
    RUN(8, 4, 2, "global_load_dwordx4 -> VGPR -> ds_write_b128");
    RUN(8, 8, 2, "global_load_dwordx4 -> VGPR -> ds_write_b128");
    RUN(9, 4, 2, "half buffer lds + half VGPR + ds_write");
    RUN(9, 8, 2, "half buffer lds + half VGPR + ds_write");
    RUN(10, 4, 3, "buffer_load lds + 16 ds_read_b128 per batch");
    return 0;
}
