// Cost of a software grid barrier on MI355X (one workgroup per CU, 512 threads): the building block of a
// persistent decode-step kernel.  Build: hipcc --offload-arch=gfx950 -O3 tools/gridbar_bench.hip -o tools/bin/gridbar_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default for global atomics
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 20000000) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// variant B: per-XCD counters (block i runs on XCD i % 8), the last arriver of a group bumps the top counter
__device__ __forceinline__ bool grid_barrier_h(unsigned* ctr, unsigned epoch, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned nb = gridDim.x, grp = blockIdx.x & 7, gsize = (nb - grp + 7) / 8;
        unsigned old = __hip_atomic_fetch_add(ctr + 32 * (1 + grp), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gsize * epoch - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8 * epoch) {
            if (++spins > 20000000) { *err = 1; ok = false; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

// variant C: epoch flags, one 64-byte line per block; wave 0 of every block polls all of them
__device__ __forceinline__ bool grid_barrier_f(unsigned* flags, unsigned epoch, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + 16 * blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned nb = gridDim.x;
        long spins = 0;
        for (;;) {
            bool done = true;
            for (unsigned b = threadIdx.x; b < nb; b += 64)
                done &= __hip_atomic_load(flags + 16 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
            if (__all(done)) break;
            if (++spins > 2000000) { *err = 1; ok = false; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

// mode 0: barrier only.  mode 1: every block writes a value before the barrier and reads its neighbour's after
// (checks cross-XCD visibility of plain stores under the release/acquire pair).
template <int VAR>
__global__ __launch_bounds__(512) void bar_kernel(unsigned* counter, int iters, int mode, int* data, int* err, int* bad) {
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        if (mode == 1 && threadIdx.x < 64) data[(it & 1) * nb * 64 + blockIdx.x * 64 + threadIdx.x] = it * 1000 + blockIdx.x;
        bool ok;
        if (VAR == 0) ok = grid_barrier(counter, nb * (unsigned)(it + 1), err);
        else if (VAR == 1) ok = grid_barrier_h(counter, (unsigned)(it + 1), err);
        else ok = grid_barrier_f(counter, (unsigned)(it + 1), err);
        if (!ok) return;
        if (mode == 1 && threadIdx.x < 64) {
            unsigned nbh = (blockIdx.x + 37) % nb;
            int v = data[(it & 1) * nb * 64 + nbh * 64 + threadIdx.x];
            if (v != it * 1000 + (int)nbh) atomicAdd(bad, 1);
        }
    }
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    int nb = p.multiProcessorCount;
    printf("CUs %d\n", nb);
    unsigned* counter; int *data, *err, *bad;
    CK(hipMalloc(&counter, 65536)); CK(hipMalloc(&data, 2 * nb * 64 * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 3; ++var)
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 65536)); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(nb), dim3(512), 0, 0, counter, iters, mode, data, err, bad);
            if (var == 1) hipLaunchKernelGGL(bar_kernel<1>, dim3(nb), dim3(512), 0, 0, counter, iters, mode, data, err, bad);
            if (var == 2) hipLaunchKernelGGL(bar_kernel<2>, dim3(nb), dim3(512), 0, 0, counter, iters, mode, data, err, bad);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr, hbad; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            printf("variant %d mode %d: %d barriers in %.3f ms -> %.2f us/barrier  (timeout flag %d, stale reads %d)\n", var, mode, iters, ms, ms * 1e3 / iters, herr, hbad);
        }
    }
    return 0;
}
