// Merge of the split-KV partial results (O, m, l) of the decode attention kernels: shared by attention.hip (product library)
// and attention_decode.hip (experimental library), header-only so that neither library depends on the other.
#pragma once
#include "common.h"

// Merge the nsplit partial (O, m, l) triples of one (token, head) row.  One wavefront per row; the
// (m, l) pairs are read by lanes 0..nsplit-1 in one go and the weighted O sums use independent,
// fully unrolled loads (the kernel is pure latency otherwise).
template <int HD, int MAXS>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ ws, bf16_t* __restrict__ out,
                                                           const int32_t* __restrict__ cu_q, int nseg, int nq, int nsplit,
                                                           int64_t rows_static) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (tok*nq + head)
    // The exact row count lives on the device (cu_q[nseg]); waiting for it before anything else would add a memory round
    // trip to a kernel that is nothing but latency.  The workspace covers the static bound, so the loads below are safe for
    // any row of the grid (clamped), and only the store at the end depends on the count.
    const int64_t valid_rows = (int64_t)cu_q[nseg] * nq;
    const float* base = ws + min(row, rows_static - 1) * nsplit * (HD + 4);
    float m = -INFINITY, l = 0.f;
    if (lane < nsplit) {
        m = base[lane * (HD + 4) + HD];
        l = base[lane * (HD + 4) + HD + 1];
    }
    const float M = wave_max(m);
    const float wgt = (m == -INFINITY) ? 0.f : umv_exp2(m - M);
    const float L = wave_sum(wgt * l);
    constexpr int PER = (HD + 63) / 64;
    float acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = 0.f;
    float vals[MAXS][PER];
#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx)
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int d = i * 64 + lane;
            vals[sidx][i] = (sidx < nsplit && d < HD) ? base[sidx * (HD + 4) + d] : 0.f;
        }
#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) {
        const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wgt), sidx));   // v_readlane, not ds_bpermute
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] += w * vals[sidx][i];
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    if (row >= valid_rows) return;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = i * 64 + lane;
        if (d < HD) out[row * HD + d] = f2bf(acc[i] * inv);
    }
}

static inline int umv_attn_combine_launch(const float* ws, uint16_t* out, const int32_t* cu_q, int nseg, int nq, int hd, int nsplit, int64_t rows,
                            hipStream_t s) {
    if (hd == 128)
        hipLaunchKernelGGL((attn_combine_kernel<128, 32>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ws, out, cu_q, nseg, nq, nsplit, rows);
    else
        hipLaunchKernelGGL((attn_combine_kernel<72, 32>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ws, out, cu_q, nseg, nq, nsplit, rows);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
