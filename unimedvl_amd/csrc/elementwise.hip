// HBM-bound row kernels: norms, gathers, q/k-norm + RoPE + KV append, argmax.
// One wavefront per row (or per head); 16-byte bf16x8 accesses; wave64 shuffles.
#include "common.h"
#include "../../include/unimedvl_hip.h"

// ----------------------------------------------------------------------------- RMSNorm
// modeling_qwen2.py:89-94: h = x.float(); h = h * rsqrt(mean(h^2) + eps); out = w * h.to(bf16)
// 4 waves per block, one row per wave; row held in registers when H <= 64*8*MAXV.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      const bf16_t* __restrict__ wg, const int32_t* __restrict__ expert,
                                                      bf16_t* __restrict__ out, int T, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const bf16_t* xr = x + (int64_t)row * H;
    const bf16_t* wr = (expert && expert[row]) ? wg : w;
    bf16x8 v[MAXV];
    float ss = 0.f;
    const int nv = H / 8;  // H % 8 == 0
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 64 + lane;
        if (c < nv) {
            v[i] = ldg_frag(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = bf2f((bf16_t)v[i][j]);
                ss += f * f;
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrt_ieee(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 64 + lane;
        if (c < nv) {
            bf16x8 ww = ldg_frag(wr + c * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float h = rbf(bf2f((bf16_t)v[i][j]) * rstd);
                o[j] = (short)f2bf(bf2f((bf16_t)ww[j]) * h);
            }
            *reinterpret_cast<bf16x8*>(out + (int64_t)row * H + c * 8) = o;
        }
    }
}

// Few rows (decode): latency bound, so one 256-thread workgroup per row with the x AND weight
// loads issued together up front (one memory round trip) and a single LDS exchange.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_rowblock_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               const bf16_t* __restrict__ wg, const int32_t* __restrict__ expert,
                                                               bf16_t* __restrict__ out, int H, float eps) {
    __shared__ float part[4];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * H;
    const bf16_t* wr = (expert && expert[row]) ? wg : w;
    const int nv = H / 8;
    bf16x8 v[MAXV], ww[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + threadIdx.x;
        v[i] = c < nv ? ldg_frag(xr + c * 8) : zero_frag();
        ww[i] = c < nv ? ldg_frag(wr + c * 8) : zero_frag();
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f((bf16_t)v[i][j]);
            ss += f * f;
        }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = (part[0] + part[1]) + (part[2] + part[3]);
    const float rstd = rsqrt_ieee(tot / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + threadIdx.x;
        if (c < nv) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)ww[i][j]) * rbf(bf2f((bf16_t)v[i][j]) * rstd));
            *reinterpret_cast<bf16x8*>(out + (int64_t)row * H + c * 8) = o;
        }
    }
}

extern "C" int umv_rmsnorm_bf16(const uint16_t* x, const uint16_t* w, const uint16_t* w_gen, const int32_t* expert,
                                uint16_t* out, int T, int H, float eps, umv_stream_t stream) {
    UMV_CHECK(x && w && out, UMV_ERR_ARG, "rmsnorm: null pointer");
    UMV_CHECK(H % 8 == 0 && H <= 64 * 8 * 16, UMV_ERR_ARG, "rmsnorm: H=%d unsupported", H);
    UMV_CHECK(!expert || w_gen, UMV_ERR_ARG, "rmsnorm: expert routing without w_gen");
    if (T == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    if (T <= 64 && H >= 1024) {
        if (H <= 256 * 8 * 2)
            hipLaunchKernelGGL((rmsnorm_rowblock_kernel<2>), dim3(T), dim3(256), 0, s, x, w, w_gen, expert, out, H, eps);
        else
            hipLaunchKernelGGL((rmsnorm_rowblock_kernel<4>), dim3(T), dim3(256), 0, s, x, w, w_gen, expert, out, H, eps);
        UMV_LAUNCH_CHECK();
        return UMV_OK;
    }
    dim3 grid((T + 3) / 4), block(256);
    if (H <= 512 * 2)
        hipLaunchKernelGGL((rmsnorm_kernel<2>), grid, block, 0, s, x, w, w_gen, expert, out, T, H, eps);
    else if (H <= 512 * 8)
        hipLaunchKernelGGL((rmsnorm_kernel<8>), grid, block, 0, s, x, w, w_gen, expert, out, T, H, eps);
    else
        hipLaunchKernelGGL((rmsnorm_kernel<16>), grid, block, 0, s, x, w, w_gen, expert, out, T, H, eps);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// Consumer of a split-K decode GEMM (umv_gemm_args.k_splits): finishes o_proj / down_proj and runs the next RMSNorm in
// one launch.   seq[t,:] = bf16( bf16(sum_s P[s][t,:]) + seq[t,:] )   (the GEMM output rounding, then the residual add:
// qwen2_navit.py:873-874,897-898), splits added in order 0..S-1;   out[t,:] = w * bf16(seq * rstd)   (modeling_qwen2.py:89-94)
// NS > 0: that many splits, known at compile time so that all their loads are requested before the first add (a run-time
// loop costs one L2 round trip per split on this latency-bound kernel); NS = 0: S of them at run time.
template <int MAXV, int NS>
__global__ __launch_bounds__(256) void residual_rmsnorm_kernel(const float* __restrict__ P, int S, int64_t sstride, int64_t ldp,
                                                               bf16_t* __restrict__ seq, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ out, int H, float eps) {
    __shared__ float part[4];
    const int row = blockIdx.x;
    const int nv = H / 8;
    bf16x8 v[MAXV], ww[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + threadIdx.x;
        v[i] = zero_frag();
        ww[i] = zero_frag();
        if (c < nv) {
            const bf16x8 res = ldg_frag(seq + (int64_t)row * H + c * 8);
            ww[i] = ldg_frag(w + c * 8);
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* p = P + (int64_t)row * ldp + c * 8;
            if constexpr (NS > 0) {
                f32x4 a0[NS], a1[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a0[s] = *reinterpret_cast<const f32x4*>(p + s * sstride);
                    a1[s] = *reinterpret_cast<const f32x4*>(p + s * sstride + 4);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    acc[0] += a0[s].x; acc[1] += a0[s].y; acc[2] += a0[s].z; acc[3] += a0[s].w;
                    acc[4] += a1[s].x; acc[5] += a1[s].y; acc[6] += a1[s].z; acc[7] += a1[s].w;
                }
            } else
            for (int s = 0; s < S; ++s) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(p + s * sstride);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + s * sstride + 4);
                acc[0] += a0.x; acc[1] += a0.y; acc[2] += a0.z; acc[3] += a0.w;
                acc[4] += a1.x; acc[5] += a1.y; acc[6] += a1.z; acc[7] += a1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = rbf(rbf(acc[j]) + bf2f((bf16_t)res[j]));
                v[i][j] = (short)f2bf(f);
                ss += f * f;
            }
            *reinterpret_cast<bf16x8*>(seq + (int64_t)row * H + c * 8) = v[i];
        }
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = (part[0] + part[1]) + (part[2] + part[3]);
    const float rstd = rsqrt_ieee(tot / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + threadIdx.x;
        if (c < nv) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)ww[i][j]) * rbf(bf2f((bf16_t)v[i][j]) * rstd));
            *reinterpret_cast<bf16x8*>(out + (int64_t)row * H + c * 8) = o;
        }
    }
}

extern "C" int umv_residual_rmsnorm_bf16(const float* partials, int n_splits, int64_t split_stride, int64_t ldp, uint16_t* seq,
                                         const uint16_t* w, uint16_t* out, int T, int H, float eps, umv_stream_t stream) {
    UMV_CHECK(partials && seq && w && out, UMV_ERR_ARG, "residual_rmsnorm: null pointer");
    UMV_CHECK(n_splits >= 1 && n_splits <= 64 && split_stride >= 0 && ldp >= H, UMV_ERR_ARG, "residual_rmsnorm: bad split layout");
    UMV_CHECK(H % 8 == 0 && H <= 256 * 8 * 4 && (ldp % 4) == 0 && (split_stride % 4) == 0, UMV_ERR_ARG,
              "residual_rmsnorm: H=%d (multiple of 8, <= 8192) / ldp / split_stride (multiples of 4) unsupported", H);
    if (T == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
#define UMV_RRN_LAUNCH(MAXV, NS) \
    hipLaunchKernelGGL((residual_rmsnorm_kernel<MAXV, NS>), dim3(T), dim3(256), 0, s, partials, n_splits, split_stride, ldp, seq, w, out, H, eps)
    if (H <= 256 * 8 * 2) {
        switch (n_splits) {
            case 2: UMV_RRN_LAUNCH(2, 2); break;
            case 3: UMV_RRN_LAUNCH(2, 3); break;
            case 4: UMV_RRN_LAUNCH(2, 4); break;
            case 6: UMV_RRN_LAUNCH(2, 6); break;      // (65..128 samples: 6 / 8 / 8 splits)
            case 8: UMV_RRN_LAUNCH(2, 8); break;
            default: UMV_RRN_LAUNCH(2, 0); break;
        }
    } else {
        UMV_RRN_LAUNCH(4, 0);
    }
#undef UMV_RRN_LAUNCH
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- LayerNorm
// F.layer_norm on bf16 (siglip_navit.py:283,296,370): fp32 statistics, one rounding to bf16.
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int T, int H,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const bf16_t* xr = x + (int64_t)row * H;
    bf16x8 v[MAXV];
    float s = 0.f;
    const int nv = H / 8;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 64 + lane;
        if (c < nv) {
            v[i] = ldg_frag(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += bf2f((bf16_t)v[i][j]);
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 64 + lane;
        if (c < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = bf2f((bf16_t)v[i][j]) - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrt_ieee(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = i * 64 + lane;
        if (c < nv) {
            bf16x8 ww = ldg_frag(w + c * 8), bb = ldg_frag(b + c * 8), o;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = (short)f2bf((bf2f((bf16_t)v[i][j]) - mean) * rstd * bf2f((bf16_t)ww[j]) + bf2f((bf16_t)bb[j]));
            *reinterpret_cast<bf16x8*>(out + (int64_t)row * H + c * 8) = o;
        }
    }
}

extern "C" int umv_layernorm_bf16(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* out, int T, int H,
                                  float eps, umv_stream_t stream) {
    UMV_CHECK(x && w && b && out, UMV_ERR_ARG, "layernorm: null pointer");
    UMV_CHECK(H % 8 == 0 && H <= 64 * 8 * 8, UMV_ERR_ARG, "layernorm: H=%d unsupported", H);
    if (T == 0) return UMV_OK;
    dim3 grid((T + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (H <= 512 * 3)
        hipLaunchKernelGGL((layernorm_kernel<3>), grid, block, 0, s, x, w, b, out, T, H, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<8>), grid, block, 0, s, x, w, b, out, T, H, eps);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- embedding gather / add rows
__global__ __launch_bounds__(256) void embed_gather_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                           const int32_t* __restrict__ out_rows, bf16_t* __restrict__ out, int T,
                                                           int H) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const bf16_t* src = table + ids[t] * (int64_t)H;
    bf16_t* dst = out + (int64_t)(out_rows ? out_rows[t] : t) * H;
    for (int c = lane; c < H / 8; c += 64) *reinterpret_cast<bf16x8*>(dst + c * 8) = ldg_frag(src + c * 8);
}

extern "C" int umv_embed_gather_bf16(const uint16_t* table, const int64_t* ids, const int32_t* out_rows, uint16_t* out, int T,
                                     int H, umv_stream_t stream) {
    UMV_CHECK(table && ids && out && H % 8 == 0, UMV_ERR_ARG, "embed_gather: bad args");
    if (T == 0) return UMV_OK;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ids, out_rows, out, T, H);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

__global__ __launch_bounds__(256) void add_rows_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ bcast,
                                                       const bf16_t* __restrict__ table, const int64_t* __restrict__ idx,
                                                       const int32_t* __restrict__ out_rows, bf16_t* __restrict__ out, int T, int H) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const bf16_t* ar = a + (int64_t)t * H;
    const bf16_t* tr = table ? table + idx[t] * (int64_t)H : nullptr;
    bf16_t* dst = out + (int64_t)(out_rows ? out_rows[t] : t) * H;
    for (int c = lane; c < H / 8; c += 64) {
        bf16x8 va = ldg_frag(ar + c * 8), o;
        bf16x8 vb = bcast ? ldg_frag(bcast + c * 8) : zero_frag();
        bf16x8 vt = tr ? ldg_frag(tr + c * 8) : zero_frag();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f((bf16_t)va[j]);
            if (bcast) f = rbf(f + bf2f((bf16_t)vb[j]));
            if (tr) f = rbf(f + bf2f((bf16_t)vt[j]));
            o[j] = (short)f2bf(f);
        }
        *reinterpret_cast<bf16x8*>(dst + c * 8) = o;
    }
}

extern "C" int umv_add_rows_bf16(const uint16_t* a, const uint16_t* bcast, const uint16_t* table, const int64_t* idx,
                                 const int32_t* out_rows, uint16_t* out, int T, int H, umv_stream_t stream) {
    UMV_CHECK(a && out && H % 8 == 0, UMV_ERR_ARG, "add_rows: bad args");
    UMV_CHECK(!table || idx, UMV_ERR_ARG, "add_rows: table without idx");
    if (T == 0) return UMV_OK;
    hipLaunchKernelGGL(add_rows_kernel, dim3((T + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, bcast, table, idx, out_rows, out, T, H);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

__device__ __forceinline__ uint64_t shfl_xor_u64_ew(uint64_t v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_xor((int)lo, mask, 64);
    hi = (uint32_t)__shfl_xor((int)hi, mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

// ----------------------------------------------------------------------------- argmax (bf16 logits, lowest index wins)
__global__ __launch_bounds__(1024) void argmax_kernel(const bf16_t* __restrict__ logits, int64_t ld, int64_t* __restrict__ out, int V) {
    __shared__ float smax[16];
    __shared__ int sidx[16];
    const int m = blockIdx.x;
    const bf16_t* row = logits + (int64_t)m * ld;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int nv = V / 8;
    // one workgroup per row is latency bound (19 dependent round trips for V = 152064): request 8 chunks per thread at a
    // time, then compare in index order (same result as the one-at-a-time loop)
    constexpr int UA = 8;
    for (int c0 = threadIdx.x; c0 < nv; c0 += blockDim.x * UA) {
        bf16x8 v[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int c = c0 + u * blockDim.x;
            v[u] = c < nv ? ldg_frag(row + c * 8) : zero_frag();
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int c = c0 + u * blockDim.x;
            if (c < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = bf2f((bf16_t)v[u][j]);
                    int i = c * 8 + j;
                    if (f > best || (f == best && i < bidx) || (f != f && !(best != best))) { best = f; bidx = i; }
                }
            }
        }
    }
    for (int i = nv * 8 + threadIdx.x; i < V; i += blockDim.x) {
        float f = bf2f(row[i]);
        if (f > best || (f == best && i < bidx)) { best = f; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smax[wave] = best; sidx[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (smax[w] > best || (smax[w] == best && sidx[w] < bidx)) { best = smax[w]; bidx = sidx[w]; }
        out[m] = bidx;
    }
}

extern "C" int umv_argmax_bf16(const uint16_t* logits, int64_t ld, int64_t* out_ids, int M, int V, umv_stream_t stream) {
    UMV_CHECK(logits && out_ids && V > 0 && (ld % 8) == 0, UMV_ERR_ARG, "argmax: bad args");
    if (M == 0) return UMV_OK;
    hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, logits, ld, out_ids, V);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- temperature sampling
// bagel.py:1297-1299: probs = softmax(logits / temperature) ; token = multinomial(probs, 1).
// torch draws one sample without replacement as argmax(probs / q), q ~ Exp(1) per element; this kernel
// does the same with a counter-based generator (splitmix64 of (seed, step, row, index)), so the stream is
// reproducible for a given seed but is NOT torch's CPU/CUDA stream.  Rounding follows the bf16 tensors of
// the reference: logits/T -> bf16, softmax output -> bf16.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ float block_reduce_max(float v, float* sm) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sm[w]);
    return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sm[w];
    return r;
}

__global__ __launch_bounds__(1024) void sample_kernel(const bf16_t* __restrict__ logits, int64_t ld, int64_t* __restrict__ out, int V,
                                                      float temp, uint64_t seed, const int64_t* __restrict__ step_ptr) {
    __shared__ float smf[16];
    __shared__ int smi[16];
    const int m = blockIdx.x;
    const bf16_t* row = logits + (int64_t)m * ld;
    const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : 0ull;
    const uint64_t key = splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull) ^ ((uint64_t)m << 32));
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, rbf(bf2f(row[i]) / temp));
    mx = block_reduce_max(mx, smf);
    float sum = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) sum += expf(rbf(bf2f(row[i]) / temp) - mx);
    sum = block_reduce_sum(sum, smf);
    float best = -1.f;
    int bidx = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        const float p = expf(rbf(bf2f(row[i]) / temp) - mx) / sum;   // fp32 probabilities, as autocast's softmax returns them
        const uint64_t h = splitmix64(key + (uint64_t)i);
        const float u = ((float)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);    // [2^-24, 1 - 2^-24], exact in fp32 (the stream of gemm_epilogue.h::epi_gumbel_value)
        const float q = fmaxf(-logf(u), 5.9604645e-8f);                     // Exp(1), q >= 5.9e-8 > 0: a token wins through p / q only
        const float sc = p / q;
        if (sc > best || (sc == best && i < bidx)) { best = sc; bidx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { smf[threadIdx.x >> 6] = best; smi[threadIdx.x >> 6] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (smf[w] > best || (smf[w] == best && smi[w] < bidx)) { best = smf[w]; bidx = smi[w]; }
        out[m] = bidx;
    }
}

extern "C" int umv_sample_bf16(const uint16_t* logits, int64_t ld, int64_t* out_ids, int M, int V, float temperature, uint64_t seed,
                               const int64_t* step, umv_stream_t stream) {
    UMV_CHECK(logits && out_ids && V > 0, UMV_ERR_ARG, "sample: bad args");
    UMV_CHECK(temperature > 0.f, UMV_ERR_ARG, "sample: temperature must be > 0 (got %g)", (double)temperature);
    if (M == 0) return UMV_OK;
    hipLaunchKernelGGL(sample_kernel, dim3(M), dim3(1024), 0, (hipStream_t)stream, logits, ld, out_ids, V, temperature, seed, step);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- fp32 -> bf16 with zero padding
__global__ void cast_pad_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo, int T, int K, int Kp) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * Kp) return;
    int t = (int)(gid / Kp), k = (int)(gid % Kp);
    out[(int64_t)t * ldo + k] = k < K ? f2bf(x[(int64_t)t * ldx + k]) : (bf16_t)0;
}
extern "C" int umv_cast_pad_f32_bf16(const float* x, int64_t ldx, uint16_t* out, int64_t ldo, int T, int K, int Kp,
                                     umv_stream_t stream) {
    UMV_CHECK(x && out && Kp >= K, UMV_ERR_ARG, "cast_pad: bad args");
    if (T == 0) return UMV_OK;
    int64_t total = (int64_t)T * Kp;
    hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ldo, T, K, Kp);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- patchify on the device
// patchify() of data_utils.py:43-50 + the fp32 -> bf16 cast autocast applies in front of the patch-embed linear (siglip_navit.py:190),
// from the transformed [C, H, W] fp32 image: token (ph, pw), column (pp * p + qq) * C + c  =  image[c][ph * p + pp][pw * p + qq], columns
// K = p * p * C .. Kp - 1 zero.  One workgroup per token; the host-side permute of the reference costs 4 ms per 448 x 448 image.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int C, int H, int W, int p, bf16_t* __restrict__ out, int64_t ldo,
                                                       int Kp) {
    const int nw = W / p;
    const int tok = blockIdx.x, ph = tok / nw, pw = tok % nw;
    const int K = p * p * C;
    bf16_t* o = out + (int64_t)tok * ldo;
    for (int col = threadIdx.x; col < Kp; col += blockDim.x) {
        bf16_t v = 0;
        if (col < K) {
            const int c = col % C, t = col / C, qq = t % p, pp = t / p;
            v = f2bf(img[((int64_t)c * H + ph * p + pp) * W + pw * p + qq]);
        }
        o[col] = v;
    }
}
extern "C" int umv_patchify_f32_bf16(const float* img, int C, int H, int W, int p, uint16_t* out, int64_t ldo, int Kp, umv_stream_t stream) {
    UMV_CHECK(img && out, UMV_ERR_ARG, "patchify: null pointer");
    UMV_CHECK(C > 0 && p > 0 && H > 0 && W > 0 && H % p == 0 && W % p == 0, UMV_ERR_ARG, "patchify: image %d x %d x %d is not a whole number of %d-pixel patches", C, H, W, p);
    UMV_CHECK(Kp >= p * p * C && ldo >= Kp, UMV_ERR_ARG, "patchify: Kp %d / ldo %lld too small for %d values per patch", Kp, (long long)ldo, p * p * C);
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((H / p) * (W / p))), dim3(256), 0, (hipStream_t)stream, img, C, H, W, p, out, ldo, Kp);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// Where token t's K row / V^T column goes: the (segment, slot) pair of tok_seg / tok_slot, or - paged KV (umv_qkv_post_args.page_table) -
// (pool page, slot inside the page); every kernel below addresses  base + seg * seg_stride + ... + slot  with these two values
__device__ __forceinline__ int kv_seg(const umv_qkv_post_args& a, int t) {
    const int seg = a.tok_seg[t];
    return a.page_table ? a.page_table[(int64_t)seg * a.page_table_stride + (a.tok_slot[t] >> UMV_KV_PAGE_LOG2)] : seg;
}
__device__ __forceinline__ int kv_slot(const umv_qkv_post_args& a, int t) {
    const int slot = a.tok_slot[t];
    return a.page_table ? (slot & (UMV_KV_PAGE - 1)) : slot;
}

// ----------------------------------------------------------------------------- q/k norm + RoPE + KV append
// One wavefront per (token, head) over the nq + 2*nkv heads of the fused QKV row.
// Lane i owns elements i*EPL.. of the first half and the matching ones of the second
// half (rotate_half pairs x[d] with x[d + hd/2], modeling_qwen2.py:188-192).
// und chain (bf16 tensors, qwen2_navit.py:544-545,576-583):
//    n = bf16(w * bf16(x*rstd));  out = bf16(bf16(n*cos) + bf16(rot(n)*sin))
// gen chain (fp32 tensors, qwen2_navit.py:568-583):
//    n = w * (x*rstd);  out = bf16(n*cos + rot(n)*sin)      (cos/sin are bf16 values)
template <int HD>
__global__ __launch_bounds__(256) void qkv_post_kernel(umv_qkv_post_args a) {
    constexpr int HALF = HD / 2;
    const int lane = threadIdx.x & 63;
    const int nheads = a.nq + 2 * a.nkv;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (int64_t)a.T * nheads) return;
    const int t = (int)(item / nheads);
    const int h = (int)(item % nheads);
    const bf16_t* src = a.qkv + (int64_t)t * nheads * HD + (int64_t)h * HD;
    const int seg = kv_seg(a, t), slot = kv_slot(a, t);
    const bool is_q = h < a.nq, is_k = !is_q && h < a.nq + a.nkv;
    const bool act = lane < HALF;  // HD=128: all 64 lanes; HD=72: 36 lanes
    // The kernel is one dependent chain of memory round trips at decode sizes, so everything that does not depend on the
    // row itself is requested first: position -> cos / sin, expert flag -> norm weight (this kernel always has the norms).
    const bool is_v = !is_q && !is_k;
    const int pos = a.tok_pos[t];
    const bf16_t* nw = is_q ? a.q_norm_w : a.k_norm_w;
    if (a.expert && a.expert[t]) nw = is_q ? a.q_norm_w_gen : a.k_norm_w_gen;
    float c1 = 0.f, s1 = 0.f, c2 = 0.f, s2 = 0.f, w1 = 0.f, w2 = 0.f;
    if (act && !is_v) {
        c1 = bf2f(a.cos_tab[(int64_t)pos * HD + lane]);
        s1 = bf2f(a.sin_tab[(int64_t)pos * HD + lane]);
        c2 = bf2f(a.cos_tab[(int64_t)pos * HD + lane + HALF]);
        s2 = bf2f(a.sin_tab[(int64_t)pos * HD + lane + HALF]);
        w1 = bf2f(nw[lane]);
        w2 = bf2f(nw[lane + HALF]);
    }
    float x1 = 0.f, x2 = 0.f;
    if (act) {
        if (a.qkv_partials) {   // split-K QKV GEMM: x = bf16(sum_s P[s] + bias), the rounding of the GEMM epilogue it replaces
            const int64_t col = (int64_t)h * HD + lane;
            const float* p = a.qkv_partials + (int64_t)t * nheads * HD + col;
            if (a.n_splits <= 4) {   // the usual case: request every split (and the bias) before the first add - one round trip
                float t1[4], t2[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int ss = s < a.n_splits ? s : 0;     // clamped address, masked below: no branch around the loads
                    t1[s] = p[ss * a.split_stride];
                    t2[s] = p[ss * a.split_stride + HALF];
                }
                const float b1 = a.qkv_bias ? bf2f(a.qkv_bias[col]) : 0.f, b2 = a.qkv_bias ? bf2f(a.qkv_bias[col + HALF]) : 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (s < a.n_splits) { x1 += t1[s]; x2 += t2[s]; }
                if (a.qkv_bias) { x1 += b1; x2 += b2; }
            } else if (a.n_splits <= 8) {   // 65..128 samples (6 splits): the same, eight wide
                float t1[8], t2[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int ss = s < a.n_splits ? s : 0;
                    t1[s] = p[ss * a.split_stride];
                    t2[s] = p[ss * a.split_stride + HALF];
                }
                const float b1 = a.qkv_bias ? bf2f(a.qkv_bias[col]) : 0.f, b2 = a.qkv_bias ? bf2f(a.qkv_bias[col + HALF]) : 0.f;
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < a.n_splits) { x1 += t1[s]; x2 += t2[s]; }
                if (a.qkv_bias) { x1 += b1; x2 += b2; }
            } else {
                for (int s = 0; s < a.n_splits; ++s) { x1 += p[s * a.split_stride]; x2 += p[s * a.split_stride + HALF]; }
                if (a.qkv_bias) { x1 += bf2f(a.qkv_bias[col]); x2 += bf2f(a.qkv_bias[col + HALF]); }
            }
            x1 = rbf(x1);
            x2 = rbf(x2);
        } else {
            x1 = bf2f(src[lane]);
            x2 = bf2f(src[lane + HALF]);
        }
    }
    if (!is_q && !is_k) {  // V head: transposed store V^T[seg][kvh][d][slot]
        const int kvh = h - a.nq - a.nkv;
        bf16_t* dst = a.vt_slab + seg * a.v_seg_stride + kvh * a.v_head_stride + slot;
        if (act) {
            dst[(int64_t)lane * a.v_d_stride] = f2bf(x1);
            dst[(int64_t)(lane + HALF) * a.v_d_stride] = f2bf(x2);
        }
        return;
    }
    float o1 = x1, o2 = x2;
    {
        const bool gen = a.fp32_chain != 0;
        float ss = wave_sum(x1 * x1 + x2 * x2);
        const float rstd = rsqrt_ieee(ss / (float)HD + a.eps);
        if (!gen) {
            float n1 = rbf(w1 * rbf(x1 * rstd));
            float n2 = rbf(w2 * rbf(x2 * rstd));
            o1 = rbf(rbf(n1 * c1) + rbf(-n2 * s1));
            o2 = rbf(rbf(n2 * c2) + rbf(n1 * s2));
        } else {
            float n1 = __fmul_rn(w1, __fmul_rn(x1, rstd));
            float n2 = __fmul_rn(w2, __fmul_rn(x2, rstd));
            o1 = __fadd_rn(__fmul_rn(n1, c1), __fmul_rn(-n2, s1));
            o2 = __fadd_rn(__fmul_rn(n2, c2), __fmul_rn(n1, s2));
        }
    }
    if (is_q) {
        bf16_t* dst = a.q_out + (int64_t)t * a.nq * HD + (int64_t)h * HD;
        if (act) { dst[lane] = f2bf(o1); dst[lane + HALF] = f2bf(o2); }
    } else {
        const int kvh = h - a.nq;
        bf16_t* dst = a.k_slab + seg * a.k_seg_stride + kvh * a.k_head_stride + (int64_t)slot * HD;
        if (act) { dst[lane] = f2bf(o1); dst[lane + HALF] = f2bf(o2); }
    }
}

// No norm / no RoPE (ViT, VAE mid-block attention): split the fused QKV row into q rows and
// K / V^T slab entries, any head_dim.
__global__ __launch_bounds__(256) void qkv_split_kernel(umv_qkv_post_args a) {
    const int lane = threadIdx.x & 63;
    const int HD = a.hd;
    const int nheads = a.nq + 2 * a.nkv;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (int64_t)a.T * nheads) return;
    const int t = (int)(item / nheads);
    const int h = (int)(item % nheads);
    const bf16_t* src = a.qkv + (int64_t)t * nheads * HD + (int64_t)h * HD;
    const int seg = kv_seg(a, t), slot = kv_slot(a, t);
    if (h < a.nq) {
        bf16_t* dst = a.q_out + (int64_t)t * a.nq * HD + (int64_t)h * HD;
        for (int d = lane; d < HD; d += 64) dst[d] = src[d];
    } else if (h < a.nq + a.nkv) {
        bf16_t* dst = a.k_slab + seg * a.k_seg_stride + (h - a.nq) * a.k_head_stride + (int64_t)slot * HD;
        for (int d = lane; d < HD; d += 64) dst[d] = src[d];
    } else {
        bf16_t* dst = a.vt_slab + seg * a.v_seg_stride + (h - a.nq - a.nkv) * a.v_head_stride + slot;
        for (int d = lane; d < HD; d += 64) dst[(int64_t)d * a.v_d_stride] = src[d];
    }
}

// The same split for hd % 8 == 0, eight tokens per workgroup: q and K rows move as 16-byte pieces, and the eight V rows
// meet in LDS so that a thread writes 8 consecutive slots (16 bytes) of one V^T row instead of eight 2-byte stores a
// cache line apart (one wave per (token, head) with 2-byte accesses took 75 us per ViT layer for 112 MB of traffic).
// Groups whose tokens are not 8 consecutive, 8-aligned slots of one segment fall back to element stores.
__global__ __launch_bounds__(256) void qkv_split_tile_kernel(umv_qkv_post_args a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t vs[];      // [8][nkv * hd]
    const int HD = a.hd, CH = HD / 8, nheads = a.nq + 2 * a.nkv;
    const int t0 = blockIdx.x * 8, nt = min(8, a.T - t0);
    const int q_ch = a.nq * CH, qk_ch = (a.nq + a.nkv) * CH, row_ch = nheads * CH, nv = a.nkv * HD;
    // q_out == null: V only - q and K stay where the GEMM wrote them (umv_attn_varlen's q_row_stride / k_key_stride form)
    const int c_lo = a.q_out ? 0 : qk_ch, span = row_ch - c_lo;
    for (int i = threadIdx.x; i < nt * span; i += 256) {
        const int tt = i / span, c = c_lo + (i - tt * span);
        const int t = t0 + tt;
        const bf16x8 v = ldg_frag(a.qkv + (int64_t)t * nheads * HD + (int64_t)c * 8);
        if (c < q_ch) {
            *reinterpret_cast<bf16x8*>(a.q_out + (int64_t)t * a.nq * HD + (int64_t)c * 8) = v;
        } else if (c < qk_ch) {
            const int h = (c - q_ch) / CH, cc = (c - q_ch) - h * CH;
            *reinterpret_cast<bf16x8*>(a.k_slab + kv_seg(a, t) * a.k_seg_stride + h * a.k_head_stride + (int64_t)kv_slot(a, t) * HD + cc * 8) = v;
        } else {
            *reinterpret_cast<bf16x8*>(vs + tt * nv + (c - qk_ch) * 8) = v;
        }
    }
    __syncthreads();
    const int seg0 = kv_seg(a, t0), slot0 = kv_slot(a, t0);
    bool run8 = nt == 8 && (slot0 & 7) == 0;
    for (int tt = 1; tt < nt && run8; ++tt) run8 = kv_seg(a, t0 + tt) == seg0 && kv_slot(a, t0 + tt) == slot0 + tt;
    if (run8) {
        for (int e = threadIdx.x; e < nv; e += 256) {
            const int h = e / HD, d = e - h * HD;
            bf16x8 o;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) o[tt] = (short)vs[tt * nv + e];
            *reinterpret_cast<bf16x8*>(a.vt_slab + seg0 * a.v_seg_stride + h * a.v_head_stride + (int64_t)d * a.v_d_stride + slot0) = o;
        }
    } else {
        for (int i = threadIdx.x; i < nt * nv; i += 256) {
            const int tt = i / nv, e = i - tt * nv;
            const int h = e / HD, d = e - h * HD;
            const int t = t0 + tt;
            a.vt_slab[kv_seg(a, t) * a.v_seg_stride + h * a.v_head_stride + (int64_t)d * a.v_d_stride + kv_slot(a, t)] = vs[tt * nv + e];
        }
    }
}

// V-only split (the cache-less SigLIP tower: q and K are read by the attention kernel where the QKV GEMM wrote them) as an
// in-register transpose: a thread owns 8 tokens x 8 dims - eight 16-byte loads (one per token), an 8 x 8 transpose of bf16
// pairs with v_perm_b32, eight 16-byte stores (one per dim: 8 consecutive slots of a V^T row).  A wave is 8 dim-octets x 8
// token-octets, so a load instruction covers 8 x 128 contiguous bytes of 8 token rows and a store instruction 8 x 128
// contiguous bytes of 8 V^T rows: whole cache lines both ways (the LDS version above writes 16 bytes per V^T row and
// workgroup: 19.6 us per ViT layer for 2 x 18.9 MB).  Token octets that are not 8 consecutive, 8-aligned slots of one
// segment fall back to element stores.
__global__ __launch_bounds__(256) void v_transpose_kernel(umv_qkv_post_args a) {
    const int HD = a.hd, nheads = a.nq + 2 * a.nkv, nv = a.nkv * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cl = lane & 7, jl = lane >> 3;
    const int e0 = (blockIdx.y * 8 + cl) * 8;                       // first of this thread's 8 V dims (over all kv heads)
    const int t0 = ((int)blockIdx.x * 4 + wave) * 64 + jl * 8;      // first of its 8 tokens
    if (e0 >= nv || t0 >= a.T) return;
    const int nt = min(8, a.T - t0);
    const bf16_t* src = a.qkv + (int64_t)t0 * nheads * HD + (int64_t)(a.nq + a.nkv) * HD + e0;
    u32x4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = i < nt ? *reinterpret_cast<const u32x4*>(src + (int64_t)i * nheads * HD) : (u32x4){0u, 0u, 0u, 0u};
    const int seg0 = kv_seg(a, t0), slot0 = kv_slot(a, t0);
    bool run8 = nt == 8 && (slot0 & 7) == 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < nt) run8 = run8 && kv_seg(a, t0 + i) == seg0 && kv_slot(a, t0 + i) == slot0 + i;
    const int h = e0 / HD, d0 = e0 - h * HD;                        // HD % 8 == 0: the 8 dims lie in one head
    if (run8) {
        bf16_t* dst = a.vt_slab + seg0 * a.v_seg_stride + h * a.v_head_stride + (int64_t)d0 * a.v_d_stride + slot0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            // bytes of {hi = r[2k+1], lo = r[2k]}.dword[d >> 1]: low halves 0x05040100, high halves 0x07060302
            const uint32_t sel = (d & 1) ? 0x07060302u : 0x05040100u;
            u32x4 o;
            o.x = __builtin_amdgcn_perm(r[1][d >> 1], r[0][d >> 1], sel);
            o.y = __builtin_amdgcn_perm(r[3][d >> 1], r[2][d >> 1], sel);
            o.z = __builtin_amdgcn_perm(r[5][d >> 1], r[4][d >> 1], sel);
            o.w = __builtin_amdgcn_perm(r[7][d >> 1], r[6][d >> 1], sel);
            *reinterpret_cast<u32x4*>(dst + (int64_t)d * a.v_d_stride) = o;
        }
    } else {
        for (int i = 0; i < nt; ++i) {
            bf16_t* dst = a.vt_slab + kv_seg(a, t0 + i) * a.v_seg_stride + h * a.v_head_stride + (int64_t)d0 * a.v_d_stride + kv_slot(a, t0 + i);
#pragma unroll
            for (int d = 0; d < 8; ++d) dst[(int64_t)d * a.v_d_stride] = (bf16_t)(r[i][d >> 1] >> ((d & 1) * 16));
        }
    }
}

// q / k heads of a LONG forward (prefill, flow passes: T >= 64 rows of bf16 qkv, head_dim 128): qkv_post_kernel's arithmetic with
// 8-byte accesses - 16 lanes per (token, head), lane `sub` owns elements 4 sub .. 4 sub + 3 of the first half and the matching
// ones of the second half (rotate_half pairs x[d] with x[d + 64]), four items per wave.  The per-(token, head) wave with 2-byte
// accesses took 25-27 us per layer of a guided flow pass (2064 rows) and 100 us per layer of an 8-image prefill.  The row sum of
// squares follows qkv_post_kernel's butterfly exactly (lane bits 5, 4, 3, 2 there are sub bits 3, 2, 1, 0 here, lane bits 1, 0
// the element index), so the results are bit-identical and a decode step still equals the prefill of the same token.
// V heads go through v_transpose_kernel.
__global__ __launch_bounds__(256) void qk_post_vec128_kernel(umv_qkv_post_args a) {
    constexpr int HD = 128, HALF = 64;
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int nqk = a.nq + a.nkv, nheads = a.nq + 2 * a.nkv;
    const int64_t item = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = item < (int64_t)a.T * nqk;
    const int t = live ? (int)(item / nqk) : 0;
    const int h = live ? (int)(item % nqk) : 0;
    const bool is_q = h < a.nq;
    const int pos = a.tok_pos[t];
    const bf16_t* nw = is_q ? a.q_norm_w : a.k_norm_w;
    if (a.expert && a.expert[t]) nw = is_q ? a.q_norm_w_gen : a.k_norm_w_gen;
    auto ld4 = [](const bf16_t* p, float* o) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(p);
        o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
        o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
    };
    const bf16_t* src = a.qkv + (int64_t)t * nheads * HD + (int64_t)h * HD + 4 * sub;
    float x1[4], x2[4], c1[4], s1[4], c2[4], s2[4], w1[4], w2[4];
    ld4(src, x1); ld4(src + HALF, x2);
    ld4(a.cos_tab + (int64_t)pos * HD + 4 * sub, c1); ld4(a.sin_tab + (int64_t)pos * HD + 4 * sub, s1);
    ld4(a.cos_tab + (int64_t)pos * HD + HALF + 4 * sub, c2); ld4(a.sin_tab + (int64_t)pos * HD + HALF + 4 * sub, s2);
    ld4(nw + 4 * sub, w1); ld4(nw + HALF + 4 * sub, w2);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float q = x1[e] * x1[e] + x2[e] * x2[e];
        q += row_xor<8>(q);      // lane bit 5 of the per-head wave
        q += row_xor<4>(q);      // bit 4
        q += row_xor<2>(q);      // bit 3
        q += row_xor<1>(q);      // bit 2
        v[e] = q;
    }
    const float ss = (v[0] + v[2]) + (v[1] + v[3]);       // bits 1, 0
    const float rstd = rsqrt_ieee(ss / (float)HD + a.eps);
    const bool gen = a.fp32_chain != 0;
    float o1[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!gen) {
            const float n1 = rbf(w1[e] * rbf(x1[e] * rstd)), n2 = rbf(w2[e] * rbf(x2[e] * rstd));
            o1[e] = rbf(rbf(n1 * c1[e]) + rbf(-n2 * s1[e]));
            o2[e] = rbf(rbf(n2 * c2[e]) + rbf(n1 * s2[e]));
        } else {
            const float n1 = __fmul_rn(w1[e], __fmul_rn(x1[e], rstd)), n2 = __fmul_rn(w2[e], __fmul_rn(x2[e], rstd));
            o1[e] = __fadd_rn(__fmul_rn(n1, c1[e]), __fmul_rn(-n2, s1[e]));
            o2[e] = __fadd_rn(__fmul_rn(n2, c2[e]), __fmul_rn(n1, s2[e]));
        }
    }
    if (!live) return;
    bf16_t* dst = is_q ? a.q_out + (int64_t)t * a.nq * HD + (int64_t)h * HD
                       : a.k_slab + kv_seg(a, t) * a.k_seg_stride + (h - a.nq) * a.k_head_stride + (int64_t)kv_slot(a, t) * HD;
    u32x2 p1, p2;
    p1.x = pack2bf(o1[0], o1[1]); p1.y = pack2bf(o1[2], o1[3]);
    p2.x = pack2bf(o2[0], o2[1]); p2.y = pack2bf(o2[2], o2[3]);
    *reinterpret_cast<u32x2*>(dst + 4 * sub) = p1;
    *reinterpret_cast<u32x2*>(dst + HALF + 4 * sub) = p2;
}

extern "C" int umv_qkv_post(const umv_qkv_post_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap, UMV_ERR_ARG, "qkv_post: null args");
    const umv_qkv_post_args& a = *ap;
    const bool v_only = !a.q_out && !a.k_slab;      // plain split of V only (no norm / RoPE, head_dim % 8 == 0): q and K are read in place
    UMV_CHECK((a.qkv || a.qkv_partials) && (v_only || (a.q_out && a.k_slab)) && a.vt_slab && a.tok_seg && a.tok_slot, UMV_ERR_ARG, "qkv_post: null pointer");
    UMV_CHECK(!v_only || (a.qkv && !a.q_norm_w && (a.hd % 8) == 0 && (size_t)8 * a.nkv * a.hd * sizeof(bf16_t) <= 64 * 1024), UMV_ERR_UNSUPPORTED,
              "qkv_post: the V-only split needs bf16 qkv rows, no norm / RoPE and head_dim %% 8 == 0");
    // v_transpose_kernel stores 8 slots of one V^T row with one 16-byte store: every V^T stride must keep those stores aligned
    UMV_CHECK(!v_only || ((a.v_d_stride % 8) == 0 && (a.v_head_stride % 8) == 0 && (a.v_seg_stride % 8) == 0), UMV_ERR_UNSUPPORTED,
              "qkv_post: the V-only split needs V^T strides that are multiples of 8 elements (d %lld, head %lld, segment %lld)",
              (long long)a.v_d_stride, (long long)a.v_head_stride, (long long)a.v_seg_stride);
    UMV_CHECK(!a.qkv_partials || (a.q_norm_w && a.n_splits >= 1 && a.n_splits <= 64), UMV_ERR_ARG,
              "qkv_post: fp32 partial input needs the norm + RoPE path and 1 <= n_splits <= 64");
    UMV_CHECK(!a.q_norm_w || (a.k_norm_w && a.cos_tab && a.sin_tab && a.tok_pos), UMV_ERR_ARG, "qkv_post: norm without rope tables");
    UMV_CHECK(!a.expert || (a.q_norm_w_gen && a.k_norm_w_gen), UMV_ERR_ARG, "qkv_post: expert routing without gen norms");
    UMV_CHECK(!a.page_table || (a.page_table_stride > 0 && a.v_d_stride == UMV_KV_PAGE), UMV_ERR_ARG,
              "qkv_post: paged KV needs page_table_stride > 0 and V^T rows of UMV_KV_PAGE = %d keys (v_d_stride %lld)", UMV_KV_PAGE, (long long)a.v_d_stride);
    if (a.T == 0) return UMV_OK;
    int64_t items = (int64_t)a.T * (a.nq + 2 * a.nkv);
    dim3 grid((unsigned)((items + 3) / 4)), block(256);
    const size_t tile_lds = (size_t)8 * a.nkv * a.hd * sizeof(bf16_t);
    const bool tile_ok = (a.hd % 8) == 0 && tile_lds <= 64 * 1024;
    if (v_only) {       // (hd % 8 == 0 checked above)
        const int nv8 = a.nkv * a.hd / 8;
        hipLaunchKernelGGL(v_transpose_kernel, dim3((unsigned)((a.T + 255) / 256), (unsigned)((nv8 + 7) / 8)), block, 0, (hipStream_t)stream, a);
    } else if (!a.q_norm_w && tile_ok) {
        hipLaunchKernelGGL(qkv_split_tile_kernel, dim3((unsigned)((a.T + 7) / 8)), block, tile_lds, (hipStream_t)stream, a);
    } else if (!a.q_norm_w) {
        hipLaunchKernelGGL(qkv_split_kernel, grid, block, 0, (hipStream_t)stream, a);
    } else {
        UMV_CHECK(a.hd == 128 || a.hd == 72, UMV_ERR_UNSUPPORTED, "qkv_post: head_dim %d unsupported (128, 72)", a.hd);
        static const int vec = [] { const char* e = getenv("UMV_QKV_POST_VEC"); return e ? atoi(e) : 1; }();      // UMV_QKV_POST_VEC=0: the per-(token, head) wave for every size (A/B only; read once, thread-safe)
        if (vec && a.hd == 128 && !a.qkv_partials && a.T >= 64 && (a.v_d_stride % 8) == 0) {
            const int64_t qk_items = (int64_t)a.T * (a.nq + a.nkv);
            hipLaunchKernelGGL(qk_post_vec128_kernel, dim3((unsigned)((qk_items + 15) / 16)), block, 0, (hipStream_t)stream, a);
            const int nv8 = a.nkv * a.hd / 8;
            hipLaunchKernelGGL(v_transpose_kernel, dim3((unsigned)((a.T + 255) / 256), (unsigned)((nv8 + 7) / 8)), block, 0, (hipStream_t)stream, a);
            UMV_LAUNCH_CHECK();
            return UMV_OK;
        }
        // (sending the V heads of a long prefill through the tile kernel and only q / k through this one was measured on the
        // flow passes, T = 2064: 20.4 + 11.4 us against 24.8 us in one kernel - the per-(token, head) wave with 2-byte
        // accesses is the cost here, not the V scatter)
        if (a.hd == 128) hipLaunchKernelGGL((qkv_post_kernel<128>), grid, block, 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((qkv_post_kernel<72>), grid, block, 0, (hipStream_t)stream, a);
    }
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

__global__ void decode_advance_kernel(int32_t* slot, int32_t* pos, int32_t* kv_len, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { slot[b] += 1; pos[b] += 1; kv_len[b] += 1; }
}
extern "C" int umv_decode_advance(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, int B, umv_stream_t stream) {
    UMV_CHECK(tok_slot && tok_pos && kv_len, UMV_ERR_ARG, "decode_advance: null pointer");
    if (B == 0) return UMV_OK;
    hipLaunchKernelGGL(decode_advance_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, tok_slot, tok_pos, kv_len, B);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// End of a decode step in ONE launch: log the token just predicted (pred_ids[s] = ids; in_ids[s + 1] = ids, the token the
// next step is fed - bagel.py:1263,1311-1312), bump slot / position / kv_len and the step counter s.
__global__ __launch_bounds__(256) void decode_step_end_kernel(int32_t* slot, int32_t* pos, int32_t* kv_len, const int64_t* ids,
                                                              int64_t* in_ids, int64_t* pred_ids, int64_t* step_idx, int B, int max_len) {
    const int64_t s = step_idx[0];
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int64_t id = ids[b];
        if (s < max_len) pred_ids[s * B + b] = id;
        if (s + 1 < max_len) in_ids[(s + 1) * B + b] = id;
        slot[b] += 1; pos[b] += 1; kv_len[b] += 1;
    }
    __syncthreads();                 // everyone has read s
    if (threadIdx.x == 0) step_idx[0] = s + 1;
}
extern "C" int umv_decode_step_end(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, const int64_t* ids, int64_t* in_ids,
                                   int64_t* pred_ids, int64_t* step_idx, int B, int max_len, umv_stream_t stream) {
    UMV_CHECK(tok_slot && tok_pos && kv_len && ids && in_ids && pred_ids && step_idx && max_len > 0, UMV_ERR_ARG, "decode_step_end: bad args");
    if (B == 0) return UMV_OK;
    hipLaunchKernelGGL(decode_step_end_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, tok_slot, tok_pos, kv_len, ids, in_ids, pred_ids,
                       step_idx, B, max_len);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// Greedy pick + end of step in one launch: one workgroup per sample takes the maximum of the per-tile keys the lm_head GEMM
// epilogue left (gemm_epilogue.h::argmax_key), then does decode_step_end_kernel's bookkeeping for its sample.  The step counter
// is PER SAMPLE - step_idx[b], all equal - so that no workgroup reads a word another workgroup of the same launch writes
// (rounds 2-3 shared step_idx[0] behind a relaxed ticket; correct on this hardware, not by the memory model).
__global__ __launch_bounds__(256) void decode_step_end_argmax_kernel(int32_t* slot, int32_t* pos, int32_t* kv_len,
                                                                     const uint64_t* __restrict__ part, int n_tiles, int64_t* ids,
                                                                     int64_t* in_ids, int64_t* pred_ids, int64_t* step_idx,
                                                                     int B, int max_len) {
    __shared__ uint64_t sm[4];
    const int b = blockIdx.x;
    const int64_t s = step_idx[b];
    const uint64_t* row = part + (int64_t)b * n_tiles;
    uint64_t best = 0;
    constexpr int UA = 8;      // all loads of a thread in flight together: one round trip for up to 2048 tiles per pass
    for (int c0 = threadIdx.x; c0 < n_tiles; c0 += 256 * UA) {
        uint64_t v[UA];
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            const int c = c0 + u * 256;
            v[u] = c < n_tiles ? row[c] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < UA; ++u) best = v[u] > best ? v[u] : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t ob = shfl_xor_u64_ew(best, o);
        best = ob > best ? ob : best;
    }
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) best = sm[w] > best ? sm[w] : best;
        const int64_t id = (int64_t)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));
        ids[b] = id;
        if (s < max_len) pred_ids[s * B + b] = id;
        if (s + 1 < max_len) in_ids[(s + 1) * B + b] = id;
        slot[b] += 1; pos[b] += 1; kv_len[b] += 1;
        step_idx[b] = s + 1;
    }
}
extern "C" int umv_decode_step_end_argmax(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, const uint64_t* argmax_partial, int n_tiles,
                                          int64_t* ids, int64_t* in_ids, int64_t* pred_ids, int64_t* step_idx, int B,
                                          int max_len, umv_stream_t stream) {
    UMV_CHECK(tok_slot && tok_pos && kv_len && argmax_partial && ids && in_ids && pred_ids && step_idx && max_len > 0 && n_tiles > 0,
              UMV_ERR_ARG, "decode_step_end_argmax: bad args");
    if (B == 0) return UMV_OK;
    hipLaunchKernelGGL(decode_step_end_argmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, tok_slot, tok_pos, kv_len, argmax_partial,
                       n_tiles, ids, in_ids, pred_ids, step_idx, B, max_len);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

