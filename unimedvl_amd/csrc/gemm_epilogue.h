// Shared GEMM epilogue for libunimedvl_hip (gfx950): bias / activation / SwiGLU / residual with a bf16 rounding
// exactly where the reference materialises a bf16 tensor (see include/unimedvl_hip.h, umv_gemm_bf16).
#pragma once
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include <type_traits>

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// ----------------------------------------------------------------------------- epilogue math
__device__ __forceinline__ float gelu_tanh_f(float x) {
    // torch gelu(approximate="tanh"): 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))), with tanh(u) = 1 - 2/(2^(2u*log2 e) + 1)
    // on v_exp_f32 / v_rcp_f32 (abs error ~1e-7, far below the bf16 rounding that follows; libm's tanhf is ~40 VALU ops)
    const float kBeta = 0.7978845608028654f;  // sqrt(2/pi)
    const float kKappa = 0.044715f;
    float x3 = x * x * x;
    float inner = kBeta * (x + kKappa * x3);
    float e = __builtin_amdgcn_exp2f(fminf(inner * 2.8853900817779268f, 126.0f));   // 2^(2u log2 e), clamped: no inf/inf
    float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
    return 0.5f * x * (1.0f + th);
}
// x * sigmoid(x) on v_exp_f32 / v_rcp_f32 (relative error ~2e-7 before the bf16 rounding)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(-x * 1.4426950408889634f, 126.0f)));
}

struct EpiCtx {
    const bf16_t* bias;
    const bf16_t* residual;
    int64_t ldr;
    void* out;
    int64_t ldo;
    int N;       // logical N of the GEMM (2I for swiglu)
    int flags;
};

// Greedy argmax as an epilogue of the lm_head GEMM (bagel.py:1295-1301: argmax over the bf16 logits): every 16-column tile
// leaves one 64-bit key per row - (order-preserving image of the bf16 logit) << 32 | (0xFFFFFFFF - column) - so that the
// maximum key is the largest logit and, among equal logits, the LOWEST column (torch.argmax's tie rule; NaN ranks highest
// like torch).  umv_decode_step_end_argmax takes the maximum over the tiles.
__device__ __forceinline__ uint64_t argmax_key(float v, int n) {
    uint32_t b = __float_as_uint(v == 0.f ? 0.f : v);            // -0 == +0
    uint32_t k = (v != v) ? 0xFFFFFFFFu : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    return ((uint64_t)k << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)n);
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl_xor((int)lo, mask, 64);
    hi = (uint32_t)__shfl_xor((int)hi, mask, 64);
    return ((uint64_t)hi << 32) | lo;
}
// The lane holds columns n0..n0+3 of row m of tile `tile` (lanes l, l^16, l^32, l^48 share the row): reduce the tile's 16
// columns and let the first lane group write partial[m][tile].  `final` are the values epi_store4 stored (bf16-exact).
// Must be called by ALL lanes of the wave (shuffles); lanes without a valid element pass valid = false.
__device__ __forceinline__ void epi_argmax_tile(uint64_t* __restrict__ partial, int64_t ld_partial, int m, int tile, int lane, bool valid,
                                                int n0, int nend, const float* final) {
    uint64_t key = 0;
    if (valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < nend) {
                const uint64_t kj = argmax_key(final[j], n0 + j);
                key = kj > key ? kj : key;
            }
    }
    uint64_t o = shfl_xor_u64(key, 16);
    key = o > key ? o : key;
    o = shfl_xor_u64(key, 32);
    key = o > key ? o : key;
    if (valid && (lane >> 4) == 0) partial[(int64_t)m * ld_partial + tile] = key;
}

// Finish 4 consecutive n (n0..n0+3) of row `orow` from fp32 accumulators.  `final` (optional) receives the stored values.
__device__ __forceinline__ void epi_store4(const EpiCtx& e, int64_t orow, int n0, float v0, float v1, float v2, float v3,
                                           float* final = nullptr) {
    float v[4] = {v0, v1, v2, v3};
    if (e.flags & UMV_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < e.N) v[j] += bf2f(e.bias[n0 + j]);
    }
    if (e.flags & UMV_EPI_OUT_F32) {
        float* o = reinterpret_cast<float*>(e.out) + orow * e.ldo + n0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < e.N) o[j] = v[j];
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = rbf(v[j]);
    if (e.flags & UMV_EPI_GELU_TANH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf(gelu_tanh_f(v[j]));
    }
    if (e.flags & UMV_EPI_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = rbf(silu_f(v[j]));
    }
    if (e.flags & UMV_EPI_RESIDUAL) {
        const bf16_t* rr = e.residual + orow * e.ldr + n0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < e.N) v[j] = rbf(v[j] + bf2f(rr[j]));
    }
    if (final) {
#pragma unroll
        for (int j = 0; j < 4; ++j) final[j] = v[j];     // already bf16-exact (every branch above ends in rbf)
    }
    bf16_t* o = reinterpret_cast<bf16_t*>(e.out) + orow * e.ldo + n0;
    if (n0 + 3 < e.N && (((e.ldo | n0) & 3) == 0)) {   // 8-byte aligned: one packed store
        u32x2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<u32x2*>(o) = pk;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < e.N) o[j] = f2bf(v[j]);
    }
}

// SwiGLU: g,u accumulators of the same 4 output columns -> act[orow][c0..c0+3]
__device__ __forceinline__ void epi_swiglu4(const EpiCtx& e, int64_t orow, int c0, int I, const float* g, const float* u) {
    bf16_t* o = reinterpret_cast<bf16_t*>(e.out) + orow * e.ldo + c0;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float gg = rbf(g[j]), uu = rbf(u[j]);
        v[j] = rbf(rbf(silu_f(gg)) * uu);   // act_fn(gate) -> bf16, * up -> bf16 (modeling_qwen2.py:235)
    }
    if (c0 + 3 < I && (((e.ldo | c0) & 3) == 0)) {
        u32x2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<u32x2*>(o) = pk;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < I) o[j] = f2bf(v[j]);
    }
}

