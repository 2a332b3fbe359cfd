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

// Finish 4 consecutive n (n0..n0+3) of row `orow` from fp32 accumulators.
__device__ __forceinline__ void epi_store4(const EpiCtx& e, int64_t orow, int n0, float v0, float v1, float v2, float v3) {
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

