// Image-head kernels: CFG combine + renorm + Euler step; VAE (FLUX autoencoder) kernels:
// NHWC implicit-GEMM convolutions on MFMA, GroupNorm(+swish), latent (un)patchify,
// sampling and pixel conversion.
#include "common.h"
#include "../../include/unimedvl_hip.h"

// ----------------------------------------------------------------------------- CFG + renorm + Euler
// bagel.py:1173-1207 and :983, with every intermediate rounded to bf16 where the reference
// holds a bf16 tensor (v_t and friends are bf16; x_t is fp32):
//   v_text_ = v_c + s_t*(v_t - v_c) ; v_ = v_i + s_i*(v_text_ - v_i)
//   scale = clamp(norm(v_t)/(norm(v_)+1e-8), min, 1) ; v = v_*scale ; x_t -= v*dt
// One workgroup per sample ("global" norms are per sample; the reference is batch-1 here).
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    return t;
}

__device__ __forceinline__ float cfg_mix(float v, float vc, float s) {
    // vc + s*(v - vc) with bf16 rounding after each op
    return rbf(vc + rbf(s * rbf(v - vc)));
}

__global__ __launch_bounds__(1024) void cfg_renorm_euler_kernel(float* __restrict__ x_t, const bf16_t* __restrict__ v_t,
                                                                const bf16_t* __restrict__ v_text, const bf16_t* __restrict__ v_img,
                                                                int64_t ldv, const int32_t* __restrict__ rows,
                                                                const int32_t* __restrict__ seg_off, float s_text, float s_img,
                                                                float renorm_min, int rtype, float dt, int D) {
    __shared__ float sm[16];
    const int s = blockIdx.x;
    const int n0 = seg_off[s], n1 = seg_off[s + 1];
    const int total = (n1 - n0) * D;
    const bool use_text = s_text > 1.0f, use_img = s_img > 1.0f;
    const float rmin = rbf(renorm_min);
    if (!use_text) {  // no guidance: v = v_t
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int n = n0 + i / D, d = i % D;
            float v = bf2f(v_t[(int64_t)rows[n] * ldv + d]);
            x_t[(int64_t)n * D + d] -= rbf(v * dt);
        }
        return;
    }
    if (rtype == 0) {  // global: one scale per sample
        float a0 = 0.f, a1 = 0.f;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int n = n0 + i / D, d = i % D;
            int64_t off = (int64_t)rows[n] * ldv + d;
            float v = bf2f(v_t[off]);
            float vm = cfg_mix(v, bf2f(v_text[off]), s_text);
            if (use_img) vm = cfg_mix(vm, bf2f(v_img[off]), s_img);
            a0 += v * v;
            a1 += vm * vm;
        }
        a0 = block_sum(a0, sm);
        a1 = block_sum(a1, sm);
        const float nv = rbf(sqrtf(a0)), nm = rbf(sqrtf(a1));
        float scale = rbf(nv / rbf(nm + 1e-8f));
        scale = fminf(fmaxf(scale, rmin), 1.0f);
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int n = n0 + i / D, d = i % D;
            int64_t off = (int64_t)rows[n] * ldv + d;
            float v = bf2f(v_t[off]);
            float vm = cfg_mix(v, bf2f(v_text[off]), s_text);
            if (use_img) vm = cfg_mix(vm, bf2f(v_img[off]), s_img);
            x_t[(int64_t)n * D + d] -= rbf(rbf(vm * scale) * dt);
        }
        return;
    }
    // per-token norms: one wave per token (D <= 64*? handled by looping)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int n = n0 + wave; n < n1; n += nwaves) {
        const int64_t base = (int64_t)rows[n] * ldv;
        float a0 = 0.f, a1 = 0.f;
        for (int d = lane; d < D; d += 64) {
            float v = bf2f(v_t[base + d]);
            float vm = cfg_mix(v, bf2f(v_text[base + d]), s_text);
            if (rtype == 1 && use_img) vm = cfg_mix(vm, bf2f(v_img[base + d]), s_img);
            a0 += v * v;
            a1 += vm * vm;
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        const float nv = rbf(sqrtf(a0)), nm = rbf(sqrtf(a1));
        float scale = rbf(nv / rbf(nm + 1e-8f));
        scale = fminf(fmaxf(scale, rmin), 1.0f);
        for (int d = lane; d < D; d += 64) {
            float v = bf2f(v_t[base + d]);
            float vm = cfg_mix(v, bf2f(v_text[base + d]), s_text);
            float out;
            if (rtype == 1) {  // channel
                if (use_img) vm = cfg_mix(vm, bf2f(v_img[base + d]), s_img);
                out = rbf(vm * scale);
            } else {           // text_channel: renorm the text-guided velocity, then image guidance
                out = rbf(vm * scale);
                if (use_img) out = cfg_mix(out, bf2f(v_img[base + d]), s_img);
            }
            x_t[(int64_t)n * D + d] -= rbf(out * dt);
        }
    }
}

extern "C" int umv_cfg_renorm_euler(float* x_t, const uint16_t* v_t, const uint16_t* v_text, const uint16_t* v_img, int64_t ldv,
                                    const int32_t* rows, const int32_t* seg_off, int nseg, float cfg_text_scale,
                                    float cfg_img_scale, float renorm_min, int renorm_type, float dt, int D,
                                    umv_stream_t stream) {
    UMV_CHECK(x_t && v_t && rows && seg_off, UMV_ERR_ARG, "cfg_renorm_euler: null pointer");
    UMV_CHECK(renorm_type >= 0 && renorm_type <= 2, UMV_ERR_ARG, "cfg_renorm_euler: renorm_type %d", renorm_type);
    UMV_CHECK(!(cfg_text_scale > 1.0f) || v_text, UMV_ERR_ARG, "cfg_renorm_euler: cfg_text_scale>1 without v_text");
    UMV_CHECK(!(cfg_text_scale > 1.0f && cfg_img_scale > 1.0f) || v_img, UMV_ERR_ARG, "cfg_renorm_euler: cfg_img_scale>1 without v_img");
    if (nseg == 0) return UMV_OK;
    hipLaunchKernelGGL(cfg_renorm_euler_kernel, dim3(nseg), dim3(1024), 0, (hipStream_t)stream, x_t, v_t, v_text, v_img, ldv,
                       rows, seg_off, cfg_text_scale, cfg_img_scale, renorm_min, renorm_type, dt, D);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
