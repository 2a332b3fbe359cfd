// Image-head kernels: CFG combine + renorm + Euler step; VAE (FLUX autoencoder) kernels:
// NHWC implicit-GEMM convolutions on MFMA, GroupNorm(+swish), latent (un)patchify,
// sampling and pixel conversion.
#include "common.h"
#include <stdlib.h>
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

// ----------------------------------------------------------------------------- timestep sinusoid
// TimestepEmbedder.timestep_embedding (modeling_utils.py:87-109): args = t[:, None] * freqs[None] in fp32,
// emb = cat(cos(args), sin(args)) cast to bf16 by autocast in front of mlp[0].  `freqs` = exp(-ln(10000) * i / half) comes from
// the caller (computed once with torch, so t * freqs has the reference's bits); cos / sin are the full-range libm versions.
__global__ void timestep_embed_kernel(const float* __restrict__ t, const float* __restrict__ freqs, bf16_t* __restrict__ out, int n, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * half) return;
    const int row = i / half, j = i - row * half;
    const float a = t[row] * freqs[j];
    float sn, cs;
    sincosf(a, &sn, &cs);
    out[(int64_t)row * 2 * half + j] = f2bf(cs);
    out[(int64_t)row * 2 * half + half + j] = f2bf(sn);
}
extern "C" int umv_timestep_embed(const float* t, const float* freqs, uint16_t* out, int n, int half, umv_stream_t stream) {
    UMV_CHECK(t && freqs && out && n >= 0 && half > 0, UMV_ERR_ARG, "timestep_embed: bad args");
    if (n == 0) return UMV_OK;
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((n * half + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, freqs, out, n, half);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
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

// ----------------------------------------------------------------------------- VAE: implicit-GEMM convolution
// NHWC bf16 activations.  out[b,oy,ox,co] = bias[co] + sum_{ky,kx,ci} in[b,iy,ix,ci] * W[co][ky][kx][ci]
// is the GEMM  out[m = pixel][n = co] = sum_k x[m][k] Wp[n][k]  with k = (ky*ks+kx)*Cin + ci,
// so it reuses the packed-weight MFMA tile of gemm.hip: W is the A operand streamed from the
// packed image, the im2col row fragment (8 consecutive ci of one tap = 16 contiguous bytes,
// Cin % 8 == 0) is gathered straight into the LDS B-fragment image - no im2col buffer.
// mode 0: stride 1, pad (ks-1)/2 (autoencoder.py:76,78,138,167,214,238)
// mode 1: nearest 2x upsample fused into the gather (Upsample, autoencoder.py:116-118)
// mode 2: stride 2 after F.pad(0,1,0,1), no other padding (Downsample, autoencoder.py:104-107)
struct ConvGeom {
    int B, Cin, Hin, Win, Cout, Hout, Wout, ks, mode;
};

__device__ __attribute__((aligned(16))) const uint32_t g_zero_page_v[4] = {0, 0, 0, 0};
typedef __attribute__((address_space(3))) void* lds_ptr_v;

// Same pipeline as gemm_tiled_kernel (gemm.hip): WN x WM waves of TN x TM MFMA tiles, k-step KTS*32, NBUF LDS
// buffers filled by LDS-DMA with counted vmcnt and one raw barrier per step.  W tiles are straight 1 KiB copies
// of the packed image; an x tile is the im2col fragment gathered per lane (tap / channel decode per k, zero page
// for padding, upsampled or strided source coordinates by mode).
template <int WN, int WM, int TN, int TM, int KTS, int NBUF>
__global__ __launch_bounds__(WN * WM * 64) void conv_tiled_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                                  const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                                  bf16_t* __restrict__ out, ConvGeom geo, int KT, int NTT, int mblocks) {
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int WTILES = BN / 16 * KTS, XTILES = BM / 16 * KTS;
    constexpr int TPW = (WTILES + XTILES) / NW;
    static_assert((WTILES + XTILES) % NW == 0, "staging tiles must divide evenly over the waves");
    constexpr int BUF = (WTILES + XTILES) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const int mblk = blockIdx.x % mblocks;
    const int nblk = blockIdx.x / mblocks;
    const int m0 = mblk * BM;
    const int nt_blk = nblk * (BN / 16);
    const int nt_base = nt_blk + wn * TN;
    const int M = geo.B * geo.Hout * geo.Wout;
    const int K = geo.ks * geo.ks * geo.Cin;
    const int nsteps = (KT + KTS - 1) / KTS;

    // per staged tile: either a W tile (pointer) or an x tile (this lane's output pixel)
    const bf16_t* wsrc[TPW];
    bool wvalid[TPW];
    int pb[TPW], poy[TPW], pox[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int f = wave * TPW + i;
        wsrc[i] = nullptr; wvalid[i] = false; pb[i] = poy[i] = pox[i] = 0;
        if (f < WTILES) {
            const int tl = f / KTS, kk = f % KTS;
            const int nt = nt_blk + tl;
            wvalid[i] = nt < NTT;
            wsrc[i] = wp + ((int64_t)(wvalid[i] ? nt : 0) * KT + kk) * 512 + lane * 8;
        } else {
            const int tl = (f - WTILES) / KTS;
            int m = m0 + tl * 16 + r;
            m = m < M ? m : M - 1;
            pb[i] = m / (geo.Hout * geo.Wout);
            const int rem = m % (geo.Hout * geo.Wout);
            poy[i] = rem / geo.Wout;
            pox[i] = rem % geo.Wout;
        }
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_v);
    const bool cin32 = (geo.Cin & 31) == 0;
    auto stage = [&](int step, int buf) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int f = wave * TPW + i;
            const bf16_t* p = zero;
            if (f < WTILES) {
                const int kt = step * KTS + f % KTS;
                if (wvalid[i] && kt < KT) p = wsrc[i] + (int64_t)step * (KTS * 512);
            } else {
                const int kt = step * KTS + (f - WTILES) % KTS;
                const int k = kt * 32 + g * 8;
                if (k < K) {
                    // Cin % 32 == 0 (every conv of the VAE but conv_in): a 32-wide k-tile lies inside one filter tap, so the tap and
                    // its (ky, kx) are wave-uniform - scalar divisions instead of ~60 VALU instructions per staged piece
                    int tap, ci;
                    if (cin32) {
                        const int tu = __builtin_amdgcn_readfirstlane((kt * 32) / geo.Cin);
                        tap = tu;
                        ci = k - tu * geo.Cin;
                    } else {
                        tap = k / geo.Cin;
                        ci = k - tap * geo.Cin;
                    }
                    const int ky = tap / geo.ks, kx = tap - ky * geo.ks;
                    int iy, ix;
                    bool ok;
                    if (geo.mode == 0) {
                        const int pad = (geo.ks - 1) >> 1;
                        iy = poy[i] + ky - pad; ix = pox[i] + kx - pad;
                        ok = iy >= 0 && iy < geo.Hin && ix >= 0 && ix < geo.Win;
                    } else if (geo.mode == 1) {
                        iy = poy[i] + ky - 1; ix = pox[i] + kx - 1;        // coordinates in the 2x upsampled image
                        ok = iy >= 0 && iy < 2 * geo.Hin && ix >= 0 && ix < 2 * geo.Win;
                        iy >>= 1; ix >>= 1;
                    } else {
                        iy = 2 * poy[i] + ky; ix = 2 * pox[i] + kx;        // zero pad on the bottom / right only
                        ok = iy < geo.Hin && ix < geo.Win;
                    }
                    if (ok) p = x + (((int64_t)pb[i] * geo.Hin + iy) * geo.Win + ix) * geo.Cin + ci;
                }
            }
            char* dst = smem + buf * BUF + f * 1024;
            __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr_v)dst, 16, 0, 0);
        }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
        if (p < nsteps) stage(p, p);
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step % NBUF;
        const int ahead = min(NBUF - 2, nsteps - 1 - step);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TPW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UMV_BARRIER();
        if (step + NBUF - 1 < nsteps) stage(step + NBUF - 1, (step + NBUF - 1) % NBUF);
        const char* wb = smem + cur * BUF;
        const char* xb = wb + WTILES * 1024;
#pragma unroll
        for (int kk = 0; kk < KTS; ++kk) {
            bf16x8 wf[TN], xf[TM];
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(wb + ((wn * TN + t) * KTS + kk) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + ((wm * TM + j) * KTS + kk) * 1024 + lane * 16);
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[t][j] = mfma16(wf[t], xf[j], acc[t][j]);
        }
    }
    // epilogue: + bias -> bf16 ; (+ residual -> bf16); 4 consecutive output channels per lane (bias once per channel group,
    // bias / residual as 8-byte loads when the four channels exist: Cout % 4 == 0 makes every such address 8-byte aligned)
    const bool vec4 = (geo.Cout & 3) == 0 && ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(out)) & 7) == 0;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int n0 = (nt_base + t) * 16 + g * 4;
        if (n0 >= geo.Cout) continue;
        const bool full = vec4 && n0 + 3 < geo.Cout;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            if (full) {
                const u32x2 pk = *reinterpret_cast<const u32x2*>(bias + n0);
                b4[0] = __uint_as_float(pk.x << 16); b4[1] = __uint_as_float(pk.x & 0xFFFF0000u);
                b4[2] = __uint_as_float(pk.y << 16); b4[3] = __uint_as_float(pk.y & 0xFFFF0000u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) b4[q] = bf2f(bias[min(n0 + q, geo.Cout - 1)]);
            }
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + (wm * TM + j) * 16 + r;
            if (m >= M) continue;
            const float v[4] = {acc[t][j].x, acc[t][j].y, acc[t][j].z, acc[t][j].w};
            float r4[4] = {0.f, 0.f, 0.f, 0.f};
            if (residual) {
                const bf16_t* rp = residual + (int64_t)m * geo.Cout + n0;
                if (full) {
                    const u32x2 pk = *reinterpret_cast<const u32x2*>(rp);
                    r4[0] = __uint_as_float(pk.x << 16); r4[1] = __uint_as_float(pk.x & 0xFFFF0000u);
                    r4[2] = __uint_as_float(pk.y << 16); r4[3] = __uint_as_float(pk.y & 0xFFFF0000u);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r4[q] = bf2f(residual[(int64_t)m * geo.Cout + min(n0 + q, geo.Cout - 1)]);
                }
            }
            bf16_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f = rbf(bias ? v[q] + b4[q] : v[q] + 0.f);
                if (residual) f = rbf(f + r4[q]);
                o[q] = f2bf(f);
            }
            bf16_t* dst = out + (int64_t)m * geo.Cout + n0;
            if (full) {
                u32x2 pk;
                pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
                pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                *reinterpret_cast<u32x2*>(dst) = pk;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n0 + q < geo.Cout) dst[q] = o[q];
            }
        }
    }
}

// ----------------------------------------------------------------------------- VAE: 3x3 convolution, input-stationary
// conv_tiled_kernel gathers the im2col fragment of every filter tap from global memory: each input pixel goes through the
// texture-address path NINE times, 16 bytes per lane from a different pixel row each (16 half cache lines per wave instruction).
// At 128-512 channels that path, not the matrix pipe, sets the pace: 768 address cycles against 544 MFMA cycles per k-step on
// a CU, 190-260 TFLOP/s on the decoder's 3x3 convolutions (profiles/r03_vae4_kernel_stats_by_grid.csv).
// Here the INPUT stays put: a workgroup owns a 16 x 16 output tile and 128 output channels; for every 64-channel slice of the
// input it brings the 18 x 18 pixel patch (halo included) into LDS ONCE - 40.5 KiB, pixel-major, 128 bytes per pixel with the
// 16-byte channel octets XOR-swizzled by the pixel index - and all nine taps read their MFMA B fragments from that patch at
// shifted pixel positions (16 consecutive pixels of a patch row per 16-lane row: conflict-free ds_read_b128).  Only the weights
// stream per tap (16 KiB = 8 n-tiles x 2 k-tiles, straight copies of the packed image, 3-deep ring).  Per (slice, tap) step a
// wave issues exactly three LDS-DMA pieces (two weight tiles + one piece of the NEXT slice's patch or a dummy), so the counted
// vmcnt waits see a uniform queue; one raw barrier per step.  Same MFMA operands in the same k order (tap-major, channel-minor
// inside a 64-channel slice... the k order is (slice, tap, channel) instead of (tap, channel): another fp32 summation order,
// same bf16 rounding points (bias, residual) as conv_tiled_kernel.
// MODE 1 = the nearest-2x upsample of Upsample.forward (autoencoder.py:116-118) fused in: the 16 x 16 OUTPUT tile reads a 10 x 10
// patch of the half-resolution input (output pixel (oy, ox), tap (dy, dx) -> input ((oy + dy - 1) >> 1, (ox + dx - 1) >> 1); lanes
// that share an input pixel read the same LDS address: a broadcast, not a conflict).
#define CP_TW 16
#define CP_TH 16
#define CP_WBUF 16384                                // 8 n-tiles x 2 k-tiles
template <int MODE> struct CpGeo {
    static constexpr int PW = MODE == 1 ? CP_TW / 2 + 2 : CP_TW + 2;
    static constexpr int PH = MODE == 1 ? CP_TH / 2 + 2 : CP_TH + 2;
    static constexpr int PIX = PW * PH;                                      // 324 / 100 patch pixels
    static constexpr int PATCH_BYTES = (PIX * 128 + 1023) / 1024 * 1024;     // 41 / 13 DMA pieces of 1 KiB
    static constexpr int PIECES = PATCH_BYTES / 1024;
    static constexpr int LDS = 2 * PATCH_BYTES + 3 * CP_WBUF + 1024;
};

template <int MODE>
__global__ __launch_bounds__(512) void conv3x3_patch_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                            bf16_t* __restrict__ out, int B, int H, int W, int Cin, int Cout, int KT,
                                                            int NTT, int tiles_x, int tiles_y) {
    // H, W: OUTPUT size (MODE 1: the input is H/2 x W/2)
    constexpr int CP_PW = CpGeo<MODE>::PW, CP_PIX = CpGeo<MODE>::PIX, CP_PATCH_BYTES = CpGeo<MODE>::PATCH_BYTES, CP_PATCH_PIECES = CpGeo<MODE>::PIECES;
    const int Hi = MODE == 1 ? H / 2 : H, Wi = MODE == 1 ? W / 2 : W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch = smem;                               // 2 buffers
    char* wbuf = smem + 2 * CP_PATCH_BYTES;           // 3 buffers
    char* dummy = wbuf + 3 * CP_WBUF;                 // 1 KiB sink of the padding pieces
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave & 1, wm = wave >> 1;          // 2 (n) x 4 (m) waves of 64 channels x 64 pixels (4 tile rows)
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid % B;
    const int nblk = bid / B;
    const int ox0 = tx * CP_TW, oy0 = ty * CP_TH;
    const int nt_blk = nblk * 8;
    const int nslices = Cin / 64;
    const int nsteps = nslices * 9;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_v);
    const bf16_t* xb = x + (int64_t)b * Hi * Wi * Cin;
    const int iy0 = MODE == 1 ? oy0 / 2 - 1 : oy0 - 1, ix0 = MODE == 1 ? ox0 / 2 - 1 : ox0 - 1;     // input coordinates of patch pixel (0, 0)

    // ---- LDS-DMA pieces of a step: W tiles f = wave*2 + {0,1} (n-tile f/2.. : tile index t = f >> 1?  8 n-tiles x 2 k-tiles = 16 pieces)
    auto stage_w = [&](int step, int buf) {
        const int slice = step / 9, tap = step - slice * 9;
        const int kt0 = (tap * Cin + slice * 64) >> 5;            // k-tile of the packed image: k = tap*Cin + ci
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = wave * 2 + i;                           // piece = (n-tile f >> 1, k-tile f & 1)
            const int nt = nt_blk + (f >> 1);
            const bf16_t* p = nt < NTT ? wp + ((int64_t)nt * KT + kt0 + (f & 1)) * 512 + lane * 8 : zero;
            __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr_v)(wbuf + buf * CP_WBUF + f * 1024), 16, 0, 0);
        }
    };
    // piece q (0..40) of the patch of `slice`: 64 consecutive 16-byte slots s = q*64 + lane = pixel*8 + pos; the slot holds channel
    // octet pos ^ (pixel & 7) of that pixel (the swizzle is applied on the SOURCE side: the LDS image of a DMA is lane-linear)
    auto stage_patch_piece = [&](int slice, int q, int buf) {
        const int sidx = q * 64 + lane;
        const int pix = sidx >> 3, pos = sidx & 7;
        const int py = pix / CP_PW, px = pix - py * CP_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const int oct = pos ^ (pix & 7);
        const bool ok = pix < CP_PIX && iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        const bf16_t* p = ok ? xb + ((int64_t)iy * Wi + ix) * Cin + slice * 64 + oct * 8 : zero;
        __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr_v)(patch + buf * CP_PATCH_BYTES + q * 1024), 16, 0, 0);
    };
    // the third piece of step (slice, tap): a piece of the NEXT slice's patch during taps 0..5 - into the buffer the PREVIOUS
    // slice used, which every wave has left by then - else a dummy
    auto stage_third = [&](int slice, int tap) {
        const int q = tap * 8 + wave;                              // 6 taps x 8 waves = 48 >= 41 (13) pieces
        if (tap < 6 && q < CP_PATCH_PIECES && slice + 1 < nslices) stage_patch_piece(slice + 1, q, (slice + 1) & 1);
        else __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_v)dummy, 16, 0, 0);
    };

    // ---- prologue: patch of slice 0 (all 41 pieces, 5-6 per wave), then W(0), W(1) with a dummy behind each: the queue of a
    //      wave then looks like the steady state (per step: two weight pieces for step + 2, then one third piece)
    for (int q = wave; q < CP_PATCH_PIECES; q += 8) stage_patch_piece(0, q, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage_w(0, 0);
    __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_v)dummy, 16, 0, 0);
    stage_w(1, 1);
    __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_v)dummy, 16, 0, 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int step = 0; step < nsteps; ++step) {
        // W(step) was issued two steps ago; behind it in the queue: that step's third piece, W(step + 1) and its third piece = 4
        // pieces that may still be in flight (the patch of a slice is issued during taps 0..5 of the slice before: long landed)
        if (step + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UMV_BARRIER();
        const int slice = step / 9, tap = step - slice * 9;
        if (step + 2 < nsteps) stage_w(step + 2, (step + 2) % 3);
        else {      // (keep three pieces per step to the end)
            __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_v)dummy, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_v)dummy, 16, 0, 0);
        }
        stage_third(slice, tap);
        const int dy = tap / 3, dx = tap - dy * 3;
        const char* wb = wbuf + (step % 3) * CP_WBUF;
        const char* pb = patch + (slice & 1) * CP_PATCH_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[4], xf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(wb + (((wn * 4 + t) * 2 + kk) * 1024) + lane * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // patch pixel of output (tile row wm*4+j, column r) at this tap
                const int pix = MODE == 1 ? ((wm * 4 + j + dy + 1) >> 1) * CP_PW + ((r + dx + 1) >> 1) : (wm * 4 + j + dy) * CP_PW + r + dx;
                xf[j] = *reinterpret_cast<const bf16x8*>(pb + pix * 128 + (((kk * 4 + g) ^ (pix & 7)) << 4));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = mfma16(wf[t], xf[j], acc[t][j]);
        }
    }
    // ---- epilogue: + bias -> bf16 ; (+ residual -> bf16); lane (r, g) of tile (t, j) holds pixel column r, channels n0..n0+3
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n0 = (nt_blk + wn * 4 + t) * 16 + g * 4;
        if (n0 >= Cout) continue;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const u32x2 pk = *reinterpret_cast<const u32x2*>(bias + n0);
            b4[0] = __uint_as_float(pk.x << 16); b4[1] = __uint_as_float(pk.x & 0xFFFF0000u);
            b4[2] = __uint_as_float(pk.y << 16); b4[3] = __uint_as_float(pk.y & 0xFFFF0000u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = oy0 + wm * 4 + j, ox = ox0 + r;
            if (oy >= H || ox >= W) continue;
            const int64_t m = ((int64_t)b * H + oy) * W + ox;
            const float v[4] = {acc[t][j].x, acc[t][j].y, acc[t][j].z, acc[t][j].w};
            float r4[4] = {0.f, 0.f, 0.f, 0.f};
            if (residual) {
                const u32x2 pk = *reinterpret_cast<const u32x2*>(residual + m * Cout + n0);
                r4[0] = __uint_as_float(pk.x << 16); r4[1] = __uint_as_float(pk.x & 0xFFFF0000u);
                r4[2] = __uint_as_float(pk.y << 16); r4[3] = __uint_as_float(pk.y & 0xFFFF0000u);
            }
            float f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f[q] = rbf(bias ? v[q] + b4[q] : v[q] + 0.f);
                if (residual) f[q] = rbf(f[q] + r4[q]);
            }
            u32x2 pk;
            pk.x = pack2bf(f[0], f[1]);
            pk.y = pack2bf(f[2], f[3]);
            *reinterpret_cast<u32x2*>(out + m * Cout + n0) = pk;
        }
    }
}

template <int MODE>
static int launch_conv_patch(const bf16_t* x, const bf16_t* wp, const bf16_t* bias, const bf16_t* residual, bf16_t* out, int B, int H, int W,
                             int Cin, int Cout, int KT, int NTT, hipStream_t s) {
    constexpr int CP_LDS = CpGeo<MODE>::LDS;
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CP_LDS);
        UMV_CHECK(e == hipSuccess, UMV_ERR_LAUNCH, "conv3x3_patch: cannot reserve %d bytes of LDS", (int)CP_LDS);
    }
    const int tiles_x = (W + CP_TW - 1) / CP_TW, tiles_y = (H + CP_TH - 1) / CP_TH, nblocks = (Cout + 127) / 128;
    hipLaunchKernelGGL(conv3x3_patch_kernel<MODE>, dim3(tiles_x * tiles_y * B * nblocks), dim3(512), CP_LDS, s, x, wp, bias, residual, out, B, H, W, Cin,
                       Cout, KT, NTT, tiles_x, tiles_y);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

template <int WN, int WM, int TN, int TM, int KTS, int NBUF>
static int launch_conv(const bf16_t* x, const bf16_t* wp, const bf16_t* bias, const bf16_t* residual, bf16_t* out, const ConvGeom& geo,
                       int KT, int NTT, hipStream_t s) {
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr size_t lds = (size_t)NBUF * (BN / 16 * KTS + BM / 16 * KTS) * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tiled_kernel<WN, WM, TN, TM, KTS, NBUF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int M = geo.B * geo.Hout * geo.Wout;
    const int mblocks = (M + BM - 1) / BM, nblocks = (geo.Cout + BN - 1) / BN;
    hipLaunchKernelGGL((conv_tiled_kernel<WN, WM, TN, TM, KTS, NBUF>), dim3(mblocks * nblocks), dim3(WN * WM * 64), lds, s, x, wp, bias,
                       residual, out, geo, KT, NTT, mblocks);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

extern "C" int umv_conv2d_nhwc_bf16(const uint16_t* x, const uint16_t* wp, const uint16_t* bias, const uint16_t* residual,
                                    uint16_t* out, int B, int Cin, int Hin, int Win, int Cout, int ksize, int mode,
                                    umv_stream_t stream) {
    UMV_CHECK(x && wp && out, UMV_ERR_ARG, "conv2d: null pointer");
    UMV_CHECK(Cin % 8 == 0, UMV_ERR_ARG, "conv2d: Cin (%d) must be a multiple of 8 (pad the input channels)", Cin);
    const int force = mode & ~3;       // tests / A-B: | 16 = the input-stationary 3x3 kernel whatever the grid, | 32 = the gather kernel
    mode &= 3;
    UMV_CHECK((ksize == 3 || ksize == 1) && mode >= 0 && mode <= 2 && (force == 0 || force == 16 || force == 32), UMV_ERR_ARG,
              "conv2d: ksize %d mode %d", ksize, mode | force);
    ConvGeom geo;
    geo.B = B; geo.Cin = Cin; geo.Hin = Hin; geo.Win = Win; geo.Cout = Cout; geo.ks = ksize; geo.mode = mode;
    if (mode == 0) { geo.Hout = Hin; geo.Wout = Win; }
    else if (mode == 1) { geo.Hout = 2 * Hin; geo.Wout = 2 * Win; }
    else { geo.Hout = (Hin + 1 - 3) / 2 + 1; geo.Wout = (Win + 1 - 3) / 2 + 1; }
    const int M = B * geo.Hout * geo.Wout;
    if (M == 0) return UMV_OK;
    const int K = ksize * ksize * Cin;
    const int KT = (K + 31) / 32, NTT = (Cout + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    // 3x3, stride 1, input channels in whole 64-channel slices, 4-channel-aligned outputs, and enough 16 x 16 tiles to fill the
    // chip: the input-stationary kernel
    const bool patch_ok = ksize == 3 && (mode == 0 || mode == 1) && Cin % 64 == 0 && Cout % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(out)) & 7) == 0;
    UMV_CHECK(force != 16 || patch_ok, UMV_ERR_UNSUPPORTED, "conv2d: the input-stationary kernel needs 3x3 / stride 1 / Cin %% 64 == 0 / Cout %% 4 == 0");
    // (whatever the grid: even 16 workgroups of it beat the gather kernel on a 32 x 32 x 512 level, and a choice that depended
    // on the batch would break "a batch == its images one by one, bit for bit")
    if (patch_ok && force != 32)
        return mode == 1 ? launch_conv_patch<1>(x, wp, bias, residual, out, B, geo.Hout, geo.Wout, Cin, Cout, KT, NTT, s)
                         : launch_conv_patch<0>(x, wp, bias, residual, out, B, geo.Hout, geo.Wout, Cin, Cout, KT, NTT, s);
    if (Cout <= 16) return launch_conv<1, 4, 1, 4, 4, 2>(x, wp, bias, residual, out, geo, KT, NTT, s);     // conv_out: 16(n) x 256(m)
    const long wg128 = (long)((M + 127) / 128) * ((Cout + 127) / 128);
    if (wg128 >= 384) return launch_conv<2, 2, 4, 4, 2, 2>(x, wp, bias, residual, out, geo, KT, NTT, s);  // 128 x 128 x 64
    return launch_conv<2, 2, 4, 2, 2, 3>(x, wp, bias, residual, out, geo, KT, NTT, s);                    // 128(n) x 64(m) x 64
}

// ----------------------------------------------------------------------------- VAE: GroupNorm(32) (+ swish), NHWC
// F.group_norm on bf16 (autoencoder.py:75,77,166,237,43) then swish x*sigmoid(x) (:34-35): fp32
// statistics over (H*W, C/32) per (sample, group), one bf16 rounding of the normalised value,
// sigmoid rounded to bf16, product rounded to bf16.  Deterministic two-level reduction:
// pass 1 writes per-chunk partial sums, pass 2 folds them in a fixed order and applies.
#define GN_CHUNK 256   // pixels per partial-sum workgroup
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                         int nchunks) {
    // grid (nchunks, B); thread t owns channel octets t, t+256, ... ; cpg = C/32 channels per group
    extern __shared__ float sm[];   // [2][C]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * GN_CHUNK, p1 = min(HW, p0 + GN_CHUNK);
    const int nv = C / 8;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
    __syncthreads();
    // each thread walks (pixel, octet) pairs with a fixed assignment -> deterministic
    const int per_row = nv;
    const int lanes = blockDim.x;
    // thread handles octet (tid % per_row) when per_row <= lanes, striding pixels by lanes/per_row
    if (per_row <= lanes) {
        const int oct = threadIdx.x % per_row, prow = threadIdx.x / per_row, pstride = lanes / per_row;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (prow < pstride) {
            for (int p = p0 + prow; p < p1; p += pstride) {
                bf16x8 v = ldg_frag(x + ((int64_t)b * HW + p) * C + oct * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = bf2f((bf16_t)v[j]); s[j] += f; q[j] += f * f; }
            }
        }
        // fold pixel-rows in a fixed order through LDS
        for (int pr = 0; pr < pstride; ++pr) {
            if (prow == pr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { sm[oct * 8 + j] += s[j]; sm[C + oct * 8 + j] += q[j]; }
            }
            __syncthreads();
        }
    } else {
        for (int oct = threadIdx.x; oct < per_row; oct += lanes) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int p = p0; p < p1; ++p) {
                bf16x8 v = ldg_frag(x + ((int64_t)b * HW + p) * C + oct * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = bf2f((bf16_t)v[j]); s[j] += f; q[j] += f * f; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { sm[oct * 8 + j] = s[j]; sm[C + oct * 8 + j] = q[j]; }
        }
        __syncthreads();
    }
    // per-group sums for this chunk
    const int cpg = C / 32;
    if (threadIdx.x < 32) {
        float s = 0.f, q = 0.f;
        for (int c = 0; c < cpg; ++c) { s += sm[threadIdx.x * cpg + c]; q += sm[C + threadIdx.x * cpg + c]; }
        float* dst = part + (((int64_t)b * nchunks + chunk) * 32 + threadIdx.x) * 2;
        dst[0] = s; dst[1] = q;
    }
}

// Fold the per-chunk partial sums into (mean, rstd) per (sample, group), ONCE per call: 8 lanes per group take contiguous chunk
// ranges (all loads of a lane independent, summed in chunk order), the 8 range sums are added in lane order - a fixed order, so
// the statistics do not depend on the launch geometry.  (Every workgroup of gn_apply used to repeat a serial walk over all
// chunks - 256 dependent L2 round trips at 256 x 256 - before touching a pixel: 193 us for a 134 MB pass.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int nchunks,
                                                          float n_per_group, float eps) {
    __shared__ float ps[32][8], pq[32][8];
    const int b = blockIdx.x;
    const int grp = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int per = (nchunks + 7) / 8;
    const int c0 = sub * per, c1 = min(nchunks, c0 + per);
    float s = 0.f, q = 0.f;
    int ch = c0;
    for (; ch + 4 <= c1; ch += 4) {
        const float* p = part + (((int64_t)b * nchunks + ch) * 32 + grp) * 2;
        const float a0 = p[0], b0 = p[1], a1 = p[64], b1 = p[65], a2 = p[128], b2 = p[129], a3 = p[192], b3 = p[193];
        s += a0; q += b0; s += a1; q += b1; s += a2; q += b2; s += a3; q += b3;
    }
    for (; ch < c1; ++ch) {
        const float* p = part + (((int64_t)b * nchunks + ch) * 32 + grp) * 2;
        s += p[0]; q += p[1];
    }
    ps[grp][sub] = s; pq[grp][sub] = q;
    __syncthreads();
    if (sub == 0) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { S += ps[grp][i]; Q += pq[grp][i]; }
        const float mean = S / n_per_group;
        const float var = fmaxf(Q / n_per_group - mean * mean, 0.f);
        stats[((int64_t)b * 32 + grp) * 2] = mean;
        stats[((int64_t)b * 32 + grp) * 2 + 1] = rsqrt_ieee(var + eps);
    }
}

// y = bf16((x - mean) * rstd * gamma + beta), then swish with the reference's roundings (sigmoid -> bf16, product -> bf16).  A thread
// keeps ONE channel octet for its whole grid-stride walk (the stride is a multiple of C / 8), so its gamma / beta / statistics
// live in registers; sigmoid on v_exp_f32 / v_rcp_f32.
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ out, int HW, int C, int swish) {
    const int b = blockIdx.y;
    const int cpg = C / 32;
    const int nv = C / 8;
    const int64_t total = (int64_t)HW * nv;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;        // a multiple of nv (host side)
    const int oct = (int)(i0 % nv);
    const bf16x8 gm = ldg_frag(gamma + oct * 8), bt = ldg_frag(beta + oct * 8);
    float mean[8], rstd[8], gw[8], bw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (oct * 8 + j) / cpg;
        mean[j] = stats[((int64_t)b * 32 + grp) * 2];
        rstd[j] = stats[((int64_t)b * 32 + grp) * 2 + 1];
        gw[j] = bf2f((bf16_t)gm[j]);
        bw[j] = bf2f((bf16_t)bt[j]);
    }
    const bf16_t* xb = x + (int64_t)b * HW * C;
    bf16_t* ob = out + (int64_t)b * HW * C;
    for (int64_t i = i0; i < total; i += stride) {
        const int64_t off = i * 8;                                   // (pixel * nv + oct) * 8
        const bf16x8 v = ldg_frag(xb + off);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = rbf((bf2f((bf16_t)v[j]) - mean[j]) * rstd[j] * gw[j] + bw[j]);
            if (swish) {
                // torch.sigmoid on the bf16 tensor, then the bf16 product.  v_exp_f32 / v_rcp_f32 give the same bf16 sigmoid as torch for
                // EVERY bf16 y > -87.5 (tests/test_vae_gpu.py::test_groupnorm_swish_every_bf16_value walks all 65280 finite values);
                // below that e^-y leaves the range the hardware units keep (denormal results flush), so those few values take IEEE
                // division and libm's expf - y -> -0 once the sigmoid underflows, as in the reference.
                float sg = rbf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fminf(-y * 1.4426950408889634f, 126.0f))));
                if (y < -87.0f) sg = rbf(__fdiv_rn(1.0f, 1.0f + expf(-y)));
                y = rbf(y * sg);
            }
            o[j] = (short)f2bf(y);
        }
        *reinterpret_cast<bf16x8*>(ob + off) = o;
    }
}

extern "C" size_t umv_groupnorm_workspace_bytes(int B, int HW) {
    return (size_t)B * ((HW + GN_CHUNK - 1) / GN_CHUNK) * 32 * 2 * sizeof(float) + (size_t)B * 32 * 2 * sizeof(float);   // partials + (mean, rstd)
}

extern "C" int umv_groupnorm_nhwc_bf16(const uint16_t* x, const uint16_t* gamma, const uint16_t* beta, uint16_t* out, void* workspace,
                                       int B, int HW, int C, float eps, int swish, umv_stream_t stream) {
    UMV_CHECK(x && gamma && beta && out && workspace, UMV_ERR_ARG, "groupnorm: null pointer");
    UMV_CHECK(C % 32 == 0 && C % 8 == 0, UMV_ERR_ARG, "groupnorm: C=%d must be a multiple of 32", C);
    if (B == 0 || HW == 0) return UMV_OK;
    const int nchunks = (HW + GN_CHUNK - 1) / GN_CHUNK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunks, B), dim3(256), 2 * C * sizeof(float), s, x, (float*)workspace, HW, C, nchunks);
    UMV_LAUNCH_CHECK();
    float* stats = (float*)workspace + (size_t)B * nchunks * 32 * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, (const float*)workspace, stats, nchunks, (float)HW * (float)(C / 32), eps);
    UMV_LAUNCH_CHECK();
    const int nv = C / 8;                                            // 256 * blocks is a multiple of nv for every C = 32 * 2^k <= 2048;
    int blocks = (int)min((int64_t)2048, ((int64_t)HW * nv + 255) / 256);   // other widths: round the grid up to a multiple of nv
    if ((256 * blocks) % nv != 0) blocks = ((blocks + nv - 1) / nv) * nv;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, s, x, (const float*)stats, gamma, beta, out, HW, C, swish);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- VAE: layout / boundary kernels
// image [B,3,H,W] fp32 NCHW -> NHWC bf16 with channels zero padded to Cp (autocast's cast before conv_in)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int C, int H, int W, int Cp) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)B * H * W * Cp;
    if (gid >= total) return;
    int c = (int)(gid % Cp);
    int64_t p = gid / Cp;
    int xw = (int)(p % W);
    int y = (int)((p / W) % H);
    int b = (int)(p / ((int64_t)W * H));
    out[gid] = c < C ? f2bf(x[(((int64_t)b * C + c) * H + y) * W + xw]) : (bf16_t)0;
}
extern "C" int umv_nchw_f32_to_nhwc_bf16(const float* x, uint16_t* out, int B, int C, int H, int W, int Cp, umv_stream_t stream) {
    UMV_CHECK(x && out && Cp >= C, UMV_ERR_ARG, "nchw_to_nhwc: bad args");
    int64_t total = (int64_t)B * H * W * Cp;
    if (total == 0) return UMV_OK;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, B, C, H, W, Cp);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// latent tokens [h*w, p*p*c] (fp32 x_t or bf16) -> NHWC bf16 [1, h*p, w*p, c] with z/scale + shift
// (inferencer.py:239-241 "nhwpqc->nchpwq" and autoencoder.py:306), bf16 rounding per op.
__global__ void unpatchify_latent_kernel(const float* __restrict__ tok, bf16_t* __restrict__ out, int h, int w, int p, int c,
                                         float scale, float shift) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)h * p * w * p * c;
    if (gid >= total) return;
    int ch = (int)(gid % c);
    int64_t pix = gid / c;
    int X = (int)(pix % (w * p)), Y = (int)(pix / (w * p));
    int hy = Y / p, py = Y % p, wx = X / p, px = X % p;
    float v = rbf(tok[((int64_t)hy * w + wx) * (p * p * c) + (py * p + px) * c + ch]);   // latent.to(bf16)
    v = rbf(rbf(v / scale) + shift);
    out[gid] = f2bf(v);
}
extern "C" int umv_unpatchify_latent(const float* tokens, uint16_t* out, int h, int w, int p, int c, float scale, float shift,
                                     umv_stream_t stream) {
    UMV_CHECK(tokens && out, UMV_ERR_ARG, "unpatchify_latent: null pointer");
    int64_t total = (int64_t)h * p * w * p * c;
    if (total == 0) return UMV_OK;
    hipLaunchKernelGGL(unpatchify_latent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tokens, out,
                       h, w, p, c, scale, shift);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// decoder output NHWC bf16 [H,W,Cs] (first 3 channels) -> uint8 [H,W,3]:
// ((x*0.5+0.5).clamp(0,1))*255 in bf16, truncating cast (inferencer.py:253-254)
__global__ void pixels_u8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ out, int64_t npix, int Cs) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * 3) return;
    int c = (int)(gid % 3);
    int64_t p = gid / 3;
    float v = bf2f(x[p * Cs + c]);
    v = rbf(rbf(v * 0.5f) + 0.5f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    v = rbf(v * 255.0f);
    out[gid] = (uint8_t)v;
}
extern "C" int umv_pixels_to_u8(const uint16_t* x, uint8_t* out, int64_t npix, int Cs, umv_stream_t stream) {
    UMV_CHECK(x && out && Cs >= 3, UMV_ERR_ARG, "pixels_to_u8: bad args");
    if (npix == 0) return UMV_OK;
    hipLaunchKernelGGL(pixels_u8_kernel, dim3((unsigned)((npix * 3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, npix, Cs);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// encoder tail: moments NHWC bf16 [B,Hm,Wm,2z] -> z = mean + exp(0.5*logvar)*noise ; scale*(z - shift)
// (autoencoder.py:266-272,300-303) then 2x2 patchify "chpwq->hwpqc" of the top-left h*p x w*p window
// (bagel.py:771-775) -> tokens bf16 [h*w, p*p*z].  noise is NCHW bf16 [B,z,Hm,Wm] as torch.randn_like draws it.
__global__ void latent_sample_patchify_kernel(const bf16_t* __restrict__ mom, const bf16_t* __restrict__ noise, bf16_t* __restrict__ tok,
                                              int b, int Hm, int Wm, int z, int h, int w, int p, float scale, float shift) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)h * w * p * p * z;
    if (gid >= total) return;
    int ch = (int)(gid % z);
    int64_t r = gid / z;
    int px = (int)(r % p); r /= p;
    int py = (int)(r % p); r /= p;
    int wx = (int)(r % w);
    int hy = (int)(r / w);
    int Y = hy * p + py, X = wx * p + px;
    const bf16_t* m = mom + (((int64_t)b * Hm + Y) * Wm + X) * (2 * z);
    float mean = bf2f(m[ch]), logvar = bf2f(m[z + ch]);
    float stdv = rbf(expf(rbf(0.5f * logvar)));
    float nz = bf2f(noise[(((int64_t)b * z + ch) * Hm + Y) * Wm + X]);
    float zz = rbf(mean + rbf(stdv * nz));
    zz = rbf(scale * rbf(zz - shift));
    tok[gid] = f2bf(zz);
}
extern "C" int umv_latent_sample_patchify(const uint16_t* moments, const uint16_t* noise, uint16_t* tokens, int b, int Hm, int Wm,
                                          int z, int h, int w, int p, float scale, float shift, umv_stream_t stream) {
    UMV_CHECK(moments && noise && tokens, UMV_ERR_ARG, "latent_sample_patchify: null pointer");
    UMV_CHECK(h * p <= Hm && w * p <= Wm, UMV_ERR_ARG, "latent_sample_patchify: window exceeds the latent");
    int64_t total = (int64_t)h * w * p * p * z;
    if (total == 0) return UMV_OK;
    hipLaunchKernelGGL(latent_sample_patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments,
                       noise, tokens, b, Hm, Wm, z, h, w, p, scale, shift);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- single-head attention of the VAE mid block as two GEMMs
// AttnBlock.attention (autoencoder.py:50-62) is ONE head of C = 512 channels over H*W positions: as S = Q K^T and O = P V those are
// ordinary GEMMs at the tiled kernel's rate (3136 x 3136 x 512 at 448 x 448), where the streaming attention kernel at hd 512 keeps
// 128 accumulator registers per lane and runs at 27 TFLOP/s (750 us of a 4.1 ms encode).  Between the GEMMs:
//   umv_softmax_rows_f32:   P[r][c] = bf16(exp((S[r][c] - max_c S[r][c]) * scale)), l[r] = sum_c of the unrounded weights
//   umv_rowscale_f32_bf16:  out[r][c] = bf16(O[r][c] / l[r])
// i.e. the flash-attention arithmetic (fp32 scores and sums, bf16 weights, one division at the end) with the row's true maximum.
template <int MAXI>             // 2 columns per thread and pass: n <= 2 * 256 * MAXI (16: 8192, 32: 16384 = a 1024 x 1024 image's latent)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* S, int64_t lds_, bf16_t* P, int64_t ldp, float* l, int n,
                                                           float scale_log2e) {
    __shared__ float red[8];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* s = S + (int64_t)r * lds_;
    float2 v[MAXI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 2 * (tid + 256 * i);
        v[i] = c < n ? *reinterpret_cast<const float2*>(s + c) : make_float2(-INFINITY, -INFINITY);
        mx = fmaxf(mx, fmaxf(v[i].x, v[i].y));
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float mref = (mx == -INFINITY) ? 0.f : mx;
    float sum = 0.f;
    bf16_t* p = P + (int64_t)r * ldp;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = 2 * (tid + 256 * i);
        if (c < n) {
            const float p0 = umv_exp2((v[i].x - mref) * scale_log2e), p1 = umv_exp2((v[i].y - mref) * scale_log2e);
            sum += p0;
            sum += p1;
            *reinterpret_cast<uint32_t*>(p + c) = pack2bf(p0, p1);
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    if (tid == 0) l[r] = (red[4] + red[5]) + (red[6] + red[7]);
}
extern "C" int umv_softmax_rows_f32(const float* S, int64_t ld_s, uint16_t* P, int64_t ld_p, float* l, int rows, int n, float scale,
                                    umv_stream_t stream) {
    UMV_CHECK(S && P && l, UMV_ERR_ARG, "softmax_rows: null pointer");
    UMV_CHECK(n > 0 && n <= 16384 && (n % 2) == 0 && (ld_s % 2) == 0 && (ld_p % 2) == 0 && ld_s >= n && ld_p >= n, UMV_ERR_UNSUPPORTED,
              "softmax_rows: n (%d) must be even and <= 16384, row strides even and >= n", n);
    if (rows <= 0) return UMV_OK;
    if (n <= 8192)
        hipLaunchKernelGGL(softmax_rows_kernel<16>, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, ld_s, P, ld_p, l, n, scale * 1.4426950408889634f);
    else
        hipLaunchKernelGGL(softmax_rows_kernel<32>, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, ld_s, P, ld_p, l, n, scale * 1.4426950408889634f);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

__global__ __launch_bounds__(256) void rowscale_kernel(const float* O, int64_t ldo_in, const float* l, bf16_t* out, int64_t ldo, int rows, int C) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cpr = C / 2;
    const int64_t r = gid / cpr;
    if (r >= rows) return;
    const int c = (int)(gid - r * cpr) * 2;
    const float lv = l[r];
    const float inv = lv > 0.f ? 1.0f / lv : 0.f;
    const float2 v = *reinterpret_cast<const float2*>(O + r * ldo_in + c);
    *reinterpret_cast<uint32_t*>(out + r * ldo + c) = pack2bf(v.x * inv, v.y * inv);
}
extern "C" int umv_rowscale_f32_bf16(const float* O, int64_t ld_in, const float* l, uint16_t* out, int64_t ld_out, int rows, int C,
                                     umv_stream_t stream) {
    UMV_CHECK(O && l && out, UMV_ERR_ARG, "rowscale: null pointer");
    UMV_CHECK(C > 0 && (C % 2) == 0 && (ld_in % 2) == 0 && (ld_out % 2) == 0, UMV_ERR_ARG, "rowscale: C and the row strides must be even");
    if (rows <= 0) return UMV_OK;
    const int64_t total = (int64_t)rows * (C / 2);
    hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, O, ld_in, l, out, ld_out, rows, C);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
