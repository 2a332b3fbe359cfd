// W8A8 GEMM on the gfx950 block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4, scales fixed at 2^0):
// the MFMA-bound half of the fp8 mode (BASELINE.json configs[4]) - prefill and flow passes, M > 64 rows.
//
//   out[m,n] = epi( sx[m] * sw[n] * sum_k xq[m,k] * wq[n,k] )        xq, wq: OCP e4m3;  sx, sw: powers of two
//
// wq / sw are the per-channel quantised weights of umv_quantize_pack_weight_fp8; xq / sx come from
// umv_quantize_act_fp8 (per ROW: sx[m] = smallest 2^e with 448 * 2^e >= max_k |x[m,k]|).  Products of two e4m3 values are
// exact in fp32, the scales are exact, so the only rounding is the fp32 accumulation: the CPU oracle restates this as a
// bf16 linear on the dequantised operands (oracle/fp8.py).  No reference counterpart exists (the reference is bf16).
//
// Operand layout (checked by tools/fp8_mfma_probe.hip): lane (r = l & 15, g = l >> 4) of the A / B operand holds 32
// consecutive k (g*32 .. g*32+31) of row r, 32 bytes; D as every 16x16 MFMA (col = l & 15, row = g*4 + reg).
//   weight image P8M[n/16][k/128][h][lane][16 B]: plane h holds bytes h*16 .. h*16+15 of every lane's 32 - two 1 KiB planes
//   per (16 x 128) tile, each a straight LDS-DMA copy; x planes are gathered per lane from row-major xq[M][ldq].
// Pipeline: the LDS-DMA / counted-wait / one-barrier-per-k-step scheme of gemm_tiled_kernel with k-steps of 128.
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((address_space(3))) void* lds8_ptr_t;
__device__ __attribute__((aligned(16))) const uint32_t g_zero_page8[4] = {0, 0, 0, 0};

// ----------------------------------------------------------------------------- activations: per-row e4m3
__device__ __forceinline__ float act_pow2_scale(float amax) {   // smallest 2^e with 448 * 2^e >= amax (1 for a zero row)
    if (!(amax > 0.f)) return 1.0f;
    int ea;
    const float ma = frexpf(amax, &ea);
    return ldexpf(1.0f, ma <= 0.875f ? ea - 9 : ea - 8);
}

typedef __attribute__((ext_vector_type(2))) __bf16 act_bf16x2_hw;

// deq (optional): the dequantised row written back as bf16 at the ORIGINAL row position (same row_idx, row stride ldd), for
// the M <= 64 kernels that take bf16 activations
__global__ __launch_bounds__(256) void quantize_act_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, const int32_t* __restrict__ row_idx,
                                                               uint8_t* __restrict__ xq, int64_t ldq, float* __restrict__ xs, int K,
                                                               bf16_t* __restrict__ deq, int64_t ldd) {
    __shared__ float part[4];
    const int m = blockIdx.x;
    const int64_t srow = row_idx ? (int64_t)row_idx[m] : (int64_t)m;
    const bf16_t* xr = x + srow * ldx;
    const int nv = K / 8;
    float mx = 0.f;
    for (int c = threadIdx.x; c < nv; c += 256) {
        const bf16x8 v = ldg_frag(xr + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(bf2f((bf16_t)v[j])));
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    const float s = act_pow2_scale(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
    if (threadIdx.x == 0) xs[m] = s;
    const float inv = 1.0f / s;
    uint8_t* q = xq ? xq + (int64_t)m * ldq : nullptr;
    for (int c = threadIdx.x; c < (int)(ldq / 8); c += 256) {
        int lo = 0, hi = 0;
        if (c < nv) {
            const bf16x8 v = ldg_frag(xr + c * 8);
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = bf2f((bf16_t)v[j]) * inv;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            if (deq) {
                union { act_bf16x2_hw h[4]; bf16x8 v; } d;
                d.h[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, s, false);
                d.h[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, s, true);
                d.h[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, s, false);
                d.h[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, s, true);
                *reinterpret_cast<bf16x8*>(deq + srow * ldd + c * 8) = d.v;
            }
        }
        u32x2 o = {(uint32_t)lo, (uint32_t)hi};
        if (xq) *reinterpret_cast<u32x2*>(q + c * 8) = o;   // zero padded up to ldq
    }
}

extern "C" int umv_quantize_act_fp8(const uint16_t* x, int64_t ldx, const int32_t* row_idx, uint8_t* xq, int64_t ldq, float* x_scale,
                                    uint16_t* deq, int64_t ldd, int M, int K, umv_stream_t stream) {
    UMV_CHECK(x && (xq || deq) && x_scale && M >= 0 && K > 0, UMV_ERR_ARG, "quantize_act_fp8: bad args");
    UMV_CHECK(!deq || (ldd % 8) == 0, UMV_ERR_ARG, "quantize_act_fp8: ldd must be a multiple of 8");
    UMV_CHECK((K % 8) == 0 && (ldx % 8) == 0 && (ldq % 128) == 0 && ldq >= K, UMV_ERR_ARG,
              "quantize_act_fp8: K (%d), ldx must be multiples of 8 and ldq (%lld) a multiple of 128 >= K", K, (long long)ldq);
    if (M == 0) return UMV_OK;
    hipLaunchKernelGGL(quantize_act_fp8_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, ldx, row_idx, xq, ldq, x_scale, K, deq, ldd);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- the GEMM
// inline-asm building blocks of the hand-ordered k-step (free functions: clang rejects asm operands naming the enclosing
// function's locals from inside a generic lambda)
template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void mfma8_asm(f32x4& c, const i32x8& a, const i32x8& b, int one) {   // scales 2^0 on both operands
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(one));
}
template <int N>
__device__ __forceinline__ void lds_wait() {   // at most N of the LDS reads issued so far may still be in flight (in order)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ i32x8 frag8(const u32x4& lo, const u32x4& hi) {
    return (i32x8){(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

// piece p of n_pieces goes behind MFMA floor((2p + 1) * span / (2 * n_pieces)) of the step's first `span` MFMAs
__host__ __device__ constexpr int dma_slot8(int i, int span, int n_pieces) {
    for (int p = 0; p < n_pieces; ++p)
        if (((2 * p + 1) * span) / (2 * n_pieces) == i) return p;
    return -1;
}

template <int WN, int WM, int TN, int TM, int NBUF>
__global__ __launch_bounds__(WN * WM * 64) void gemm_tiled8_kernel(umv_gemm8_args a, int KT, int NTT, int mblocks, int nblocks, int gn) {
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int WPL = 2 * (BN / 16), XPL = 2 * (BM / 16);   // 1 KiB planes per k-step: W then x
    constexpr int TPW = (WPL + XPL) / NW;
    static_assert((WPL + XPL) % NW == 0, "staging planes must divide evenly over the waves");
    constexpr int BUF = (WPL + XPL) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    const int nwg = mblocks * nblocks;
    int bid = blockIdx.x;
    {   // XCD-aware order, as gemm_tiled_kernel
        const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    int mblk, nblk;   // strips of gn n-blocks, n-block within the strip fastest (see gemm_tiled_kernel)
    {
        const int per = mblocks * gn, strip = bid / per, rem = bid - strip * per;
        const int w = min(gn, nblocks - strip * gn);
        mblk = rem / w;
        nblk = strip * gn + rem % w;
    }
    const int m0 = mblk * BM;
    const int nt_blk = nblk * (BN / 16);
    const int nt_base = nt_blk + wn * TN;

    const uint8_t* src[TPW];
    bool tvalid[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int f = wave * TPW + i;
        if (f < WPL) {
            const int nt = nt_blk + (f >> 1), h = f & 1;
            tvalid[i] = nt < NTT;
            src[i] = a.wp + (((int64_t)(tvalid[i] ? nt : 0) * KT) * 2 + h) * 1024 + lane * 16;     // + kt * 2048 per step
        } else {
            const int fx = f - WPL;
            const int m = m0 + (fx >> 1) * 16 + r, h = fx & 1;
            tvalid[i] = m < a.M;
            src[i] = a.xq + (int64_t)(tvalid[i] ? m : 0) * a.ldq + g * 32 + h * 16;                 // + kt * 128 per step
        }
    }
    const uint8_t* zero = reinterpret_cast<const uint8_t*>(g_zero_page8);
    f32x4 acc[TN][TM];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // With >= 3 buffers the LDS-DMA pieces of k-step kt + NBUF - 1 are issued BETWEEN the MFMAs of step kt (as in
    // gemm_tiled_kernel: in a burst behind the barrier the TPW pieces keep both lockstep waves of a SIMD off the matrix pipe
    // for TPW x 100-185 cycles; 2048 x 3584 x 18944: 165 -> 154 us); steps past the end copy the zero page, so that every
    // counted wait sees the same number of pieces in flight.  With two buffers (the 256 x 256 tile) a piece issued inside the
    // step has less than a step to land before the next step's vmcnt(0): measured 2-5 % slower even confined to the first
    // half of the MFMAs, so that tile keeps the burst.
    auto piece = [&](int kt, int buf, int i) {
        const int f = wave * TPW + i;
        const uint8_t* p = (tvalid[i] && kt < KT) ? src[i] + (int64_t)kt * (f < WPL ? 2048 : 128) : zero;
        __builtin_amdgcn_global_load_lds((const void*)p, (lds8_ptr_t)(smem + buf * BUF + f * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
#pragma unroll
        for (int i = 0; i < TPW; ++i) piece(p, p, i);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds8_ptr_t)smem;
    const int one = 0x7f7f7f7f;    // E8M0 scale 2^0 in every byte
    constexpr int NMMA = TN * TM, DMA_SPAN = NMMA;
    constexpr bool DMA_IN = NBUF >= 3;
    static_assert(TPW <= DMA_SPAN, "at most one DMA piece per MFMA");
    for (int kt = 0; kt < KT; ++kt) {
        if (DMA_IN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * TPW) : "memory");   // step kt landed (mine); newer ones fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UMV_BARRIER();
        const int kst = kt + NBUF - 1, bst = kst % NBUF;
        if constexpr (!DMA_IN) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) piece(kst, bst, i);      // (also past the end: the zero page, into the buffer just vacated)
        }
        // Hand-ordered step.  Left to the compiler, every wave issues its 2 (TM + TN) ds_read_b128 right after the barrier
        // and then its TN x TM MFMAs: the 8 waves' 192 KiB of reads (~1500 LDS cycles) and the ~2000 MFMA cycles per SIMD
        // add up (MfmaUtil 47 %).  Here only the x fragments and the first two W fragments are requested up front; the W
        // fragment of n-tile t+2 is requested between the MFMAs of n-tile t (three rotating register sets), and each wait
        // lets the two newest reads stay in flight (LDS returns in order).
        const uint32_t wb = lds0 + (kt % NBUF) * BUF + lane * 16 + wn * TN * 2048;
        const uint32_t xb = lds0 + (kt % NBUF) * BUF + WPL * 1024 + lane * 16 + wm * TM * 2048;
        u32x4 xlo[TM], xhi[TM], wlo[3], whi[3];
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            lds_read16<j * 2048>(xlo[j], xb);
            lds_read16<j * 2048 + 1024>(xhi[j], xb);
        });
        static_for<0, (TN < 2 ? TN : 2)>([&](auto T) {
            constexpr int t = decltype(T)::value;
            lds_read16<t * 2048>(wlo[t], wb);
            lds_read16<t * 2048 + 1024>(whi[t], wb);
        });
        i32x8 xf[TM];
        static_for<0, TN>([&](auto T) {
            constexpr int t = decltype(T)::value;
            if constexpr (t + 1 < TN) lds_wait<2>(); else lds_wait<0>();     // W fragment t (and the x fragments) have landed
            if constexpr (t == 0) {
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[j] = frag8(xlo[j], xhi[j]);
            }
            const i32x8 wf = frag8(wlo[t % 3], whi[t % 3]);
            static_for<0, TM>([&](auto J) {
                constexpr int j = decltype(J)::value;
                mfma8_asm(acc[t][j], wf, xf[j], one);
                if constexpr (j == (TM > 1 ? 1 : 0) && t + 2 < TN) {
                    lds_read16<(t + 2) * 2048>(wlo[(t + 2) % 3], wb);
                    lds_read16<(t + 2) * 2048 + 1024>(whi[(t + 2) % 3], wb);
                }
                constexpr int pc = DMA_IN ? dma_slot8(t * TM + j, DMA_SPAN, TPW) : -1;       // the DMA piece (if any) behind this MFMA
                if constexpr (pc >= 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(kst, bst, pc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the trailing zero-page pieces: the epilogue reuses the buffers
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs are opaque to the hazard recogniser: XDL write -> VALU read
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(acc[t][j]));
    // epilogue: exact power-of-two scales, then the shared bf16 epilogue (bias / activation / SwiGLU / residual)
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    if constexpr (BN * BM * 2 <= NBUF * BUF) {   // whole rows through LDS (gemm_epilogue.h): same values, 16-byte stores
        UMV_BARRIER();             // every wave has read its last fragments: the staging buffers are free
        epi_wave_tile_lds<TN, TM>(e, acc, smem + wave * (TN * TM * 512), lane, m0 + wm * TM * 16, a.M, a.row_idx, nt_base, NTT, nullptr,
                                  a.w_scale, a.x_scale);
        return;
    }
    const bool swiglu = (a.epilogue & UMV_EPI_SWIGLU) != 0;
    static_for<0, TM>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int m = m0 + (wm * TM + j) * 16 + r;
        if (m < a.M) {
            const int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
            const float sx = a.x_scale[m];
            if (swiglu) {
                static_for<0, TN / 2>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    const int ntile = nt_base + 2 * p;
                    if (ntile < NTT) {
                        const int c0 = (ntile >> 1) * 16 + g * 4;
                        const f32x4 sg = *reinterpret_cast<const f32x4*>(a.w_scale + ntile * 16 + g * 4);
                        const f32x4 su = *reinterpret_cast<const f32x4*>(a.w_scale + (ntile + 1) * 16 + g * 4);
                        float gg[4] = {acc[2 * p][j].x * sg.x * sx, acc[2 * p][j].y * sg.y * sx, acc[2 * p][j].z * sg.z * sx, acc[2 * p][j].w * sg.w * sx};
                        float uu[4] = {acc[2 * p + 1][j].x * su.x * sx, acc[2 * p + 1][j].y * su.y * sx, acc[2 * p + 1][j].z * su.z * sx,
                                       acc[2 * p + 1][j].w * su.w * sx};
                        epi_swiglu4(e, orow, c0, a.N / 2, gg, uu);
                    }
                });
            } else {
                static_for<0, TN>([&](auto T) {
                    constexpr int t = decltype(T)::value;
                    const int n0 = (nt_base + t) * 16 + g * 4;
                    if (n0 < a.N) {
                        const f32x4 sw = *reinterpret_cast<const f32x4*>(a.w_scale + n0);
                        epi_store4(e, orow, n0, acc[t][j].x * sw.x * sx, acc[t][j].y * sw.y * sx, acc[t][j].z * sw.z * sx, acc[t][j].w * sw.w * sx);
                    }
                });
            }
        }
    });
}

static int raster8_gn() {   // n-blocks per strip of the tile order; UMV_GEMM_RASTER overrides (tuning only)
    static const int gn = [] { const char* e = getenv("UMV_GEMM_RASTER"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();   // (read once, thread-safe)
    return gn;
}

template <int WN, int WM, int TN, int TM, int NBUF>
static int launch_tiled8(const umv_gemm8_args& a, int KT, int NTT, hipStream_t s) {
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr size_t lds = (size_t)NBUF * 2 * (BN / 16 + BM / 16) * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tiled8_kernel<WN, WM, TN, TM, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    }
    const int mblocks = (a.M + BM - 1) / BM, nblocks = (a.N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_tiled8_kernel<WN, WM, TN, TM, NBUF>), dim3(mblocks * nblocks), dim3(WN * WM * 64), lds, s, a, KT, NTT, mblocks,
                       nblocks, raster8_gn());
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

extern "C" int umv_gemm_fp8a8w(const umv_gemm8_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap != nullptr, UMV_ERR_ARG, "gemm_fp8a8w: null args");
    const umv_gemm8_args a = *ap;
    UMV_CHECK(a.xq && a.x_scale && a.wp && a.w_scale && a.out, UMV_ERR_ARG, "gemm_fp8a8w: null pointer");
    UMV_CHECK(a.M >= 0 && a.N > 0 && a.K > 0 && (a.ldq % 128) == 0 && a.ldq >= a.K, UMV_ERR_ARG,
              "gemm_fp8a8w: bad shape M=%d N=%d K=%d ldq=%lld (ldq: multiple of 128, >= K)", a.M, a.N, a.K, (long long)a.ldq);
    UMV_CHECK(!(a.epilogue & UMV_EPI_BIAS) || a.bias, UMV_ERR_ARG, "gemm_fp8a8w: BIAS without bias pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_RESIDUAL) || a.residual, UMV_ERR_ARG, "gemm_fp8a8w: RESIDUAL without residual pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_SWIGLU) || (a.N % 32) == 0, UMV_ERR_ARG, "gemm_fp8a8w: SWIGLU needs N %% 32 == 0");
    UMV_CHECK(!(a.epilogue & UMV_EPI_OUT_F32), UMV_ERR_UNSUPPORTED, "gemm_fp8a8w: OUT_F32 unsupported");
    if (a.M == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    const int KT = (a.K + 127) / 128, NTT = (a.N + 15) / 16;
    static const int force = [] { const char* e = getenv("UMV_GEMM8_TILE"); return e ? atoi(e) : 0; }();   // tuning only: UMV_GEMM8_TILE=<256|258|259|128> (read once, thread-safe)
    // Measured (tools/gemm_bench.py --fp8): the 256 x 256 tile wins once it gives >= ~144 workgroups (M=2048,N=4608: 69.6 vs
    // 77.4 us); below that the 256 x 128 tile with three stage buffers does (M=1024,N=3584,K=18944: 159 vs 251 us), and its
    // third buffer is worth 5-10% over two on every shape.
    const long wg256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    int cfg = wg256 >= 144 ? 256 : 259;
    if (force) cfg = force;
    if (cfg == 256) return launch_tiled8<2, 4, 8, 4, 2>(a, KT, NTT, s);   // 256 x 256 x 128, 2 buffers of 64 KiB
    if (cfg == 128) return launch_tiled8<2, 2, 4, 4, 2>(a, KT, NTT, s);   // 128 x 128 x 128, 2 buffers of 32 KiB
    if (cfg == 258) return launch_tiled8<4, 2, 4, 4, 2>(a, KT, NTT, s);   // 256(n) x 128(m) x 128, 2 buffers of 48 KiB
    return launch_tiled8<4, 2, 4, 4, 3>(a, KT, NTT, s);                   // 256(n) x 128(m) x 128, 3 buffers (144 KiB)
}
