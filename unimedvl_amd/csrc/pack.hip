// Weight images of libunimedvl_hip (gfx950): everything that turns a checkpoint tensor into what the GEMM kernels stream.
//
//   bf16   P[n/16][k/32][lane = g*16 + r][8]: element j of lane (r, g) is W[nt*16 + r][kt*32 + g*8 + j] - MFMA A-fragment order, one
//          wavefront instruction fetches a 16(n) x 32(k) tile as 1 KiB contiguous (umv_pack_weight_bf16; SwiGLU: gate / up tiles
//          interleaved, umv_pack_weight_swiglu_bf16); Q[n/th][k/32][g][r < th][8] = the same with th-row tiles for the decode GEMM
//   e4m3   P8[n/16][k/64][lane][16 B] + one power-of-two fp32 scale per output channel (umv_quantize_pack_weight_fp8), and the
//          K = 128 image of the scaled fp8 MFMA made from it (umv_repack_weight_fp8_mfma)
//
// This file (with PACK_LAYOUT_VERSION) is what unimedvl_amd/packstore.py stamps its on-disk cache of packed images with: an edit here
// invalidates the cache, an edit of a GEMM / attention / vision kernel does not.
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include "gemm_internal.h"
#include <stdlib.h>

// ----------------------------------------------------------------------------- bf16 images
__global__ void pack_weight_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ p, int N, int K, int NTT,
                                   int KT, int interleave_I, const bf16_t* __restrict__ w2) {
    // one thread per 8-element group of the packed image
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)NTT * KT * 64;
    if (gid >= total) return;
    int lane = (int)(gid & 63);
    int64_t tile = gid >> 6;
    int kt = (int)(tile % KT);
    int nt = (int)(tile / KT);
    int r = lane & 15, g = lane >> 4;
    int k = kt * 32 + g * 8;
    const bf16_t* src = w;
    int n;
    if (interleave_I > 0) {  // swiglu: even tiles gate, odd tiles up
        int t = nt >> 1;
        n = t * 16 + r;
        src = (nt & 1) ? w2 : w;
        if (n >= interleave_I) n = -1;
    } else {
        n = nt * 16 + r;
        if (n >= N) n = -1;
    }
    bf16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n >= 0 && k + j < K) ? src[(int64_t)n * K + k + j] : (bf16_t)0;
    u32x4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16);
    o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16);
    o.w = v[6] | ((uint32_t)v[7] << 16);
    *reinterpret_cast<u32x4*>(p + gid * 8) = o;
}

// Re-tile a standard packed image (16-row tiles) into `th`-row tiles for the decode GEMM:
//   Q[n/th][k/32][g][r < th][k%8]   (th <= 16; th == 16 is the standard image)
// With th = N / 256 (e.g. 14 rows for N = 3584) the skinny GEMM gets exactly one tile per CU.
__global__ void repack_rows_kernel(const bf16_t* __restrict__ p16, bf16_t* __restrict__ q, int N, int KT, int th, int64_t total) {
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-element group per thread
    if (gid >= total) return;
    const int r = (int)(gid % th);
    const int g = (int)((gid / th) % 4);
    const int kt = (int)((gid / (4 * th)) % KT);
    const int64_t nt = gid / ((int64_t)4 * th * KT);
    const int64_t n = nt * th + r;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < N) v = *reinterpret_cast<const u32x4*>(p16 + (((n >> 4) * KT + kt) * 64 + g * 16 + (n & 15)) * 8);
    *reinterpret_cast<u32x4*>(q + gid * 8) = v;
}

extern "C" size_t umv_repacked_weight_elems(int N, int K, int th) {
    size_t nt = ((size_t)N + th - 1) / th, kt = (size_t)(K + 31) / 32;
    return nt * kt * 4 * th * 8;
}

extern "C" int umv_repack_weight_rows_bf16(const uint16_t* packed16, uint16_t* out, int N, int K, int th, umv_stream_t stream) {
    UMV_CHECK(packed16 && out && N > 0 && K > 0 && th >= 1 && th <= 16, UMV_ERR_ARG, "repack_weight_rows: bad args (th=%d)", th);
    const int KT = (K + 31) / 32;
    const int64_t total = (int64_t)((N + th - 1) / th) * KT * 4 * th;
    hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed16, out, N,
                       KT, th, total);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

extern "C" size_t umv_packed_weight_elems(int N, int K) {
    size_t ntt = (size_t)(N + 15) / 16, kt = (size_t)(K + 31) / 32;
    return ntt * kt * 512;
}

extern "C" int umv_pack_weight_bf16(const uint16_t* w, uint16_t* packed, int N, int K, umv_stream_t stream) {
    UMV_CHECK(w && packed && N > 0 && K > 0, UMV_ERR_ARG, "pack_weight: bad args");
    int NTT = (N + 15) / 16, KT = (K + 31) / 32;
    int64_t total = (int64_t)NTT * KT * 64;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, N, K, NTT, KT, 0,
                       (const bf16_t*)nullptr);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

extern "C" int umv_pack_weight_swiglu_bf16(const uint16_t* gate, const uint16_t* up, uint16_t* packed, int I, int K,
                                           umv_stream_t stream) {
    UMV_CHECK(gate && up && packed && I > 0 && K > 0, UMV_ERR_ARG, "pack_weight_swiglu: bad args");
    int NTT = 2 * ((I + 15) / 16), KT = (K + 31) / 32;
    int64_t total = (int64_t)NTT * KT * 64;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, gate, packed, 2 * I, K, NTT,
                       KT, I, up);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- e4m3 images
// smallest power of two s with 448 * s >= amax (448 = 0.875 * 2^9 is the largest finite e4m3 value)
__device__ __forceinline__ float fp8_pow2_scale(float amax) {
    if (!(amax > 0.f)) return 1.0f;
    int ea;
    float ma = frexpf(amax, &ea);   // amax = ma * 2^ea, ma in [0.5, 1)
    return ldexpf(1.0f, ma <= 0.875f ? ea - 9 : ea - 8);
}

// One workgroup (256 threads) per packed 16-row tile: row maxima -> scales -> e4m3 image (+ optional W' in bf16).
__global__ __launch_bounds__(256) void quantize_pack_fp8_kernel(const bf16_t* __restrict__ w, const bf16_t* __restrict__ w2,
                                                                uint8_t* __restrict__ p8, float* __restrict__ scale,
                                                                bf16_t* __restrict__ deq, bf16_t* __restrict__ deq2, int rows,
                                                                int K, int KT8) {
    __shared__ float smax[16][17];
    __shared__ float sscale[16];
    const int nt = blockIdx.x, tid = threadIdx.x;
    const bool inter = w2 != nullptr;
    const bf16_t* src = (inter && (nt & 1)) ? w2 : w;
    bf16_t* dq = (inter && (nt & 1)) ? deq2 : deq;
    const int row0 = (inter ? (nt >> 1) : nt) * 16;
    {   // 16 threads per row
        const int r = tid >> 4, c = tid & 15;
        float m = 0.f;
        if (row0 + r < rows)
            for (int k = c; k < K; k += 16) m = fmaxf(m, fabsf(bf2f(src[(int64_t)(row0 + r) * K + k])));
        smax[r][c] = m;
    }
    __syncthreads();
    if (tid < 16) {
        float m = 0.f;
        for (int c = 0; c < 16; ++c) m = fmaxf(m, smax[tid][c]);
        const float s = fp8_pow2_scale(m);
        sscale[tid] = s;
        scale[nt * 16 + tid] = s;
    }
    __syncthreads();
    // one thread per (kt8, lane) 16-byte group
    for (int idx = tid; idx < KT8 * 64; idx += 256) {
        const int lane = idx & 63, kt8 = idx >> 6;
        const int r = lane & 15, g = lane >> 4;
        const bool rowok = row0 + r < rows;
        const float inv = 1.0f / sscale[r];   // exact: a power of two
        uint32_t o[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k0 = kt8 * 64 + h * 32 + g * 8;
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (rowok && k0 + j < K) ? bf2f(src[(int64_t)(row0 + r) * K + k0 + j]) * inv : 0.f;
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            o[2 * h] = (uint32_t)lo;
            o[2 * h + 1] = (uint32_t)hi;
            if (dq && rowok) {
                u32x4 qq = {(uint32_t)lo, (uint32_t)hi, 0u, 0u};
                bf16x8 d, unused;
                cvt_fp8x16(qq, sscale[r], d, unused);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j < K) dq[(int64_t)(row0 + r) * K + k0 + j] = (bf16_t)d[j];
            }
        }
        u32x4 v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<u32x4*>(p8 + ((int64_t)nt * KT8 * 64 + idx) * 16) = v;
    }
}

extern "C" size_t umv_packed_weight_fp8_bytes(int N, int K) {
    return ((size_t)(N + 15) / 16) * ((size_t)(K + 63) / 64) * 1024;
}

extern "C" int umv_quantize_pack_weight_fp8(const uint16_t* w, const uint16_t* w_up, uint8_t* packed8, float* scale,
                                            uint16_t* deq, uint16_t* deq_up, int rows, int K, umv_stream_t stream) {
    UMV_CHECK(w && packed8 && scale && rows > 0 && K > 0, UMV_ERR_ARG, "quantize_pack_weight_fp8: bad args");
    UMV_CHECK(!w_up || (rows % 16) == 0, UMV_ERR_ARG, "quantize_pack_weight_fp8: SwiGLU image needs I %% 16 == 0 (I=%d)", rows);
    UMV_CHECK(!(deq_up && !w_up), UMV_ERR_ARG, "quantize_pack_weight_fp8: deq_up without w_up");
    const int ntt = (w_up ? 2 : 1) * ((rows + 15) / 16), KT8 = (K + 63) / 64;
    hipLaunchKernelGGL(quantize_pack_fp8_kernel, dim3(ntt), dim3(256), 0, (hipStream_t)stream, w, w_up, packed8, scale, deq, deq_up,
                       rows, K, KT8);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}


// ----------------------------------------------------------------------------- weight image for the MFMA
// from the decode image P8[nt][k/64][lane = g*16 + r][16 B: (k%64)/32 * 8 + k%8]; one thread per 8-byte piece
__global__ void repack_fp8_mfma_kernel(const uint8_t* __restrict__ p8, uint8_t* __restrict__ out, int KT8, int KT128, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // ((nt*KT128 + kt)*2 + h)*64 + lane)*2 + half
    if (gid >= total) return;
    const int half = (int)(gid & 1);
    const int lane = (int)((gid >> 1) & 63);
    const int h = (int)((gid >> 7) & 1);
    const int64_t tile = gid >> 8;
    const int kt = (int)(tile % KT128);
    const int64_t nt = tile / KT128;
    const int r = lane & 15, g = lane >> 4;
    const int k0 = kt * 128 + g * 32 + h * 16 + half * 8;
    const int kt8 = k0 >> 6, rem = k0 & 63;
    u32x2 v = {0u, 0u};
    if (kt8 < KT8) v = *reinterpret_cast<const u32x2*>(p8 + ((nt * KT8 + kt8) * 64 + ((rem & 31) >> 3) * 16 + r) * 16 + (rem >> 5) * 8);
    *reinterpret_cast<u32x2*>(out + gid * 8) = v;
}

extern "C" size_t umv_packed_weight_fp8_mfma_bytes(int N, int K) {
    return ((size_t)(N + 15) / 16) * ((size_t)(K + 127) / 128) * 2048;
}

extern "C" int umv_repack_weight_fp8_mfma(const uint8_t* packed8, uint8_t* out, int N, int K, umv_stream_t stream) {
    UMV_CHECK(packed8 && out && N > 0 && K > 0, UMV_ERR_ARG, "repack_weight_fp8_mfma: bad args");
    const int KT8 = (K + 63) / 64, KT128 = (K + 127) / 128;
    const int64_t total = (int64_t)((N + 15) / 16) * KT128 * 2 * 64 * 2;
    hipLaunchKernelGGL(repack_fp8_mfma_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed8, out,
                       KT8, KT128, total);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

