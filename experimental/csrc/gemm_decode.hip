// Decode GEMM for libunimedvl_hip (gfx950): out[m,n] = epi(sum_k x[m,k] W[n,k]) for M <= 16 rows (one token per
// sample, Bagel.generate_text bagel.py:1262-1314), the HBM-bound weight-streaming case.
//
// One persistent workgroup per CU (grid = G, 8 waves).  The weight is re-tiled once into a "decode image" that gives
// every workgroup ONE contiguous slab: its C = ceil(N/G) output channels as `tpw` tiles of `th` <= 16 rows
// (ragged last tile), all of K:
//     bf16: D[wg][tile][k/32][g][r < th][8]            e4m3: D8[wg][tile][k/64][g][r < th][16]
// (SwiGLU: a "tile" is a (gate, up) pair of th-row tiles).  Because the workgroup owns many tiles,
//   * x (and, fused, Qwen2RMSNorm(x) * w, modeling_qwen2.py:89-94) is fetched / normalised ONCE per workgroup and
//     each wave keeps the B-fragments of its K slice in registers: the per-tile x re-reads of the one-tile-per-
//     workgroup kernel (gemm.hip) were costing as much L2->L1 traffic as the weights themselves (measured: e4m3
//     down_proj 18.2 us with x loads vs 12.8 us without, tools/skinny_bench.py);
//   * the weight stream never stops at tile boundaries: the next tile's first chunks are already in flight while
//     the 8 K-slices are reduced through LDS and the epilogue runs.
// Work split and summation order are those of gemm_skinny_kernel (8 contiguous K slices in wave order), so for the
// same x the results are bit-identical to umv_gemm_bf16 / umv_gemm_fp8w.
#include "common.h"
#include "unimedvl_hip_experimental.h"
#include "gemm_epilogue.h"

#define DG_WAVES 8
#define DG_XT 16   // k-tiles of x a wave keeps in registers (XREG path: K <= 8 * 16 * 32 = 4096)

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;

__device__ __forceinline__ void dg_cvt_fp8x16(u32x4 q, float scale, bf16x8& lo, bf16x8& hi) {
    union { bf16x2_hw h[4]; bf16x8 v; } a, b;
    a.h[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, scale, false);
    a.h[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, scale, true);
    a.h[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, scale, false);
    a.h[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, scale, true);
    b.h[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.z, scale, false);
    b.h[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.z, scale, true);
    b.h[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.w, scale, false);
    b.h[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.w, scale, true);
    lo = a.v;
    hi = b.v;
}

// ----------------------------------------------------------------------------- layout
extern "C" int umv_decode_layout_for(int rows, int G, umv_decode_layout* out) {
    UMV_CHECK(out && rows > 0 && G > 0, UMV_ERR_ARG, "decode_layout_for: bad args");
    const int C = (rows + G - 1) / G;
    int best_th = 16, best_cost = 1 << 30;
    for (int th = 16; th >= 8; --th) {
        const int tpw = (C + th - 1) / th;
        const int cost = 2 * tpw * th + tpw;   // rows streamed (padding included) + half a row per tile of fixed work
        if (cost < best_cost) { best_cost = cost; best_th = th; }
    }
    out->G = G;
    out->C = C;
    out->th = best_th;
    out->tpw = (C + best_th - 1) / best_th;
    return UMV_OK;
}

extern "C" size_t umv_decode_image_bytes(int K, int swiglu, int fp8, const umv_decode_layout* L) {
    if (!L) return 0;
    const size_t ktu = fp8 ? (size_t)(K + 63) / 64 : (size_t)(K + 31) / 32;
    return (size_t)L->G * L->tpw * (swiglu ? 2 : 1) * ktu * 4 * L->th * 16;
}

// One thread per 16-byte group of the decode image; the source is the standard 16-row image (bf16: P[n/16][k/32][g*16+r][8],
// e4m3: P8[n/16][k/64][g*16+r][16]; SwiGLU sources interleave gate / up tiles).  Works on 16-byte groups in both cases.
__global__ void repack_decode_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, const float* __restrict__ sscale,
                                     float* __restrict__ dscale, int rows, int KTU, int swiglu, umv_decode_layout L, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int th = L.th, parts = swiglu ? 2 : 1;
    const int r = (int)(gid % th);
    const int g = (int)((gid / th) % 4);
    const int kt = (int)((gid / (4 * th)) % KTU);
    const int64_t t = gid / ((int64_t)4 * th * KTU);   // ((wg * tpw + tile) * parts + part)
    const int part = (int)(t % parts);
    const int tile = (int)((t / parts) % L.tpw);
    const int wg = (int)(t / ((int64_t)parts * L.tpw));
    const int c = tile * th + r;                        // channel within the workgroup's slab
    const int64_t n = (int64_t)wg * L.C + c;
    const bool ok = c < L.C && n < rows;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) {
        const int64_t st = swiglu ? ((n >> 4) * 2 + part) : (n >> 4);
        v = src[(st * KTU + kt) * 64 + g * 16 + (n & 15)];
        if (dscale && kt == 0 && g == 0) dscale[t * 16 + r] = sscale[st * 16 + (n & 15)];
    } else if (dscale && kt == 0 && g == 0) {
        dscale[t * 16 + r] = 1.0f;
    }
    dst[gid] = v;
}

extern "C" int umv_repack_weight_decode(const void* packed16, const float* scale16, void* out, float* scale_out, int rows, int K,
                                        int swiglu, int fp8, const umv_decode_layout* L, umv_stream_t stream) {
    UMV_CHECK(packed16 && out && L && rows > 0 && K > 0, UMV_ERR_ARG, "repack_weight_decode: bad args");
    UMV_CHECK(L->th >= 1 && L->th <= 16 && L->tpw >= 1 && L->G >= 1 && (int64_t)L->G * L->C >= rows && L->tpw * L->th >= L->C,
              UMV_ERR_ARG, "repack_weight_decode: layout G=%d C=%d th=%d tpw=%d does not cover %d rows", L->G, L->C, L->th, L->tpw, rows);
    UMV_CHECK(!fp8 || (scale16 && scale_out), UMV_ERR_ARG, "repack_weight_decode: the e4m3 image needs its scales");
    const int KTU = fp8 ? (K + 63) / 64 : (K + 31) / 32;
    const int64_t total = (int64_t)L->G * L->tpw * (swiglu ? 2 : 1) * KTU * 4 * L->th;
    hipLaunchKernelGGL(repack_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const u32x4*)packed16, (u32x4*)out, fp8 ? scale16 : nullptr, fp8 ? scale_out : nullptr, rows, KTU, swiglu, *L, total);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// ----------------------------------------------------------------------------- kernel
// W8: e4m3 image.  NORM: 0 = x as is, 8 / 16 = fused RMSNorm keeping that many rows.  XREG: x fragments of the wave's
// K slice live in registers (K <= 4096); otherwise they are loaded with every chunk (down_proj, K = 18944).
// U = 16-byte weight loads per lane, per tile part, per chunk (bf16: U k-tiles; e4m3: 2U k-tiles).
template <bool W8, int NORM, bool XREG, bool SWIGLU, int U>
__global__ __launch_bounds__(DG_WAVES * 64) void gemm_decode_kernel(umv_gemm_args a, umv_decode_layout L, int KT) {
    constexpr int NP = SWIGLU ? 2 : 1;       // tile parts per step (gate, up)
    constexpr int KPU = W8 ? 2 : 1;          // k-tiles per 16-byte weight unit
    extern __shared__ __attribute__((aligned(16))) float red[];   // 2 x [DG_WAVES][NP][64] f32x4, then norm scratch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: K-slice offsets stay in SGPRs
    const int r = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x;
    const int th = L.th, tpw = L.tpw;
    const int KTU = (KT + KPU - 1) / KPU;                         // weight units along K
    const int ku_per = (KTU + DG_WAVES - 1) / DG_WAVES;
    const int ku_begin = wave * ku_per;
    const int ku_end = min(KTU, ku_begin + ku_per);
    const int nku = max(0, ku_end - ku_begin);
    const int nchunks = (nku + U - 1) / U;
    const bool rowlane = r < th;
    const int64_t tile_bytes = (int64_t)KTU * 4 * th * 16;        // one tile part
    const char* wslab = reinterpret_cast<const char*>(a.wp) + (int64_t)wg * tpw * NP * tile_bytes + (int64_t)(g * th + (rowlane ? r : 0)) * 16;

    struct Buf {
        u32x4 w[U][NP];
        bf16x8 x[XREG ? 1 : U * KPU];
    };
    const bool xvalid = r < a.M;
    const bf16_t* xrow = a.x + (xvalid ? (a.row_idx ? (int64_t)a.row_idx[r] : (int64_t)r) : 0) * a.ldx;

    auto load_chunk = [&](int tile, int c, Buf& b) {
        const char* wt = wslab + (int64_t)tile * NP * tile_bytes;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ku = ku_begin + c * U + u;
            const bool ok = ku < ku_end && rowlane;
#pragma unroll
            for (int p = 0; p < NP; ++p)
                b.w[u][p] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wt + p * tile_bytes + (int64_t)ku * 4 * th * 16))
                               : (u32x4){0u, 0u, 0u, 0u};
            if constexpr (!XREG) {
#pragma unroll
                for (int h = 0; h < KPU; ++h) {
                    const int k = (ku * KPU + h) * 32 + g * 8;
                    b.x[u * KPU + h] = (ku < ku_end && xvalid && k < a.K) ? ldg_frag(xrow + k) : zero_frag();
                }
            }
        }
    };

    // chunks 0 and 1 of the slab (chunk 1 may belong to the second tile)
    Buf b0, b1;
    auto first_loads = [&]() {
        if (tpw > 0 && nchunks > 0) load_chunk(0, 0, b0);
        if (nchunks > 1 || XREG) load_chunk(0, 1, b1);                 // XREG: chunk 1 may be a padding chunk (loads nothing)
        else if (tpw > 1 && nchunks > 0) load_chunk(1, 0, b1);
    };

    // ---- x: fetched before the first weight chunk so that it is not queued behind HBM latency (loads return in order)
    bf16x8 xreg[XREG ? DG_XT : 1];
    if constexpr (XREG && NORM == 0) {
#pragma unroll
        for (int i = 0; i < DG_XT; ++i) {
            const int kt = ku_begin * KPU + i;
            const int k = kt * 32 + g * 8;
            xreg[i] = (i < nku * KPU && xvalid && k < a.K) ? ldg_frag(xrow + k) : zero_frag();
        }
        first_loads();
    } else if constexpr (XREG) {
        // Qwen2RMSNorm(x) * w staged once per workgroup through LDS in B-fragment order: slot (kt, g, row) holds the
        // 8 bf16 of `row` at k = kt*32 + g*8.  Every thread owns MP slots; all of its x and norm_w loads are one batch.
        constexpr int MP = NORM;
        constexpr int sh = MP == 8 ? 3 : 4;
        float* part = red + 2 * DG_WAVES * NP * 256;                   // after the two reduction buffers: [DG_WAVES][16]
        bf16_t* xl = reinterpret_cast<bf16_t*>(part + DG_WAVES * 16);  // [KT*4*MP][8]
        const int rr = tid & (MP - 1);
        const int nslots = KT * 4 * MP;
        const bool rowok = rr < a.M;
        const bf16_t* xr = a.x + (rowok ? (a.row_idx ? (int64_t)a.row_idx[rr] : (int64_t)rr) : 0) * a.ldx;
        // norm_w goes through LDS (one 16-byte load per thread instead of MP redundant ones): [KT*4][8] after xl
        bf16_t* wl = xl + (int64_t)nslots * 8;
        bf16x8 xv[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int sidx = tid + i * DG_WAVES * 64;
            const int k = (sidx >> (sh + 2)) * 32 + ((sidx >> sh) & 3) * 8;
            xv[i] = (sidx < nslots && k < a.K && rowok) ? ldg_frag(xr + k) : zero_frag();
        }
        const bf16x8 wn1 = (tid * 8 < a.K) ? ldg_frag(a.norm_w + tid * 8) : zero_frag();
        first_loads();
        *reinterpret_cast<bf16x8*>(wl + tid * 8) = wn1;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = bf2f((bf16_t)xv[i][j]);
                ss += f * f;
            }
        if (MP == 8) ss += __shfl_xor(ss, 8, 64);     // lanes sharing a row: lane & (MP-1)
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        if (lane < MP) part[wave * 16 + lane] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < DG_WAVES; ++w) tot += part[w * 16 + rr];
        const float rstd = rsqrt_ieee(tot / (float)a.K + a.norm_eps);
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            const int sidx = tid + i * DG_WAVES * 64;
            if (sidx < nslots) {
                const int k = (sidx >> (sh + 2)) * 32 + ((sidx >> sh) & 3) * 8;
                const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wl + k);
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)wv[j]) * rbf(bf2f((bf16_t)xv[i][j]) * rstd));   // two roundings
                *reinterpret_cast<bf16x8*>(xl + (int64_t)sidx * 8) = o;
            }
        }
        __syncthreads();
        const int rsel = r & (MP - 1);
#pragma unroll
        for (int i = 0; i < DG_XT; ++i) {
            const int kt = ku_begin * KPU + i;
            xreg[i] = (i < nku * KPU && kt < KT && r < MP) ? *reinterpret_cast<const bf16x8*>(xl + ((int64_t)(kt * 4 + g) * MP + rsel) * 8)
                                                           : zero_frag();
        }
    } else {
        first_loads();
    }

    float wscale[NP];
    f32x4 acc[NP];
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    const int rows_total = SWIGLU ? a.N / 2 : a.N;

    auto begin_tile = [&](int tile) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (W8) wscale[p] = a.w_scale[((int64_t)(wg * tpw + tile) * NP + p) * 16 + r];
        }
    };
    // ci = chunk index within the tile (compile time on the XREG path so that xreg[] is never indexed dynamically)
    auto consume = [&](auto ci, Buf& b) {
        const int c = ci;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (W8) {
                bf16x8 lo[NP], hi[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) dg_cvt_fp8x16(b.w[u][p], wscale[p], lo[p], hi[p]);
                bf16x8 x0, x1;
                if constexpr (XREG) {
                    constexpr int xi = (decltype(ci)::value * U) * 2;
                    x0 = xreg[(xi + 2 * u) % DG_XT];
                    x1 = xreg[(xi + 2 * u + 1) % DG_XT];
                } else {
                    x0 = b.x[u * 2];
                    x1 = b.x[u * 2 + 1];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = mfma16(lo[p], x0, acc[p]);
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = mfma16(hi[p], x1, acc[p]);
            } else {
                bf16x8 x0;
                if constexpr (XREG) {
                    constexpr int xi = decltype(ci)::value * U;
                    x0 = xreg[(xi + u) % DG_XT];
                } else {
                    x0 = b.x[u];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    union { u32x4 q; bf16x8 v; } cv;
                    cv.q = b.w[u][p];
                    acc[p] = mfma16(cv.v, x0, acc[p]);
                }
            }
        }
        (void)c;
    };
    auto finish_tile = [&](int tile) {
        f32x4* rbuf = reinterpret_cast<f32x4*>(red) + (tile & 1) * (DG_WAVES * NP * 64);   // two buffers: one barrier per tile
#pragma unroll
        for (int p = 0; p < NP; ++p) rbuf[(wave * NP + p) * 64 + lane] = acc[p];
        __syncthreads();
        if (tid < 64) {   // wave 0 reduces in wave order and finishes the tile: lane (r = x row m, g): 4 consecutive channels
            f32x4 s[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                s[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < DG_WAVES; ++w) s[p] += rbuf[(w * NP + p) * 64 + lane];
            }
            const int c0 = tile * th + g * 4;                         // channel within the slab
            const int cend = min(L.C, tile * th + th);                 // this tile stops at th rows / the slab end
            const int64_t n0 = (int64_t)wg * L.C + c0;
            const int nend = (int)min((int64_t)rows_total, (int64_t)wg * L.C + cend);
            if (r < a.M && c0 < cend && n0 < nend) {
                const int64_t orow = a.row_idx ? (int64_t)a.row_idx[r] : (int64_t)r;
                if constexpr (SWIGLU) {
                    float gg[4] = {s[0].x, s[0].y, s[0].z, s[0].w}, uu[4] = {s[1].x, s[1].y, s[1].z, s[1].w};
                    epi_swiglu4(e, orow, (int)n0, nend, gg, uu);
                } else {
                    EpiCtx et = e;
                    et.N = nend;
                    epi_store4(et, orow, (int)n0, s[0].x, s[0].y, s[0].z, s[0].w);
                }
            }
        }
    };

    if constexpr (XREG) {
        // chunks per tile padded to an even count so that chunk c always lives in buffer c & 1 (a padding chunk loads nothing)
        constexpr int MAXC = DG_XT / (U * KPU);
        static_assert(MAXC >= 2 && (MAXC % 2) == 0, "XREG: U must leave an even number (>= 2) of chunks for 16 k-tiles");
        const int nce = (nchunks + 1) & ~1;
        for (int tile = 0; tile < tpw; ++tile) {
            begin_tile(tile);
            static_for<0, MAXC>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                if (c < nce) {
                    Buf& b = (c & 1) ? b1 : b0;
                    consume(ci, b);
                    int t2 = tile, c2 = c + 2;       // chunk c + 2 (possibly of the next tile) reuses this buffer
                    if (c2 >= nce) { c2 -= nce; ++t2; }
                    if (t2 < tpw) load_chunk(t2, c2, b);
                }
            });
            finish_tile(tile);
        }
    } else {
        const int total = tpw * nchunks;
        if (total > 0) begin_tile(0);
        int tile = 0, c = 0;   // position of chunk q
        for (int q = 0; q < total; ++q) {
            Buf& b = (q & 1) ? b1 : b0;
            consume(std::integral_constant<int, 0>{}, b);
            int t2 = tile, c2 = c + 2;              // chunk q + 2 reuses this buffer
            while (c2 >= nchunks && t2 < tpw) { c2 -= nchunks; ++t2; }
            if (t2 < tpw) load_chunk(t2, c2, b);
            if (++c == nchunks) {
                finish_tile(tile);
                c = 0;
                if (++tile < tpw) begin_tile(tile);
            }
        }
    }
}

template <bool W8, int NORM, bool XREG, bool SWIGLU, int U>
static int launch_decode(const umv_gemm_args& a, const umv_decode_layout& L, int KT, hipStream_t s) {
    constexpr int NP = SWIGLU ? 2 : 1;
    size_t lds = (size_t)2 * DG_WAVES * NP * 64 * sizeof(f32x4);
    if (NORM) lds += DG_WAVES * 16 * sizeof(float) + (size_t)KT * 4 * NORM * 16 + DG_WAVES * 64 * 16;   // partials, xl, wl
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_decode_kernel<W8, NORM, XREG, SWIGLU, U>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL((gemm_decode_kernel<W8, NORM, XREG, SWIGLU, U>), dim3(L.G), dim3(DG_WAVES * 64), lds, s, a, L, KT);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

template <bool W8, bool SWIGLU>
static int dispatch_decode(const umv_gemm_args& a, const umv_decode_layout& L, int KT, hipStream_t s) {
    constexpr int UX = W8 ? (SWIGLU ? 4 : 4) : (SWIGLU ? 4 : 8);   // 16-byte loads per lane per part per chunk
    if (a.K <= DG_WAVES * DG_XT * 32) {
        if (a.norm_w) return a.M <= 8 ? launch_decode<W8, 8, true, SWIGLU, UX>(a, L, KT, s) : launch_decode<W8, 16, true, SWIGLU, UX>(a, L, KT, s);
        return launch_decode<W8, 0, true, SWIGLU, UX>(a, L, KT, s);
    }
    return launch_decode<W8, 0, false, SWIGLU, SWIGLU ? 2 : 4>(a, L, KT, s);
}

extern "C" int umv_gemm_decode(const umv_gemm_args* ap, const umv_decode_layout* Lp, int fp8, umv_stream_t stream) {
    UMV_CHECK(ap != nullptr && Lp != nullptr, UMV_ERR_ARG, "gemm_decode: null args");
    umv_gemm_args a = *ap;
    const umv_decode_layout L = *Lp;
    UMV_CHECK(a.x && a.wp && a.out, UMV_ERR_ARG, "gemm_decode: null pointer");
    UMV_CHECK(a.M >= 0 && a.N > 0 && a.K > 0, UMV_ERR_ARG, "gemm_decode: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    UMV_CHECK(a.M <= 16, UMV_ERR_UNSUPPORTED, "gemm_decode: the decode image serves M <= 16 (got %d); use umv_gemm_bf16", a.M);
    UMV_CHECK((a.K % 8) == 0 && (a.ldx % 8) == 0, UMV_ERR_ARG, "gemm_decode: K (%d) and ldx (%lld) must be multiples of 8", a.K,
              (long long)a.ldx);
    UMV_CHECK(!fp8 || a.w_scale, UMV_ERR_ARG, "gemm_decode: e4m3 image without scales");
    UMV_CHECK(!(a.epilogue & UMV_EPI_BIAS) || a.bias, UMV_ERR_ARG, "gemm_decode: BIAS without bias pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_RESIDUAL) || a.residual, UMV_ERR_ARG, "gemm_decode: RESIDUAL without residual pointer");
    const bool swiglu = (a.epilogue & UMV_EPI_SWIGLU) != 0;
    const int rows = swiglu ? a.N / 2 : a.N;
    UMV_CHECK(!swiglu || (a.N % 2) == 0, UMV_ERR_ARG, "gemm_decode: SWIGLU needs an even N");
    UMV_CHECK(L.th >= 1 && L.th <= 16 && L.tpw >= 1 && L.G >= 1 && (int64_t)L.G * L.C >= rows && L.tpw * L.th >= L.C, UMV_ERR_ARG,
              "gemm_decode: layout G=%d C=%d th=%d tpw=%d does not cover %d channels", L.G, L.C, L.th, L.tpw, rows);
    UMV_CHECK(!a.norm_w || a.K <= DG_WAVES * DG_XT * 32, UMV_ERR_UNSUPPORTED, "gemm_decode: fused RMSNorm needs K <= %d (got %d)",
              DG_WAVES * DG_XT * 32, a.K);
    if (a.M == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    const int KT = (a.K + 31) / 32;
    if (fp8) return swiglu ? dispatch_decode<true, true>(a, L, KT, s) : dispatch_decode<true, false>(a, L, KT, s);
    return swiglu ? dispatch_decode<false, true>(a, L, KT, s) : dispatch_decode<false, false>(a, L, KT, s);
}
