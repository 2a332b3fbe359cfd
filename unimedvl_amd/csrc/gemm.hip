// GEMM family for libunimedvl_hip (gfx950).  out[m,n] = epi(sum_k x[m,k] W[n,k]).
//
// W is pre-tiled (umv_pack_weight_bf16) into MFMA A-fragment order so that one
// wavefront instruction fetches a whole 16(n) x 32(k) tile as 1 KiB contiguous:
//   P[nt][kt][lane][8],  lane = g*16 + r,  element j  <->  W[nt*16 + r][kt*32 + g*8 + j]
// With W as the A operand and x^T as the B operand of v_mfma_f32_16x16x32_bf16 the
// accumulator of lane (r,g) holds out[m = r][n = g*4 + reg]: four consecutive n per
// lane, i.e. one 8-byte bf16x4 store per 16x16 tile.
//
// Two kernels:
//   * gemm_skinny  (M <= 64, decode / MoT text rows): HBM-bound weight streaming.
//     One workgroup = NT n-tiles x all of K; its 8 waves split K and reduce through
//     LDS (deterministic, no atomics, no inter-workgroup split-K).  Weight fragments
//     go straight from HBM to VGPRs (no LDS round trip: each byte is used once).
//   * gemm_tiled   (M > 64, prefill / ViT / diffusion): 128x128 workgroup tile,
//     4 waves of 64x64, W fragments from the packed stream, x staged through LDS.
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include "gemm_epilogue.h"
#include "gemm_internal.h"
#include <string.h>
#include <stdlib.h>

// ----------------------------------------------------------------------------- skinny (M <= 64)
// Weight streaming, HBM-bound.  One workgroup = NT n-tiles x all of K; its 8 waves take
// contiguous K slices and reduce through LDS.  Each wave keeps U weight fragments per n-tile
// in flight (U KiB of contiguous HBM per n-tile) and, when DB, prefetches the next chunk
// while the MFMAs of the current one issue.  NORM fuses Qwen2RMSNorm (modeling_qwen2.py:89-94)
// of the x rows into the prologue: the wave holds its whole K slice of x in registers,
// the row sum of squares is combined across waves through LDS, and the fragments are
// normalised in place with the reference's two bf16 roundings before feeding the MFMAs.
#define SK_WAVES 8
#define SK_XMAX 16   // NORM: K <= 8*16*32 = 4096

template <int MB, int NT, int U, int XL = 0>
struct SkBuf {
    bf16x8 w[U][NT];
    bf16x8 x[U][MB];
    u32x4 xp[XL ? U / 2 : 1][XL == 1 ? 1 : (XL ? 2 * MB : 1)];     // XL: the x pieces of the chunk's k-tile pairs on their way to LDS
};

// XL (round 4): x reaches the MFMAs in FULL 128-byte lines.  The plain kernel loads x in fragment shape - per wave instruction
// 16 rows x 64 bytes, half a line per row - and every workgroup re-reads all of x through L2 -> L1 (at 8 rows that is half the
// weight bytes, at 32 rows twice them).  Here a wave loads 8 rows x 128 bytes per instruction (a k-tile PAIR of 8 rows, lane L:
// row L >> 3, chunk (L & 7) ^ (L >> 3)), parks the piece in its own KiB of LDS (lane-linear ds_write_b128: the image is row-major,
// XOR-swizzled by the row) and reads the B fragments of the two k-tiles back conflict free (lane (r, g), k half h: chunk
// (4h + g) ^ (r & 7) of row r) - the tiled kernel's full-line staging (SCHED = 3) without the DMA.  Same operands, same MFMAs,
// same order: bit-identical to the plain kernel.  XL = 2: two pieces per 16-row tile; XL = 1 (M <= 8): one piece, rows 8..15 of a
// fragment re-read rows 0..7 (their output columns are never stored).  Needs an even U.
template <int MB, int NT, int U, bool DB, int NORM, int XL = 0>   // NORM: 0 = off, 8 / 16 = fused RMSNorm keeping that many x rows
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(umv_gemm_args a, int KT, int NTT) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [SK_WAVES][NT*MB*4][64] (+ norm partials)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.x * NT;

    const bf16_t* xrow[MB];
    bool xvalid[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = mb * 16 + r;
        xvalid[mb] = m < a.M;
        int64_t row = xvalid[mb] ? (a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m) : 0;
        xrow[mb] = a.x + row * a.ldx;
    }
    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K (a.k_splits > 1): blockIdx.y owns the k-tiles [ks0, ks1) and stores raw fp32 partial sums
    const int nsplit = a.k_splits > 1 ? a.k_splits : 1;
    const int kts = (KT + nsplit - 1) / nsplit;
    const int ks0 = (int)blockIdx.y * kts, ks1 = min(KT, ks0 + kts);
    const int kt_per = (max(0, ks1 - ks0) + SK_WAVES - 1) / SK_WAVES;
    const int kt_begin = ks0 + wave * kt_per;
    const int kt_end = min(ks1, kt_begin + kt_per);
    // XL works on whole k-tile PAIRS (one 128-byte line of x per row): a slice that starts on an odd k-tile starts one tile early
    // with that tile's weights masked to zero - an MFMA that adds exact zeros (the x it multiplies is the neighbour wave's, finite).
    // PRECONDITION of "bit-identical to the plain kernel": x is finite.  Where x holds Inf / NaN in the neighbour's k-tile the masked
    // product is 0 * Inf = NaN and this wave's partial sum becomes NaN where the plain kernel's would not (the row's final result is
    // Inf / NaN either way - the neighbour's own product sees the same value; only WHICH of the two non-finite values differs).
    const int kt_lo = XL != 0 ? (kt_begin & ~1) : kt_begin;
    const int nk = max(0, kt_end - kt_lo);
    const int nchunks = (nk + U - 1) / U;
    // TH = rows per n-tile of the packed image (16 standard; < 16 for the exact-partition decode copies,
    // whose lanes r >= TH carry no row): tile (nt, kt) holds [g][r < TH][8] = 4*TH*8 elements
    const int TH = a.tile_rows > 0 ? a.tile_rows : 16;
    const int tile_elems = 4 * TH * 8;
    const bool rowlane = r < TH;
    const bf16_t* wbase[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool tv = (nt0 + t) < NTT;
        wbase[t] = a.wp + ((int64_t)(tv ? nt0 + t : 0) * KT) * tile_elems + (g * TH + (rowlane ? r : 0)) * 8;
    }
    static_assert(!XL || (U % 2 == 0 && NORM == 0), "full-line x staging: whole k-tile pairs per chunk, no fused norm");
    static_assert(XL != 1 || MB == 1, "one-piece staging serves one 16-row tile of at most 8 valid rows");
    constexpr int XLP = XL == 1 ? 1 : 2 * MB;          // pieces per k-tile pair
    // XL: this lane's row of each 8-row piece and its 16-byte chunk of the pair's 128 bytes
    const int xchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const bf16_t* xprow[XL ? XLP : 1];
    bool xpvalid[XL ? XLP : 1];
    if constexpr (XL != 0) {
#pragma unroll
        for (int q = 0; q < XLP; ++q) {
            const int m = q * 8 + (lane >> 3);
            xpvalid[q] = m < a.M;
            const int64_t row = xpvalid[q] ? (a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m) : 0;
            xprow[q] = a.x + row * a.ldx + xchunk * 8;
        }
    }
    char* xstage = reinterpret_cast<char*>(red) + wave * (U / 2 * XLP * 1024);      // XL: this wave's own staging KiBs
    auto load_chunk = [&](int c, SkBuf<MB, NT, U, XL>& b) {
        if constexpr (XL != 0) {
#pragma unroll
            for (int pr = 0; pr < U / 2; ++pr) {
                const int kt = kt_lo + c * U + 2 * pr;
                const int k = kt * 32 + xchunk * 8;
#pragma unroll
                for (int q = 0; q < XLP; ++q)
                    b.xp[pr][q] = (kt < kt_end && xpvalid[q] && k < a.K) ? *reinterpret_cast<const u32x4*>(xprow[q] + (int64_t)kt * 32)
                                                                         : (u32x4){0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt_lo + c * U + u;
            const bool ok = kt >= kt_begin && kt < kt_end;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                b.w[u][t] = (ok && rowlane) ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wbase[t] + (int64_t)kt * tile_elems))
                                            : zero_frag();
            if (!NORM && XL == 0) {
                const int k = kt * 32 + g * 8;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) b.x[u][mb] = (ok && xvalid[mb] && k < a.K) ? ldg_frag(xrow[mb] + k) : zero_frag();
            }
        }
    };
    // XL: the chunk's pieces go through the wave's LDS KiBs and come back as the B fragments the MFMAs take (LDS operations of
    // one wave execute in order and nobody else touches these bytes: no barrier)
    auto unstage = [&](SkBuf<MB, NT, U, XL>& b) {
        if constexpr (XL != 0) {
#pragma unroll
            for (int pr = 0; pr < U / 2; ++pr)
#pragma unroll
                for (int q = 0; q < XLP; ++q) *reinterpret_cast<u32x4*>(xstage + (pr * XLP + q) * 1024 + lane * 16) = b.xp[pr][q];
            const int rr = XL == 1 ? (r & 7) : r;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    b.x[u][mb] = *reinterpret_cast<const bf16x8*>(xstage + (u >> 1) * (XLP * 1024) + (mb * 16 + rr) * 128 +
                                                                  ((((u & 1) * 4 + g) ^ (rr & 7)) << 4));
        }
    };
    SkBuf<MB, NT, U, XL> b0, b1;
    if (nchunks > 0) load_chunk(0, b0);

    if constexpr (NORM != 0) {
        static_assert(!NORM || MB == 1, "fused RMSNorm supports M <= 16");
        // Stage RMSNorm(x) * norm_w ONCE per workgroup into LDS, already in MFMA B-fragment order:
        // slot (kt, g, r) holds the 8 bf16 of row r at k = kt*32 + g*8.  MP = rows kept (8 or 16).
        constexpr int MP = NORM ? NORM : 8;
        bf16_t* xl = reinterpret_cast<bf16_t*>(red + SK_WAVES * NT * MB * 4 * 64 + SK_WAVES * 16);
        float* part = red + SK_WAVES * NT * MB * 4 * 64;   // [SK_WAVES][16]
        const int rr = tid & (MP - 1);
        constexpr int sh = MP == 8 ? 3 : 4;        // log2(MP)
        constexpr int XS = MP;                     // 16-byte groups per thread: (4096/32) * 4 * MP / 512
        const int nslots = KT * 4 * MP;
        const bool rowok = rr < a.M;
        const bf16_t* xr = a.x + (rowok ? (a.row_idx ? (int64_t)a.row_idx[rr] : (int64_t)rr) : 0) * a.ldx;
        // ONE batch of loads per phase: a loop of dependent L2 round trips (7 per pass at K = 3584) costs ~10 us per
        // workgroup, a batch about one round trip.  Phase 1: x -> registers -> row sums of squares, raw x parked in LDS.
        {
            bf16x8 xv[XS];
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                const int sidx = tid + i * SK_WAVES * 64;
                const int k = (sidx >> (sh + 2)) * 32 + ((sidx >> sh) & 3) * 8;
                xv[i] = (sidx < nslots && k < a.K && rowok) ? ldg_frag(xr + k) : zero_frag();
            }
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < XS; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = bf2f((bf16_t)xv[i][j]);
                    ss += f * f;
                }
                const int sidx = tid + i * SK_WAVES * 64;
                if (sidx < nslots) *reinterpret_cast<bf16x8*>(xl + (int64_t)sidx * 8) = xv[i];
            }
            // lanes sharing a row: lane & (MP-1)
            if (MP == 8) ss += __shfl_xor(ss, 8, 64);
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (lane < MP) part[wave * 16 + lane] = ss;
        }
        // phase 2: norm_w batch (in flight across the barrier), then normalise the thread's own slots in place
        bf16x8 wv[XS];
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int sidx = tid + i * SK_WAVES * 64;
            const int k = (sidx >> (sh + 2)) * 32 + ((sidx >> sh) & 3) * 8;
            wv[i] = (sidx < nslots && k < a.K) ? ldg_frag(a.norm_w + k) : zero_frag();
        }
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) tot += part[w * 16 + rr];
        const float rstd = rsqrt_ieee(tot / (float)a.K + a.norm_eps);
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const int sidx = tid + i * SK_WAVES * 64;
            if (sidx < nslots) {
                bf16x8 v = *reinterpret_cast<const bf16x8*>(xl + (int64_t)sidx * 8), o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(bf2f((bf16_t)wv[i][j]) * rbf(bf2f((bf16_t)v[j]) * rstd));   // two roundings
                *reinterpret_cast<bf16x8*>(xl + (int64_t)sidx * 8) = o;
            }
        }
        __syncthreads();
        const int rsel = r & (MP - 1);
        auto xfrag = [&](int kt) -> bf16x8 {
            return *reinterpret_cast<const bf16x8*>(xl + ((int64_t)(kt * 4 + g) * MP + rsel) * 8);
        };
        for (int c = 0; c < nchunks; c += 2) {
            if (DB && c + 1 < nchunks) load_chunk(c + 1, b1);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kt = min(kt_begin + c * U + u, KT - 1);
                bf16x8 xf = xfrag(kt);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][0] = mfma16(b0.w[u][t], xf, acc[t][0]);
            }
            if (c + 1 < nchunks) {
                if (!DB) load_chunk(c + 1, b1);
                if (c + 2 < nchunks && DB) load_chunk(c + 2, b0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kt = min(kt_begin + (c + 1) * U + u, KT - 1);
                    bf16x8 xf = xfrag(kt);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t][0] = mfma16(b1.w[u][t], xf, acc[t][0]);
                }
                if (c + 2 < nchunks && !DB) load_chunk(c + 2, b0);
            }
        }
    } else {
        for (int c = 0; c < nchunks; c += 2) {
            if (DB && c + 1 < nchunks) load_chunk(c + 1, b1);
            unstage(b0);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16(b0.w[u][t], b0.x[u][mb], acc[t][mb]);
            if (c + 1 < nchunks) {
                if (!DB) load_chunk(c + 1, b1);
                if (c + 2 < nchunks && DB) load_chunk(c + 2, b0);
                unstage(b1);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16(b1.w[u][t], b1.x[u][mb], acc[t][mb]);
                if (c + 2 < nchunks && !DB) load_chunk(c + 2, b0);
            }
        }
    }
    // cross-wave reduction through LDS
    if constexpr (XL != 0) __syncthreads();      // the reduction buffer overlays the waves' x staging KiBs: everyone has left the main loop
    constexpr int E4 = NT * MB;  // f32x4 fragments per lane
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            f32x4* dst = reinterpret_cast<f32x4*>(red) + ((wave * E4 + t * MB + mb) * 64 + lane);
            *dst = acc[t][mb];
        }
    __syncthreads();
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    if (nsplit > 1) {   // partial sums: fp32, no bias / activation / residual (the consumer kernel finishes the row)
        e.out = reinterpret_cast<float*>(a.out) + (int64_t)blockIdx.y * a.split_stride;
        e.flags = UMV_EPI_OUT_F32;
    }
    if (a.epilogue & UMV_EPI_SWIGLU) {
        // tiles come in (gate, up) pairs; NT is even
        for (int idx = tid; idx < (NT / 2) * MB * 64; idx += SK_WAVES * 64) {
            int l = idx & 63;
            int f = idx >> 6;  // pair*MB + mb
            int pair = f / MB, mb = f % MB;
            f32x4 sg = {0, 0, 0, 0}, su = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) {
                sg += reinterpret_cast<f32x4*>(red)[(w * E4 + (2 * pair) * MB + mb) * 64 + l];
                su += reinterpret_cast<f32x4*>(red)[(w * E4 + (2 * pair + 1) * MB + mb) * 64 + l];
            }
            int m = mb * 16 + (l & 15);
            int ntile = nt0 + 2 * pair;
            if (m < a.M && ntile < NTT) {
                int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
                int c0 = (ntile >> 1) * 16 + (l >> 4) * 4;
                float gg[4] = {sg.x, sg.y, sg.z, sg.w}, uu[4] = {su.x, su.y, su.z, su.w};
                epi_swiglu4(e, orow, c0, a.N / 2, gg, uu);
            }
        }
    } else {
        for (int idx = tid; idx < E4 * 64; idx += SK_WAVES * 64) {
            int l = idx & 63;
            int f = idx >> 6;  // t*MB + mb
            int t = f / MB, mb = f % MB;
            f32x4 s = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) s += reinterpret_cast<f32x4*>(red)[(w * E4 + f) * 64 + l];
            int m = mb * 16 + (l & 15);
            int n0 = (nt0 + t) * TH + (l >> 4) * 4;
            int nend = min(a.N, (nt0 + t) * TH + TH);            // rows of this tile stop at TH
            const bool valid = m < a.M && n0 < nend;
            float fin[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
                EpiCtx et = e;
                et.N = nend;
                epi_store4(et, orow, n0, s.x, s.y, s.z, s.w, fin);
            }
            if (a.argmax_partial && nt0 + t < NTT)   // wave-uniform: greedy argmax rides on the lm_head epilogue
                epi_argmax_tile(a.argmax_partial, NTT, m, nt0 + t, l, valid, n0, nend, fin, a.sample_temperature, a.sample_seed, a.sample_step);
        }
    }
}

// ----------------------------------------------------------------------------- fp8 weights (decode, BASELINE.json configs[4])
// Weight-only e4m3 (OCP) with one power-of-two scale per output channel (an E8M0 exponent, as in the MX
// formats): W' = q * 2^e is exactly representable in bf16, so streaming q and converting in registers
// (v_cvt_scalef32_pk_bf16_fp8: two elements per instruction, the channel scale rides along for free) feeds the
// SAME bf16 MFMAs with the SAME operands as the bf16 kernel on W'.  Decode reads half the bytes; prefill /
// diffusion keep using the bf16 image of W' on the tiled kernel, and both paths agree bit for bit.
//   image: P8[nt][kt8][lane][16 B], lane = g*16 + r; bytes 0..7  <-> W[nt*16 + r][kt8*64 +      g*8 + j]
//                                                    bytes 8..15 <-> W[nt*16 + r][kt8*64 + 32 + g*8 + j]
//   scale: f32 [ntt*16] in packed row order (SwiGLU images interleave gate / up 16-row tiles like the bf16 one)


template <int MB, int NT, int U, int XL = 0>
struct SkBuf8 {
    u32x4 w[U][NT];
    bf16x8 x[U][2][MB];
    u32x4 xp[XL ? U : 1][XL == 1 ? 1 : (XL ? 2 * MB : 1)];      // XL: the 8-row x 128-byte x pieces of the chunk's k super-tiles
};

// Same work decomposition as gemm_skinny_kernel (8 waves split K in contiguous slices, LDS reduce in wave
// order), over 64-wide k super-tiles; for K % 512 == 0 the slices - and so the fp32 sums - are identical.
// XL = full-line x staging as in gemm_skinny_kernel (a 64-wide k super-tile is exactly one 128-byte line per row): 1 = M <= 8,
// one piece per super-tile (rows 8..15 of the fragment re-read rows 0..7: their output columns are never stored), 2 = two pieces
// per 16-row tile.
template <int MB, int NT, int U, int XL = 0>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny8_kernel(umv_gemm_args a, int KT8, int NTT) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.x * NT;
    const uint8_t* wq = reinterpret_cast<const uint8_t*>(a.wp);

    const bf16_t* xrow[MB];
    bool xvalid[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = mb * 16 + r;
        xvalid[mb] = m < a.M;
        int64_t row = xvalid[mb] ? (a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m) : 0;
        xrow[mb] = a.x + row * a.ldx;
    }
    f32x4 acc[NT][MB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[t][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsplit = a.k_splits > 1 ? a.k_splits : 1;   // split-K as in gemm_skinny_kernel (over 64-wide super-tiles)
    const int kts = (KT8 + nsplit - 1) / nsplit;
    const int ks0 = (int)blockIdx.y * kts, ks1 = min(KT8, ks0 + kts);
    const int kt_per = (max(0, ks1 - ks0) + SK_WAVES - 1) / SK_WAVES;
    const int kt_begin = ks0 + wave * kt_per;
    const int kt_end = min(ks1, kt_begin + kt_per);
    const int nk = max(0, kt_end - kt_begin);
    const int nchunks = (nk + U - 1) / U;
    const uint8_t* wbase[NT];
    float wscale[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool tv = (nt0 + t) < NTT;
        const int nt = tv ? nt0 + t : 0;
        wbase[t] = wq + ((int64_t)nt * KT8 * 64 + lane) * 16;
        wscale[t] = a.w_scale[nt * 16 + r];
    }
    // x costs as much L2->L1 traffic as the e4m3 weights at NT = 1 (M = 8 rows x 2 B vs 16 rows x 1 B per k) and is what
    // holds this kernel below the HBM rate: down_proj 18.2 us with these x loads, 12.8 us with x from a constant
    // (tools/skinny_bench.py; rotating the K order per workgroup or pre-packing x in fragment order did not help).
    static_assert(XL != 1 || MB == 1, "one-piece staging serves one 16-row tile of at most 8 valid rows");
    constexpr int XLP = XL == 1 ? 1 : 2 * MB;        // pieces per k super-tile
    const int xchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const bf16_t* xprow[XL ? XLP : 1];
    bool xpvalid[XL ? XLP : 1];
    if constexpr (XL != 0) {
#pragma unroll
        for (int q = 0; q < XLP; ++q) {
            const int m = q * 8 + (lane >> 3);
            xpvalid[q] = m < a.M;
            const int64_t row = xpvalid[q] ? (a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m) : 0;
            xprow[q] = a.x + row * a.ldx + xchunk * 8;
        }
    }
    char* xstage = reinterpret_cast<char*>(red) + wave * (U * XLP * 1024);
    auto load_chunk = [&](int c, SkBuf8<MB, NT, U, XL>& b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = kt_begin + c * U + u;
            const bool ok = kt < kt_end;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                b.w[u][t] = ok ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wbase[t] + (int64_t)kt * 1024)) : (u32x4){0u, 0u, 0u, 0u};
            if constexpr (XL != 0) {
                const int k = kt * 64 + xchunk * 8;
#pragma unroll
                for (int q = 0; q < XLP; ++q)
                    b.xp[u][q] = (ok && xpvalid[q] && k < a.K) ? *reinterpret_cast<const u32x4*>(xprow[q] + (int64_t)kt * 64) : (u32x4){0u, 0u, 0u, 0u};
            } else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kt * 64 + h * 32 + g * 8;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) b.x[u][h][mb] = (ok && xvalid[mb] && k < a.K) ? ldg_frag(xrow[mb] + k) : zero_frag();
            }
        }
    };
    auto consume = [&](SkBuf8<MB, NT, U, XL>& b) {
        if constexpr (XL != 0) {       // pieces -> the wave's own LDS KiBs (row-major, XOR-swizzled by the row) -> B fragments
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < XLP; ++q) *reinterpret_cast<u32x4*>(xstage + (u * XLP + q) * 1024 + lane * 16) = b.xp[u][q];
            const int rr = XL == 1 ? (r & 7) : r;
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        b.x[u][h][mb] = *reinterpret_cast<const bf16x8*>(xstage + u * (XLP * 1024) + (mb * 16 + rr) * 128 + (((h * 4 + g) ^ (rr & 7)) << 4));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 wlo[NT], whi[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) cvt_fp8x16(b.w[u][t], wscale[t], wlo[t], whi[t]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16(wlo[t], b.x[u][0][mb], acc[t][mb]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[t][mb] = mfma16(whi[t], b.x[u][1][mb], acc[t][mb]);
        }
    };
    SkBuf8<MB, NT, U, XL> b0, b1;
    if (nchunks > 0) load_chunk(0, b0);
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks) load_chunk(c + 1, b1);
        consume(b0);
        if (c + 1 < nchunks) {
            if (c + 2 < nchunks) load_chunk(c + 2, b0);
            consume(b1);
        }
    }
    // cross-wave reduction + epilogue: identical to gemm_skinny_kernel (16-row tiles)
    if constexpr (XL != 0) __syncthreads();      // the reduction buffer overlays the waves' x staging KiBs
    constexpr int E4 = NT * MB;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) reinterpret_cast<f32x4*>(red)[(wave * E4 + t * MB + mb) * 64 + lane] = acc[t][mb];
    __syncthreads();
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    if (nsplit > 1) {
        e.out = reinterpret_cast<float*>(a.out) + (int64_t)blockIdx.y * a.split_stride;
        e.flags = UMV_EPI_OUT_F32;
    }
    if (a.epilogue & UMV_EPI_SWIGLU) {
        for (int idx = tid; idx < (NT / 2) * MB * 64; idx += SK_WAVES * 64) {
            int l = idx & 63;
            int f = idx >> 6;
            int pair = f / MB, mb = f % MB;
            f32x4 sg = {0, 0, 0, 0}, su = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) {
                sg += reinterpret_cast<f32x4*>(red)[(w * E4 + (2 * pair) * MB + mb) * 64 + l];
                su += reinterpret_cast<f32x4*>(red)[(w * E4 + (2 * pair + 1) * MB + mb) * 64 + l];
            }
            int m = mb * 16 + (l & 15);
            int ntile = nt0 + 2 * pair;
            if (m < a.M && ntile < NTT) {
                int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
                int c0 = (ntile >> 1) * 16 + (l >> 4) * 4;
                float gg[4] = {sg.x, sg.y, sg.z, sg.w}, uu[4] = {su.x, su.y, su.z, su.w};
                epi_swiglu4(e, orow, c0, a.N / 2, gg, uu);
            }
        }
    } else {
        for (int idx = tid; idx < E4 * 64; idx += SK_WAVES * 64) {
            int l = idx & 63;
            int f = idx >> 6;
            int t = f / MB, mb = f % MB;
            f32x4 s = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < SK_WAVES; ++w) s += reinterpret_cast<f32x4*>(red)[(w * E4 + f) * 64 + l];
            int m = mb * 16 + (l & 15);
            int n0 = (nt0 + t) * 16 + (l >> 4) * 4;
            const bool valid = m < a.M && n0 < a.N;
            float fin[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
                epi_store4(e, orow, n0, s.x, s.y, s.z, s.w, fin);
            }
            if (a.argmax_partial && nt0 + t < NTT)
                epi_argmax_tile(a.argmax_partial, NTT, m, nt0 + t, l, valid, n0, a.N, fin, a.sample_temperature, a.sample_seed, a.sample_step);
        }
    }
}

static int skinny8_xl() {     // UMV_SKINNY8_XL: 0 never, 1 above 8 rows, 2 always (default; A/B only)
    static const int v = umv_env_int("UMV_SKINNY8_XL", 2);
    return v;
}

template <int MB, int NT, int U, int XL>
static int launch_skinny8_v(const umv_gemm_args& a, int KT8, int NTT, hipStream_t s) {
    int blocks = (NTT + NT - 1) / NT;
    size_t lds = (size_t)SK_WAVES * NT * MB * 4 * 64 * sizeof(float);
    if (XL != 0) {
        const size_t xl = (size_t)SK_WAVES * U * (XL == 1 ? 1 : 2 * MB) * 1024;
        if (xl > lds) lds = xl;
        static bool attr_set[UMV_MAX_DEVICES] = {};
        if (lds > 64 * 1024 && umv_first_on_device(attr_set))
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny8_kernel<MB, NT, U, XL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm_skinny8_kernel<MB, NT, U, XL>), dim3(blocks, a.k_splits > 1 ? a.k_splits : 1), dim3(SK_WAVES * 64), lds, s, a,
                       KT8, NTT);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

template <int MB, int NT, int U>
static int launch_skinny8(const umv_gemm_args& a, int KT8, int NTT, hipStream_t s) {
    const bool lines = (a.ldx % 64) == 0 && ((uintptr_t)a.x % 128) == 0;      // rows start on a 128-byte boundary
    const int mode = skinny8_xl();
    if (lines && mode && (mode > 1 || a.M > 8)) {
        if constexpr (MB == 1) {
            if (a.M <= 8) return launch_skinny8_v<MB, NT, U, 1>(a, KT8, NTT, s);
        }
        return launch_skinny8_v<MB, NT, U, 2>(a, KT8, NTT, s);
    }
    return launch_skinny8_v<MB, NT, U, 0>(a, KT8, NTT, s);
}

extern "C" int umv_gemm_fp8w(const umv_gemm_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap != nullptr, UMV_ERR_ARG, "gemm_fp8w: null args");
    umv_gemm_args a = *ap;
    UMV_CHECK(a.x && a.wp && a.out && a.w_scale, UMV_ERR_ARG, "gemm_fp8w: null pointer (x, wp, out and w_scale are required)");
    UMV_CHECK(a.M >= 0 && a.N > 0 && a.K > 0, UMV_ERR_ARG, "gemm_fp8w: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    UMV_CHECK(a.M <= 64, UMV_ERR_UNSUPPORTED, "gemm_fp8w: the e4m3 image is the decode (M <= 64) layout; use the bf16 image of the "
              "dequantised weights with umv_gemm_bf16 for M=%d", a.M);
    UMV_CHECK((a.K % 8) == 0 && (a.ldx % 8) == 0, UMV_ERR_ARG, "gemm_fp8w: K (%d) and ldx (%lld) must be multiples of 8", a.K,
              (long long)a.ldx);
    UMV_CHECK(!(a.epilogue & UMV_EPI_BIAS) || a.bias, UMV_ERR_ARG, "gemm_fp8w: BIAS without bias pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_RESIDUAL) || a.residual, UMV_ERR_ARG, "gemm_fp8w: RESIDUAL without residual pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_SWIGLU) || (a.N % 32) == 0, UMV_ERR_ARG, "gemm_fp8w: SWIGLU needs N %% 32 == 0");
    UMV_CHECK(!a.norm_w && (a.tile_rows == 0 || a.tile_rows == 16), UMV_ERR_UNSUPPORTED, "gemm_fp8w: no fused norm / th-row tiles");
    UMV_CHECK(a.k_splits <= 1 || (!(a.epilogue & UMV_EPI_SWIGLU) && a.split_stride > 0 && a.k_splits <= 64), UMV_ERR_UNSUPPORTED,
              "gemm_fp8w: split-K (k_splits=%d) needs no SwiGLU, split_stride > 0, k_splits <= 64", a.k_splits);
    UMV_CHECK(!a.argmax_partial || (a.k_splits <= 1 && !a.row_idx && !(a.epilogue & (UMV_EPI_SWIGLU | UMV_EPI_OUT_F32))), UMV_ERR_UNSUPPORTED,
              "gemm_fp8w: argmax_partial needs bf16 out, no SwiGLU / split-K / row_idx");
    if (a.M == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    const int KT8 = (a.K + 63) / 64, NTT = (a.N + 15) / 16;
    if (a.k_splits > 1) {   // split-K decode mode: see umv_gemm_bf16
        if (a.M <= 16) return launch_skinny8<1, 4, 1>(a, KT8, NTT, s);
        if (a.M <= 32) return launch_skinny8<2, 4, 1>(a, KT8, NTT, s);
        return launch_skinny8<4, 4, 1>(a, KT8, NTT, s);
    }
    const bool two = (a.epilogue & UMV_EPI_SWIGLU) || NTT >= 1024;
    // (NT, U) from a sweep on MI355X at M = 8 (tools/skinny_bench.py, FP8=1): gate/up 25.5 us with <2,2> (110 VGPRs, two
    // workgroups per CU) vs 33.9 <2,4>, 27.9 <4,1>; down_proj 18.2 us with <1,8> vs 23.7 for NT = 2 (only 112 workgroups)
    if (a.M <= 16) return two ? launch_skinny8<1, 2, 2>(a, KT8, NTT, s) : launch_skinny8<1, 1, 8>(a, KT8, NTT, s);
    static const int nt4 = umv_env_int("UMV_GEMM_SKINNY_NT", 4) == 2 ? 0 : 1;   // as in umv_gemm_bf16: 4 n-tiles per workgroup on the wide-N GEMMs at M > 16 (UMV_GEMM_SKINNY_NT=2 reverts)
    if (two && nt4) return a.M <= 32 ? launch_skinny8<2, 4, 1>(a, KT8, NTT, s) : launch_skinny8<4, 4, 1>(a, KT8, NTT, s);
    if (a.M <= 32) return two ? launch_skinny8<2, 2, 2>(a, KT8, NTT, s) : launch_skinny8<2, 1, 4>(a, KT8, NTT, s);
    return two ? launch_skinny8<4, 2, 1>(a, KT8, NTT, s) : launch_skinny8<4, 1, 2>(a, KT8, NTT, s);
}

// ----------------------------------------------------------------------------- tiled (M > 64)
// Workgroup tile 128(n) x 128(m) x 64(k), 4 waves as 2(n) x 2(m), wave tile 64 x 64 = 4x4 MFMA
// tiles, two LDS buffers of 32 KiB filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction) while the MFMAs of the previous k-step run:
//   * the packed weight image IS the MFMA A-fragment order, so a W tile is a straight 1 KiB copy;
//   * an x fragment (16 rows x 64 B) is gathered by giving every lane its own source address
//     (row index list, K tail -> a zero page), so it lands in B-fragment order too.
// Every ds_read_b128 is lane-linear (conflict free) and no fragment passes through VGPRs on its way in.
__device__ __attribute__((aligned(16))) const uint32_t g_zero_page[4] = {0, 0, 0, 0};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// inline-asm building blocks of the interleaved schedule (free functions: clang rejects asm operands that name locals of
// the enclosing function from inside a generic lambda)
template <int OFF>
__device__ __forceinline__ void lds_read_frag(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// read r of NRD goes right after MFMA number (r * SPAN) / NRD, SPAN = 3/4 of the step's MFMAs: evenly spread over the first three
// quarters, first one after the first MFMA, so that the last quarter's MFMAs cover the latency of the last reads before the
// step's closing s_waitcnt lgkmcnt(0) (spread over the whole step the last read sat 2 MFMAs before that wait)
// piece p of n_pieces goes behind MFMA floor((2p + 1) * n_mma / (2 * n_pieces)): evenly spread, never behind the last MFMA
__host__ __device__ constexpr int dma_slot(int i, int n_mma, int n_pieces) {
    for (int p = 0; p < n_pieces; ++p)
        if (((2 * p + 1) * n_mma) / (2 * n_pieces) == i) return p;
    return -1;
}
constexpr int interleave_slot(int i, int nmma, int nrd) {
    const int span = (nmma * 3 / 4 >= nrd) ? nmma * 3 / 4 : nmma;
    for (int r = 0; r < nrd; ++r)
        if ((r * span) / nrd == i) return r;
    return -1;
}
__device__ __forceinline__ void mfma16_asm(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// the same with the accumulator in the AGPR half of the register file (4-wave tiles with 128 x 128 per wave: 256 accumulator
// registers, one wave per SIMD on the full 512-entry file)
__device__ __forceinline__ void mfma16_asm_acc(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
typedef __attribute__((ext_vector_type(16))) float f32x16;   // one 32x32 MFMA C/D fragment
__device__ __forceinline__ void mfma32_asm(f32x16& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// WN x WM waves, each owning TN x TM MFMA tiles: workgroup tile (WN*TN*16)(n) x (WM*TM*16)(m) x (KTS*32)(k),
// NBUF LDS buffers.  Pipeline per k-step t (one raw s_barrier, never a full vmcnt drain in steady state):
//     s_waitcnt vmcnt((NBUF-2) tiles)   my part of tile t has landed, tiles t+1.. stay in flight
//     s_barrier                         everyone's part of tile t landed AND everyone finished reading tile t-1
//     issue LDS-DMA for tile t+NBUF-1   into the buffer tile t-1 just vacated
//     ds_read fragments of tile t, MFMAs
// SCHED = 1: the same pipeline with the MFMAs of tile t and the ds_reads of tile t+1 interleaved by hand (see below).
template <int WN, int WM, int TN, int TM, int KTS, int NBUF, int SCHED = 0>
__global__ __launch_bounds__(WN * WM * 64) void gemm_tiled_kernel(umv_gemm_args a, int KT, int NTT, int mblocks, int nblocks, int gn,
                                                                  int ksplit, int ms, int lean) {
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int WTILES = BN / 16 * KTS, XTILES = BM / 16 * KTS;      // 1 KiB fragment tiles per k-step
    constexpr int NT_ALL = WTILES + XTILES;
    constexpr int TPW = (NT_ALL + NW - 1) / NW;                          // tiles staged per wave per k-step
    // A tile whose 1 KiB pieces do not divide evenly over the waves (288 x 128: 26, 224 x 128: 22) rounds TPW up: the surplus
    // slots copy the zero page into a spare KiB each behind the buffers, so that every wave issues the same number of
    // LDS-DMA pieces per k-step and the counted s_waitcnt vmcnt(N) below stay exact.
    constexpr int NDUMMY = NW * TPW - NT_ALL;
    constexpr int BUF = NT_ALL * 1024;
    // bytes of the staging area (the tile's bias sits behind it): NBUF whole-step buffers, or - SCHED = 3 - a 3-slot ring of W
    // k-steps + a 3-slot ring of x k-step pairs
    constexpr int STAGE_BYTES = (SCHED & 15) == 3 ? 3 * WTILES * 1024 + 3 * BM * 128 : NBUF * BUF;
    constexpr int DUMPOFF = STAGE_BYTES + (BN * 2 + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];        // [NBUF][BUF]: W tiles [BN/16][KTS], then x tiles [BM/16][KTS]
    auto dst_of = [&](int buf, int f) -> char* { return (NDUMMY == 0 || f < NT_ALL) ? smem + buf * BUF + f * 1024 : smem + DUMPOFF + (f - NT_ALL) * 1024; };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    // XCD-aware order: blockIdx round-robins over the 8 XCDs, so give each XCD a contiguous run of
    // tiles (m fastest) and let its private L2 keep one W panel hot.
    // M super-blocks (ms m-blocks each, launch_tiled sizes them to ~64 MB of x): all XCDs work through one super-block
    // before the next, so that its x rows stay in the 256 MiB memory-side cache while the strips of W stream past - with 32
    // images (M = 32 832, x = 235 MB, act = 1.2 GB) every strip of n-blocks otherwise re-streams all of x from HBM.
    const int sb_tiles = ms * nblocks;
    const int sb = (int)blockIdx.x / sb_tiles;                        // this workgroup's super-block ...
    const int mb0 = sb * ms, mb_n = min(ms, mblocks - mb0);         // ... its m-blocks
    const int nwg = mb_n * nblocks;
    int bid = (int)blockIdx.x - sb * sb_tiles;
    {
        const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    // Tile order inside the run: strips of gn n-blocks, m-block next, n-block within the strip fastest - the ~32 tiles an
    // XCD works on at a time then cover (32 / gn) m-blocks x gn n-blocks and share both operands' k-slices in its L2
    // (gn = 1 is plain m-fastest: every CU of the XCD streams its own x panel and only W is shared).
    int mblk, nblk;
    {
        const int per = mb_n * gn, strip = bid / per, rem = bid - strip * per;
        const int w = min(gn, nblocks - strip * gn);
        mblk = mb0 + rem / w;
        nblk = strip * gn + rem % w;
    }
    const int m0 = mblk * BM;
    const int nt_blk = nblk * (BN / 16);
    const int nt_base = nt_blk + wn * TN;
    // the tile's BN bias values wait in LDS behind the staging buffers: read after the main loop, a global load there would
    // expose its whole latency once per tile (the first barrier of the main loop orders this write before any read)
    bf16_t* bias_lds = reinterpret_cast<bf16_t*>(smem + STAGE_BYTES);
    if ((a.epilogue & UMV_EPI_BIAS) && tid < BN) {
        const int n = nt_blk * 16 + tid;
        bias_lds[tid] = n < a.N ? a.bias[n] : (bf16_t)0;
    }
    // split-K (ksplit = k-tiles per split, 0 = none): blockIdx.y owns k-tiles [kt0, kt1) and stores raw fp32 partial sums
    // (the decode GEMMs with N = 3584 / 4608 at 65..128 rows: 14-36 workgroups otherwise)
    const int kt0 = ksplit ? (int)blockIdx.y * ksplit : 0;
    const int kt1 = ksplit ? min(KT, kt0 + ksplit) : KT;
    const int KTL = max(0, kt1 - kt0);
    const int nsteps = (KTL + KTS - 1) / KTS;

    // ---- staging: tile f = wave*TPW + i; f < WTILES copies a W tile, otherwise gathers an x tile
    const bf16_t* src[TPW];
    bool tvalid[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int f = wave * TPW + i;
        if (f < WTILES) {
            const int tl = f / KTS, kk = f % KTS;
            const int nt = nt_blk + tl;
            tvalid[i] = nt < NTT;
            src[i] = a.wp + ((int64_t)(tvalid[i] ? nt : 0) * KT + kt0 + kk) * 512 + lane * 8;
        } else if (NDUMMY != 0 && f >= NT_ALL) {
            tvalid[i] = false;                      // surplus slot: zero page, zero bump
            src[i] = reinterpret_cast<const bf16_t*>(g_zero_page);
        } else {
            const int fx = f - WTILES;
            const int tl = fx / KTS, kk = fx % KTS;
            const int m = m0 + tl * 16 + r;
            tvalid[i] = true;                       // rows past M are clamped (their outputs are masked)
            const int mm = m < a.M ? m : a.M - 1;
            const int64_t row = a.row_idx ? (int64_t)a.row_idx[mm] : (int64_t)mm;
            src[i] = a.x + row * a.ldx + (kt0 + kk) * 32 + g * 8;
            if constexpr ((SCHED >> 4) == 6) src[i] = a.x + (int64_t)m0 * a.ldx + fx * 512 + lane * 8;   // ablation 6: x read as contiguous KiB (wrong data)
        }
    }
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    // Fast path of the staging (every k-step but a ragged last one): one pointer bump per tile, no branches - this code
    // sits between the barrier and the first MFMA of every step.  A tile of n-rows past N reads the zero page with a zero
    // bump.  Steps are staged in order, so the pointers advance incrementally.
    const bf16_t* cur[TPW];
    int bump[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int f = wave * TPW + i;
        cur[i] = tvalid[i] ? src[i] : zero;
        bump[i] = !tvalid[i] ? 0 : (f < WTILES ? KTS * 512 : ((SCHED >> 4) == 6 ? XTILES * 512 : KTS * 32));
    }
    const bool ragged = (KTL % KTS) != 0 || ((a.K & 31) != 0 && kt1 == KT);     // the last k-step needs per-tile / per-lane zero fill
    auto stage = [&](int step, int buf) {
        if (ragged && step == nsteps - 1) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int f = wave * TPW + i;
                const bf16_t* p;
                if (f < WTILES) {
                    const int kt = step * KTS + f % KTS;
                    p = (tvalid[i] && kt < KTL) ? src[i] + (int64_t)step * (KTS * 512) : zero;
                } else if (NDUMMY != 0 && f >= NT_ALL) {
                    p = zero;
                } else {
                    const int kt = step * KTS + (f - WTILES) % KTS;
                    const int k = (kt0 + kt) * 32 + g * 8;
                    p = (kt < KTL && k < a.K) ? src[i] + (int64_t)step * (KTS * 32) : zero;
                }
                char* dst = dst_of(buf, f);
                __builtin_amdgcn_global_load_lds((const void*)p, (lds_ptr_t)dst, 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int f = wave * TPW + i;
            char* dst = dst_of(buf, f);
            __builtin_amdgcn_global_load_lds((const void*)cur[i], (lds_ptr_t)dst, 16, 0, 0);
            cur[i] += bump[i];
        }
    };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr bool PX = (SCHED & 15) == 3;     // full-line x staging: see the SCHED = 3 block below (own prologue)
#pragma unroll
    for (int p = 0; p < (PX ? 0 : NBUF - 1); ++p) {
        if (p < nsteps) stage(p, p);
        else if (SCHED != 0) {       // (K < 96) keep the number of pieces in flight uniform: see the interleaved loop below
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                __builtin_amdgcn_global_load_lds((const void*)zero, (lds_ptr_t)dst_of(p, wave * TPW + i), 16, 0, 0);
        }
    }
    // SCHED = 2: the interleaved schedule on v_mfma_f32_32x32x16_bf16.  Same LDS image (1 KiB tiles in 16x16x32 fragment order), same
    // number of fragment reads per k-step; a 32 x 32 x 16 operand is gathered from TWO neighbouring 16-row tiles by giving the
    // lanes the addresses  tile[(lane >> 4) & 1] + ((2h + (lane >> 5)) * 16 + (lane & 15)) * 16  (h = k half of the step) - the
    // four quarter-wave groups of a ds_read_b128 still cover 256 distinct bytes mod 256 each: conflict free.  Half the matrix
    // instructions per k-step (16 of 8 passes each instead of 32 of 4 for the 256 x 256 tile), the accumulators stay 128 registers.
    // SCHED >> 4 = ablation number (TIMING ONLY, results are wrong; UMV_GEMM_TILE=966x, profiles/r04_gemm_ablations.txt): 1 no LDS-DMA pieces
    // in the main loop, 2 every other fragment read, 3 no MFMAs, 4 no barrier, 5 MFMAs + barrier only, 6 x pieces read contiguous
    // KiB instead of 16 rows x 64 B, 7 no x pieces, 8 no W pieces
    constexpr int ABL = SCHED >> 4, SCH = SCHED & 15;
    constexpr bool M32 = SCH == 2;
    static_assert(!M32 || (TN % 2 == 0 && TM % 2 == 0), "32x32 MFMA tiles need even TN / TM");
    f32x16 acc32[M32 ? TN / 2 : 1][M32 ? TM / 2 : 1];
    if constexpr (M32) {
#pragma unroll
        for (int u = 0; u < TN / 2; ++u)
#pragma unroll
            for (int v = 0; v < TM / 2; ++v)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc32[u][v][q] = 0.f;
    }
    if constexpr (SCH == 1 || SCH == 2) {
        // Interleaved schedule (KTS == 1).  With the plain loop every wave leaves the barrier, issues its 12 ds_read_b128
        // at once and only then its 32 MFMAs: the 8 waves' 96 KiB of fragment reads keep the LDS pipe busy for ~768
        // cycles during which the matrix pipes mostly wait, then LDS idles for the ~1024 cycles of MFMAs - the two phases
        // add up (MfmaUtil 44 % from the PMC counters).  Here the fragments of tile t+1 are requested one ds_read at a
        // time, spread evenly between the MFMAs of tile t (every 2-3 MFMAs for the 256 x 256 tile), so each wave starts its MFMAs right after the barrier and the LDS traffic is spread
        // over the whole step.  MFMAs and ds_reads are inline asm so that the order is exactly the one written.
        static_assert(SCHED == 0 || (KTS == 1 && NBUF >= 3), "interleaved schedule needs KTS == 1 and >= 3 LDS buffers");
        bf16x8 wfA[TN], xfA[TM], wfB[TN], xfB[TM];
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
        // fragment i of a wave's W (x) tiles: 16x16x32 = tile i, lane-linear; 32x32x16 = tile pair i >> 1, k half i & 1
        const uint32_t lane_off = M32 ? (uint32_t)(((lane >> 4) & 1) * 1024 + ((lane >> 5) * 16 + (lane & 15)) * 16) : (uint32_t)(lane * 16);
        const uint32_t woff = wn * TN * 1024 + lane_off, xoff = WTILES * 1024 + wm * TM * 1024 + lane_off;
        constexpr auto frag_off = [](int i) constexpr { return M32 ? (i >> 1) * 2048 + (i & 1) * 512 : i * 1024; };
        auto wait_tiles = [&](int allowed) {   // tiles (of TPW DMA ops each) that may stay in flight
            if (allowed >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TPW) : "memory");
            else if (allowed == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto land = [&](bf16x8(&wf)[TN], bf16x8(&xf)[TM]) {   // all fragment reads issued so far have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < TN; ++t) asm volatile("" : "+v"(wf[t]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(xf[j]));
        };
        wait_tiles(NBUF - 2);
        UMV_BARRIER();
        static_for<0, TN>([&](auto T) {
            constexpr int t = decltype(T)::value;
            lds_read_frag<frag_off(t)>(wfA[t], lds0 + woff);
        });
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            lds_read_frag<frag_off(j)>(xfA[j], lds0 + xoff);
        });
        land(wfA, xfA);
        constexpr int NRD = TN + TM, NMMA = M32 ? TN * TM / 2 : TN * TM;
        static_assert(SCHED == 0 || NRD <= NMMA, "at most one fragment read per MFMA");
        // The TPW LDS-DMA pieces of tile step + NBUF - 1 are issued BETWEEN the MFMAs as well, one every NMMA / TPW MFMAs.
        // Issued in a burst behind the barrier (as the fragment reads once were) they keep the wave off the matrix pipe for
        // TPW x 100-185 cycles per k-step (the guide's price of a piece inside a busy phase) while its twin on the SIMD, in
        // lockstep, does the same.  That tile's buffer has been free since the barrier of the step before, and the counted
        // waits only need the pieces to be issued before the next step's wait: placement inside the step is free.  One
        // sequence for every step: the ragged last tile swaps its source pointers in before the sequence, and the last
        // NBUF - 1 steps, which have nothing left to stage, copy the zero page into the (free) buffer so that the count of
        // pieces in flight stays the same at every wait.
        static_assert(SCHED == 0 || TPW <= NMMA, "at most one DMA piece per MFMA");
        auto body = [&](int step, bf16x8(&wc)[TN], bf16x8(&xc)[TM], bf16x8(&wnx)[TN], bf16x8(&xnx)[TM]) {
            wait_tiles(NBUF - 3);                                        // tile step+1 landed (mine); tile step+2's pieces may fly
            if constexpr (ABL != 4) UMV_BARRIER();        // ... everyone's; and tile step-1's buffer is free
            const int st = step + NBUF - 1;                              // the tile staged during this step
            if (st >= nsteps) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) { cur[i] = zero; bump[i] = 0; }
            } else if (ragged && st == nsteps - 1) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int f = wave * TPW + i;
                    if (f < WTILES) {
                        const int kt = st * KTS + f % KTS;
                        cur[i] = (tvalid[i] && kt < KTL) ? src[i] + (int64_t)st * (KTS * 512) : zero;
                    } else if (NDUMMY != 0 && f >= NT_ALL) {
                        cur[i] = zero;
                    } else {
                        const int kt = st * KTS + (f - WTILES) % KTS;
                        const int k = (kt0 + kt) * 32 + g * 8;
                        cur[i] = (kt < KTL && k < a.K) ? src[i] + (int64_t)st * (KTS * 32) : zero;
                    }
                    bump[i] = 0;
                }
            }
            const int dma_buf = st % NBUF;
            // the reads of the last step fetch a tile nobody uses (the buffer exists): no branch inside the sequence
            const uint32_t nb = lds0 + ((step + 1) % NBUF) * BUF;
            const uint32_t wa = nb + woff, xa = nb + xoff;
            static_for<0, NMMA>([&](auto I) {
                constexpr int i = decltype(I)::value;
                if constexpr (ABL == 3) {
                } else if constexpr (M32) {      // k half outermost: two MFMAs on one accumulator are NMMA / 2 instructions apart
                    constexpr int h = i / (NMMA / 2), rem = i % (NMMA / 2), u = rem / (TM / 2), v = rem % (TM / 2);
                    mfma32_asm(acc32[u][v], wc[2 * u + h], xc[2 * v + h]);
                } else {
                    constexpr int t = i / TM, j = i % TM;
                    if constexpr (TN * TM > 40) mfma16_asm_acc(acc[t][j], wc[t], xc[j]);     // more accumulators than VGPRs can hold beside the fragments
                    else mfma16_asm(acc[t][j], wc[t], xc[j]);
                }
                constexpr int rd0 = interleave_slot(i, NMMA, NRD);   // the read (if any) that follows MFMA i
                constexpr int rd = (ABL == 5 || (ABL == 2 && (rd0 & 1))) ? -1 : rd0;
                if constexpr (rd >= 0 && rd < TN) lds_read_frag<frag_off(rd < TN ? rd : 0)>(wnx[rd < TN ? rd : 0], wa);
                else if constexpr (rd >= TN) lds_read_frag<frag_off(rd >= TN ? rd - TN : 0)>(xnx[rd >= TN ? rd - TN : 0], xa);
                constexpr int pc = dma_slot(i, NMMA, TPW);          // the DMA piece (if any) that follows MFMA i
                if constexpr (pc >= 0 && ABL != 1 && ABL != 5) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(ABL == 7 && wave * TPW + pc >= WTILES) && !(ABL == 8 && wave * TPW + pc < WTILES))
                    __builtin_amdgcn_global_load_lds((const void*)cur[pc], (lds_ptr_t)dst_of(dma_buf, wave * TPW + pc), 16, 0, 0);
                    cur[pc] += bump[pc];
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            land(wnx, xnx);
        };
        for (int step = 0; step < nsteps; step += 2) {
            body(step, wfA, xfA, wfB, xfB);
            if (step + 1 < nsteps) body(step + 1, wfB, xfB, wfA, xfA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing zero-page pieces: the epilogue reuses the buffers
        // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> VALU-read wait states by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if constexpr (M32) {
            // rename the 32 x 32 accumulators into the quads the epilogue takes (gemm_epilogue.h, L32): quad q of tile (u, v) is
            // rows q * 8 + 4 * (lane >> 5) .. + 3 of the tile's n side = column tile 2u + (q >> 1), column group 2 (q & 1) + (lane >> 5)
            static_for<0, TN / 2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                static_for<0, TM / 2>([&](auto V) {
                    constexpr int v = decltype(V)::value;
                    asm volatile("" : "+v"(acc32[u][v]));
                    static_for<0, 4>([&](auto Q) {
                        constexpr int q = decltype(Q)::value;
                        acc[2 * u + (q >> 1)][2 * v + (q & 1)] =
                            (f32x4){acc32[u][v][4 * q], acc32[u][v][4 * q + 1], acc32[u][v][4 * q + 2], acc32[u][v][4 * q + 3]};
                    });
                });
            });
        } else if constexpr (TN * TM > 40) {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < TM; ++j) asm volatile("" : "+a"(acc[t][j]));
        } else {
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(acc[t][j]));
        }
    } else if constexpr (PX) {
        // SCHED = 3: the interleaved schedule of SCHED = 1 with the x operand staged in FULL 128-byte lines.  The 1 KiB x piece of
        // SCHED = 1 gathers 16 rows x 64 bytes - half a line per row; the other half is fetched by the next k-step's piece, ~0.8 us
        // later, when the line has long left the 32 KiB L1 - so an x byte costs twice the L1 miss entries and L2 -> L1 traffic of a
        // W byte.  Measured at 8192^3 (TIMING-ONLY ablations, UMV_GEMM_TILE=966x): x pieces read as contiguous KiB 827 -> 766 us, no
        // x pieces 712, no W pieces 663, no pieces at all 535 (2.05 PFLOP/s), pieces and fragment reads without MFMAs 722 us AT
        // 2.4 GHz: the staging path, not the matrix pipe, sets the pace of this kernel.  (Issuing the two half-line pieces back
        // to back in one step was tried first: 846 -> 941 us - the second request does not merge with the miss in flight.)
        // Here a piece is 8 rows x 128 bytes = one k-step PAIR of 8 rows: lane L brings the 16-byte chunk (L & 7) ^ ((L >> 3) & 7) of
        // row L >> 3, so that the row-major image [row][8 chunks] in LDS is XOR-swizzled by the row and the B fragment read of k-tile
        // 2q + h, lane (r, g) -> chunk (4h + g) ^ (r & 7) of row r, is conflict free (each quarter-wave group covers all 64 banks).
        // Rings: W 3 slots of one k-step (W(t+3) takes the slot of tile t, free behind the head barrier of body t), x 3 slots of one
        // k-step pair (pair q is issued half in body 2q-5, half in body 2q-4): every wave issues WPW + XPB pieces per body, x first,
        // so the counted wait at the head of a body - W(t+1) landed, the pieces of the previous body may fly - is one constant.
        // Same MFMAs on the same operands in the same order: bit-identical to SCHED = 1.
        // (W tiles that do not divide evenly over the waves - 288 columns: 18 - round WPW up; the surplus slots copy the zero page
        // into a spare KiB each behind the bias so that every wave issues the same number of pieces)
        constexpr int WPW = (WTILES + NW - 1) / NW, XPP = (BM / 8) / NW, XPB = XPP / 2, NP = WPW + XPB;
        constexpr int WDUMP = STAGE_BYTES + (BN * 2 + 15) / 16 * 16;
        static_assert(KTS == 1 && (BM / 8) % NW == 0 && XPP % 2 == 0, "full-line x staging: even split of the x pieces over the waves");
        constexpr int WSLOT = WTILES * 1024, XSLOT = BM * 128, XBASE = 3 * WSLOT;
        constexpr int NRD = TN + TM, NMMA = TN * TM;
        static_assert(NRD <= NMMA && NP <= NMMA, "at most one read / piece per MFMA");
        const int kx_rel = min(a.K, kt1 * 32) - kt0 * 32;             // valid k (elements) of this block's range, relative to kt0
        const bf16_t* curW[WPW];
        int bumpW[WPW];
        const bf16_t* curX[XPP];
        const int xchunk = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int nt = nt_blk + wave * WPW + i;
            const bool ok = nt < NTT && wave * WPW + i < WTILES;
            curW[i] = ok ? a.wp + ((int64_t)nt * KT + kt0) * 512 + lane * 8 : zero;
            bumpW[i] = ok ? 512 : 0;
        }
#pragma unroll
        for (int i = 0; i < XPP; ++i) {
            const int m = m0 + (wave * XPP + i) * 8 + (lane >> 3);
            const int mm = m < a.M ? m : a.M - 1;                // rows past M are clamped (their outputs are masked)
            const int64_t row = a.row_idx ? (int64_t)a.row_idx[mm] : (int64_t)mm;
            curX[i] = a.x + row * a.ldx + kt0 * 32 + xchunk * 8;
        }
        // (char* and a cast at the call: a lambda RETURNING an address_space(3) pointer makes the host pass drop the kernel's stub
        // without a diagnostic - the library then fails to load with an undefined __device_stub__ symbol)
        auto dstW = [&](int slot, int i) -> char* {
            const int f = wave * WPW + i;
            return (WTILES % NW == 0 || f < WTILES) ? smem + slot * WSLOT + f * 1024 : smem + WDUMP + (f - WTILES) * 1024;
        };
        auto dstX = [&](int slot, int i) -> char* { return smem + XBASE + slot * XSLOT + (wave * XPP + i) * 1024; };
        const bf16_t* pw[WPW];
        const bf16_t* px[XPB];
        auto prep_w = [&](int kt) {                 // the W pieces of k-tile kt (relative to kt0): zero page past the K range
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                pw[i] = kt < KTL ? curW[i] : zero;
                curW[i] += bumpW[i];
            }
        };
        auto prep_x = [&](int q, auto HALF) {       // pieces [HALF * XPB, +XPB) of k-step pair q: per-lane zero fill at the K tail
            constexpr int hf = decltype(HALF)::value;
            const int k0 = q * 64;
            if (k0 + 64 <= kx_rel) {
#pragma unroll
                for (int i = 0; i < XPB; ++i) { px[i] = curX[hf * XPB + i]; curX[hf * XPB + i] += 64; }
            } else {
#pragma unroll
                for (int i = 0; i < XPB; ++i) { px[i] = (k0 + xchunk * 8 < kx_rel) ? curX[hf * XPB + i] : zero; curX[hf * XPB + i] += 64; }
            }
        };
        // prologue, in the order the loop would have issued it: x pair 0, W(0), x pair 1, W(1), first half of x pair 2, W(2)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            prep_x(t, std::integral_constant<int, 0>{});
#pragma unroll
            for (int i = 0; i < XPB; ++i) __builtin_amdgcn_global_load_lds((const void*)px[i], (lds_ptr_t)dstX(t, i), 16, 0, 0);
            if (t < 2) {
                prep_x(t, std::integral_constant<int, 1>{});
#pragma unroll
                for (int i = 0; i < XPB; ++i) __builtin_amdgcn_global_load_lds((const void*)px[i], (lds_ptr_t)dstX(t, XPB + i), 16, 0, 0);
            }
            prep_w(t);
#pragma unroll
            for (int i = 0; i < WPW; ++i) __builtin_amdgcn_global_load_lds((const void*)pw[i], (lds_ptr_t)dstW(t, i), 16, 0, 0);
        }
        bf16x8 wfA[TN], xfA[TM], wfB[TN], xfB[TM];
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
        const uint32_t woff = wn * TN * 1024 + lane * 16;
        // x fragment of k half h: row (wm * TM + j) * 16 + r, chunk (4h + g) ^ (r & 7)
        const uint32_t xoff0 = XBASE + (wm * TM * 16 + r) * 128 + ((g ^ (r & 7)) << 4), xoff1 = xoff0 ^ 64;
        auto land = [&](bf16x8(&wf)[TN], bf16x8(&xf)[TM]) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < TN; ++t) asm volatile("" : "+v"(wf[t]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(xf[j]));
        };
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XPP + XPB + 2 * WPW) : "memory");      // x pair 0 and W(0) landed; pair 1, W(1), half of pair 2 and W(2) may fly
        UMV_BARRIER();
        static_for<0, TN>([&](auto T) {
            constexpr int t = decltype(T)::value;
            lds_read_frag<t * 1024>(wfA[t], lds0 + woff);
        });
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            lds_read_frag<j * 2048>(xfA[j], lds0 + xoff0);
        });
        land(wfA, xfA);
        auto body = [&](auto EVEN, int step, bf16x8(&wc)[TN], bf16x8(&xc)[TM], bf16x8(&wnx)[TN], bf16x8(&xnx)[TM]) {
            constexpr bool even = decltype(EVEN)::value;
            // the fragments of tile `step` are in registers; tile step + 1 must have landed before its reads below
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
            UMV_BARRIER();                                   // ... everyone's; and the slot of W tile `step` is free
            const int q = (step + 5) >> 1;                   // the x pair this body stages half of
            if constexpr (even) prep_x(q, std::integral_constant<int, 1>{});
            else prep_x(q, std::integral_constant<int, 0>{});
            prep_w(step + 3);
            const int sw = step % 3, sx = q % 3;
            const uint32_t wa = lds0 + ((step + 1) % 3) * WSLOT + woff;
            const uint32_t xa = lds0 + (((step + 1) >> 1) % 3) * XSLOT + (even ? xoff1 : xoff0);     // tile step + 1 is the odd half in an even body
            static_for<0, NMMA>([&](auto I) {
                constexpr int i = decltype(I)::value, t = i / TM, j = i % TM;
                mfma16_asm(acc[t][j], wc[t], xc[j]);
                constexpr int rd = interleave_slot(i, NMMA, NRD);
                if constexpr (rd >= 0 && rd < TN) lds_read_frag<(rd < TN ? rd : 0) * 1024>(wnx[rd < TN ? rd : 0], wa);
                else if constexpr (rd >= TN) lds_read_frag<(rd >= TN ? rd - TN : 0) * 2048>(xnx[rd >= TN ? rd - TN : 0], xa);
                constexpr int pc = dma_slot(i, NMMA, NP);
                if constexpr (pc >= 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (pc < XPB)
                        __builtin_amdgcn_global_load_lds((const void*)px[pc < XPB ? pc : 0], (lds_ptr_t)dstX(sx, (even ? XPB : 0) + pc), 16, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((const void*)pw[pc >= XPB ? pc - XPB : 0], (lds_ptr_t)dstW(sw, pc - XPB), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            land(wnx, xnx);
        };
        for (int step = 0; step < nsteps; step += 2) {
            body(std::true_type{}, step, wfA, xfA, wfB, xfB);
            if (step + 1 < nsteps) body(std::false_type{}, step + 1, wfB, xfB, wfA, xfA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" : "+v"(acc[t][j]));
    } else
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step % NBUF;
        // tiles still allowed in flight behind tile `step`
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
    // epilogue with compile-time accumulator indices (a runtime-indexed acc[][] would be demoted to scratch)
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    const bool lds_epilogue_enabled = a.norm_eps != 54321.f;   // A/B hook (tuning only)
    if (ksplit) {   // partial sums: fp32, no bias / activation / residual (umv_qkv_post / umv_residual_rmsnorm_bf16 finish the row)
        e.out = reinterpret_cast<float*>(a.out) + (int64_t)blockIdx.y * a.split_stride;
        e.flags = UMV_EPI_OUT_F32;
    }
    // bf16 outputs leave through LDS as whole rows (gemm_epilogue.h); fp32 outputs (split-K partials, OUT_F32) directly
    constexpr bool LDS_EPI = BN * BM * 2 <= STAGE_BYTES;
    if (LDS_EPI && !(e.flags & UMV_EPI_OUT_F32) && lds_epilogue_enabled) {
        UMV_BARRIER();      // every wave has read its last fragments: the staging buffers are free
        // lean >= 0: the branch-free form of gemm_epilogue.h for this call's flag combination (epi_lean_kind); bit-identical
        if constexpr (!M32) {
            if (lean >= 0 && !ksplit &&
                epi_wave_tile_lean_any<TN, TM>(lean, a, acc, smem + wave * (TN * TM * 512), lane, m0 + wm * TM * 16, nt_base, bias_lds + wn * TN * 16))
                return;
        }
        epi_wave_tile_lds<TN, TM, M32>(e, acc, smem + wave * (TN * TM * 512), lane, m0 + wm * TM * 16, a.M, a.row_idx, nt_base, NTT,
                                       bias_lds + wn * TN * 16);
        return;
    }
    const bool swiglu = (a.epilogue & UMV_EPI_SWIGLU) != 0;
    if (swiglu) {
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int m = m0 + wm * TM * 16 + epi_row_of<M32>(j, lane);
            if (m < a.M) {
                const int64_t orow = a.row_idx ? (int64_t)a.row_idx[m] : (int64_t)m;
                static_for<0, TN / 2>([&](auto P) {
                    constexpr int p = decltype(P)::value;
                    const int ntile = nt_base + 2 * p;
                    if (ntile < NTT) {
                        const int c0 = (ntile >> 1) * 16 + epi_grp_of<M32>(j, lane) * 4;
                        float gg[4] = {acc[2 * p][j].x, acc[2 * p][j].y, acc[2 * p][j].z, acc[2 * p][j].w};
                        float uu[4] = {acc[2 * p + 1][j].x, acc[2 * p + 1][j].y, acc[2 * p + 1][j].z, acc[2 * p + 1][j].w};
                        epi_swiglu4(e, orow, c0, a.N / 2, gg, uu);
                    }
                });
            }
        });
        return;
    }
    // column groups outside, rows inside: the bias of a column group is loaded once (8 bytes), not once per row
    int64_t orow[TM];
    bool mok[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * TM * 16 + epi_row_of<M32>(j, lane);
        mok[j] = m < a.M;
        orow[j] = (mok[j] && a.row_idx) ? (int64_t)a.row_idx[m] : (int64_t)m;
    }
    static_for<0, TN>([&](auto T) {
        constexpr int t = decltype(T)::value;
        static_for<0, (M32 ? 2 : 1)>([&](auto GI) {       // the one or two column groups this lane meets in column tile t
            constexpr int gi = decltype(GI)::value;
            const int n0 = (nt_base + t) * 16 + epi_grp_of<M32>(gi, lane) * 4;
            if (n0 < a.N) {
                float b4[4] = {0.f, 0.f, 0.f, 0.f};
                if (e.flags & UMV_EPI_BIAS) epi_bias4(e, n0, b4);
                static_for<0, TM>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    if constexpr (!M32 || (j & 1) == gi)
                        if (mok[j]) epi_store4(e, orow[j], n0, acc[t][j].x, acc[t][j].y, acc[t][j].z, acc[t][j].w, nullptr, b4);
                });
            }
        });
    });
}

static int raster_gn() {   // n-blocks per strip of the tile order; UMV_GEMM_RASTER overrides (tuning only)
    static const int gn = umv_env_int("UMV_GEMM_RASTER", 4) < 1 ? 1 : umv_env_int("UMV_GEMM_RASTER", 4);
    return gn;
}

// the epilogue form of a tiled call: >= 0 = gemm_epilogue.h's lean form for this flag combination, -1 = the general one
// (UMV_GEMM_LEAN_EPI=0: always the general one - A/B, tuning only; results are bit-identical)
int umv_gemm_lean_epilogue(const umv_gemm_args& a) {
    static const int on = umv_env_int("UMV_GEMM_LEAN_EPI", 1);
    return on ? epi_lean_kind(a) : -1;
}

template <int WN, int WM, int TN, int TM, int KTS, int NBUF, int SCHED = 0>
static int launch_tiled(const umv_gemm_args& a, int KT, int NTT, hipStream_t s) {
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int NT_ALL = BN / 16 * KTS + BM / 16 * KTS, NWV = WN * WM;
    constexpr size_t stage_bytes = (SCHED & 15) == 3 ? (size_t)3 * (BN / 16) * 1024 + (size_t)3 * BM * 128 : (size_t)NBUF * NT_ALL * 1024;
    constexpr size_t ndummy = (SCHED & 15) == 3 ? (size_t)((BN / 16 + NWV - 1) / NWV * NWV - BN / 16) : (size_t)((NT_ALL + NWV - 1) / NWV * NWV - NT_ALL);
    constexpr size_t lds = stage_bytes + (BN * 2 + 15) / 16 * 16                        // staging buffers + the tile's bias
                           + ndummy * 1024;                                            // + a spare KiB per surplus staging slot
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tiled_kernel<WN, WM, TN, TM, KTS, NBUF, SCHED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int mblocks = (a.M + BM - 1) / BM, nblocks = (a.N + BN - 1) / BN;
    const int splits = a.k_splits > 1 ? a.k_splits : 1;
    const int ksplit = splits > 1 ? ((KT + splits - 1) / splits + KTS - 1) / KTS * KTS : 0;    // whole k-steps per split
    // m-blocks per super-block: ~64 MB of x rows (UMV_GEMM_MSB overrides, tuning only; 0 = one super-block)
    static const int msb_env = umv_env_int("UMV_GEMM_MSB", -2);
    int ms = (int)(((int64_t)64 << 20) / ((int64_t)BM * a.K * 2));
    ms = ms < 8 ? 8 : ms;
    if ((int64_t)mblocks * BM < 16384) ms = mblocks;   // measured (us, on / off): M = 32 832 gate/up 7040 / 7480, down 3880 / 4070, qkv 1013 / 1056;
                                                       // M = 16 416 down 1975 / 2010, gate/up 3543 / 3528; M = 8208 down 1062 / 1047: off below 16k rows
    if (msb_env >= 0) ms = msb_env == 0 ? mblocks : msb_env;
    if (ms > mblocks || ms * 3 / 2 >= mblocks) ms = mblocks;        // a short second super-block is not worth a second pass over W
    else ms = (mblocks + (mblocks + ms - 1) / ms - 1) / ((mblocks + ms - 1) / ms);   // equal super-blocks: no stub at the end
    hipLaunchKernelGGL((gemm_tiled_kernel<WN, WM, TN, TM, KTS, NBUF, SCHED>), dim3(mblocks * nblocks, splits), dim3(WN * WM * 64), lds, s, a,
                       KT, NTT, mblocks, nblocks, raster_gn(), ksplit, ms, umv_gemm_lean_epilogue(a));
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// full-line x staging of the weight-streaming kernels: 2 (default) = always, 1 = above 8 rows only, 0 = never (UMV_SKINNY_XL, A/B only).
// Measured on MI355X (tools/skinny_bench.py, us, plain -> XL; profiles/r04_skinny_xl.txt): 32 rows qkv 21.3 -> 16.4, o 13.0 -> 9.9,
// gate/up 59.3 -> 53.3, down 51.0 -> 34.8 (configs[3] decode step 4.445 -> 4.099 ms, 7199 -> 7807 tokens/s); 16 rows 14.3 -> 11.9 /
// 9.0 -> 8.1 / 48.5 -> 47.2 / 33.5 -> 27.9.  At 8 rows the two-piece form costs gate/up its second resident workgroup (125 -> 142
// registers: 42.4 -> 45.4 us), the ONE-piece form (rows 0..7 only, 124 registers) wins: gate/up 42.5 -> 41.4 (6.57 TB/s), down
// 26.0 -> 24.7, qkv 11.4 -> 9.2, headline step 3.161 -> 3.111 ms.
static int skinny_xl() {
    static const int v = umv_env_int("UMV_SKINNY_XL", 2);
    return v;
}

template <int MB, int NT, int U, bool DB, int NORM>
static int launch_skinny(const umv_gemm_args& a, int KT, int NTT, hipStream_t s) {
    int blocks = (NTT + NT - 1) / NT;
    size_t lds = (size_t)SK_WAVES * NT * MB * 4 * 64 * sizeof(float);
    if (NORM) lds += SK_WAVES * 16 * sizeof(float) + (size_t)KT * 4 * NORM * 16;
    const int nsplit = a.k_splits > 1 ? a.k_splits : 1;
    if constexpr (NORM == 0 && U % 2 == 0) {
        // (rows that start on a 128-byte boundary, so that a k-tile pair is one cache line; any K, any split)
        if ((skinny_xl() > 1 || (skinny_xl() == 1 && a.M > 8)) && (a.ldx % 64) == 0 && ((uintptr_t)a.x % 128) == 0) {
            auto go = [&](auto XLV) {
                constexpr int XL = decltype(XLV)::value;
                size_t l2 = lds;
                const size_t xl = (size_t)SK_WAVES * (U / 2) * (XL == 1 ? 1 : 2 * MB) * 1024;
                if (xl > l2) l2 = xl;
                static bool attr_set[UMV_MAX_DEVICES] = {};
                if (l2 > 64 * 1024 && umv_first_on_device(attr_set))
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<MB, NT, U, DB, NORM, XL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
                hipLaunchKernelGGL((gemm_skinny_kernel<MB, NT, U, DB, NORM, XL>), dim3(blocks, nsplit), dim3(SK_WAVES * 64), l2, s, a, KT, NTT);
            };
            if constexpr (MB == 1) {
                if (a.M <= 8) go(std::integral_constant<int, 1>{});
                else go(std::integral_constant<int, 2>{});
            } else {
                go(std::integral_constant<int, 2>{});
            }
            UMV_LAUNCH_CHECK();
            return UMV_OK;
        }
    }
    hipLaunchKernelGGL((gemm_skinny_kernel<MB, NT, U, DB, NORM>), dim3(blocks, nsplit), dim3(SK_WAVES * 64), lds, s,
                       a, KT, NTT);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// Tile choice from measurements on MI355X (tools/gemm_bench.py, profiles/r01_gemm_tiles_auto.txt).  The 256x256x32
// 4-buffer tile with the interleaved schedule wins whenever it yields >= ~144 workgroups (885-1120 TF/s on the
// prefill / flow / ViT shapes); below that the 256(n) x 128(m) interleaved tile (M ~ 2048: 920-1020 TF/s), then
// 128 x 128 with two workgroups per CU (M ~ 1024: 560-680), then 128(n) x 64(m).
// UMV_GEMM_TILE=<256|266|258|268|384|288|129|130|270|64> overrides (tuning only).
// Exported so that tests can assert which kernel a shape is sent to (returns 0 for M <= 64: weight-streaming kernels).
// compute units of the current device (256 on MI355X; 256 as well when no device answers - the policy is also queried on hosts without one)
static int umv_cu_count() {
    static const int n = [] {
        int d = 0, v = 0;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) return 256;
        return v;
    }();
    return n;
}

extern "C" int umv_gemm_tile_config(int M, int N, int K) {
    static const int force = umv_env_int("UMV_GEMM_TILE", 0);
    if (force) return force;
    if (M <= 64) return 0;
    const long wg256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    const long wg128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long wg258 = (long)((M + 127) / 128) * ((N + 255) / 256);
    if (K < 1024) return 64;              // short K: the 4-buffer prologue does not amortise
    if (wg256 >= 144 && (N <= 8192 || M <= 1024)) {   // (wide N: many n-blocks either way, 256 x 256 wins or ties - 2064 x 37888: 602 vs 606 us -
                                                      // except at few rows: a single image's guided flow pass, 512 x 37888 x 3584, is 296 tiles = two
                                                      // rounds of 256 x 256 against two rounds of the smaller 384 x 128: 231 -> 206 us)
        // N = 1152 (SigLIP out / fc2) is 4.5 tiles of 256: 10 % padding and 160 tiles for 256 CUs.  As 3 x 384 columns by 128 rows
        // it is 192 tiles of 3/4 the work: rounds x tile area decides (out 46.7 -> 37.3 us, fc2 103.8 -> 89.9, fc1 (N = 4304: 544
        // tiles = 2.1 rounds against 768 = 3 rounds of 3/4) 120.3 -> 111.7; the fused q/k/v GEMM, N = 3456, stays on
        // 256 x 256: 76 vs 95 us)
        const long cus = umv_cu_count(), t384 = (long)((M + 127) / 128) * ((N + 383) / 384);
        const long c266 = (wg256 + cus - 1) / cus * 65536, c384 = (t384 + cus - 1) / cus * 49152;
        // 288 x 128 (round 3): N = 1152 / 4608 = 4 / 16 x 288 columns give exactly 256 tiles at 8192 / 2048 rows where 384 x 128 fills
        // 192 of the 256 CUs.  Its 18 MFMAs per k-step carry the same per-step overhead as the bigger tiles' 24 - 32 (about 0.8 of
        // their rate per unit of tile area), so it has to win by more than that: out-proj 38.4 -> 34.8 us, fc2 91.9 -> 85.4, the
        // flow passes' q/k/v GEMM (2048 x 4608 x 3584) 80.2 -> 73.8
        if (N % 288 == 0) {
            const long t288 = (long)((M + 127) / 128) * (N / 288), c288 = (t288 + cus - 1) / cus * 36864;
            if (c288 * 6 < c384 * 5 && c288 * 6 < c266 * 5) return 288;
        }
        // few rows on a wide N (the 8 x 34-token question prefill: 272 x 37888 is 297 tiles of 384 x 128 = two rounds for 1.16 rounds of
        // work): 256 x 128 tiles at ~1.15x the cost per unit of area (444 tiles = two rounds of two thirds the size): 154 -> 134 us
        // (profiles/r04_m272_tiles.txt); the same rule sends the 65..128-row decode gate/up GEMM to 148 tiles of 256 x 128 instead of 99 of
        // 384 x 128: 128 samples 7.80 -> 7.47 ms per step, 96: 7.07 -> 6.76, 72: 6.50 -> 6.17 (profiles/r04_fewrow_tile.txt)
        const long c268 = (wg258 + cus - 1) / cus * 32768 * 115 / 100;
        static const int fewrow = umv_env_int("UMV_GEMM_FEWROW", 1);      // UMV_GEMM_FEWROW=0: without this rule (A/B, tuning only)
        if (fewrow && M <= 512 && c268 < c384 && c268 < c266) return 268;
        if (c384 * 10 < c266 * 9) return 384;
    }
    if (wg256 >= 144) return 266;
    if (wg258 >= 140) return 268;
    if (wg128 >= 128) return 270;
    return 64;
}

extern "C" int umv_gemm_bf16(const umv_gemm_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap != nullptr, UMV_ERR_ARG, "gemm: null args");
    umv_gemm_args a = *ap;
    UMV_CHECK(a.x && a.wp && a.out, UMV_ERR_ARG, "gemm: null pointer");
    UMV_CHECK(a.M >= 0 && a.N > 0 && a.K > 0, UMV_ERR_ARG, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    UMV_CHECK((a.K % 8) == 0 && (a.ldx % 8) == 0, UMV_ERR_ARG, "gemm: K (%d) and ldx (%lld) must be multiples of 8", a.K,
              (long long)a.ldx);
    UMV_CHECK(!(a.epilogue & UMV_EPI_BIAS) || a.bias, UMV_ERR_ARG, "gemm: BIAS without bias pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_RESIDUAL) || a.residual, UMV_ERR_ARG, "gemm: RESIDUAL without residual pointer");
    UMV_CHECK(!(a.epilogue & UMV_EPI_SWIGLU) || (a.N % 32) == 0, UMV_ERR_ARG, "gemm: SWIGLU needs N %% 32 == 0");
    UMV_CHECK(!a.norm_w || (a.M <= 16 && a.K <= SK_WAVES * SK_XMAX * 32), UMV_ERR_UNSUPPORTED,
              "gemm: fused RMSNorm needs M <= 16 and K <= %d (got M=%d K=%d)", SK_WAVES * SK_XMAX * 32, a.M, a.K);
    UMV_CHECK(a.k_splits <= 1 || (a.M <= 128 && !a.norm_w && !(a.epilogue & UMV_EPI_SWIGLU) && a.tile_rows % 16 == 0 && a.split_stride > 0),
              UMV_ERR_UNSUPPORTED, "gemm: split-K (k_splits=%d) is a decode mode: M <= 128, 16-row image, no SwiGLU / fused norm, "
              "split_stride > 0", a.k_splits);
    UMV_CHECK(a.k_splits <= 64, UMV_ERR_ARG, "gemm: k_splits %d > 64", a.k_splits);
    UMV_CHECK(!a.argmax_partial || (a.M <= 64 && a.k_splits <= 1 && !a.row_idx && !a.norm_w && (a.tile_rows == 0 || a.tile_rows == 16) &&
                                    !(a.epilogue & (UMV_EPI_SWIGLU | UMV_EPI_OUT_F32))),
              UMV_ERR_UNSUPPORTED, "gemm: argmax_partial is an epilogue of the decode lm_head GEMM (M <= 64, 16-row image, bf16 out, "
              "no SwiGLU / split-K / row_idx / fused norm)");
    UMV_CHECK(a.sample_temperature >= 0.f && (a.sample_temperature == 0.f || a.argmax_partial), UMV_ERR_ARG,
              "gemm: sample_temperature (%g) is a mode of the argmax_partial epilogue and must be >= 0", (double)a.sample_temperature);
    if (a.M == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    const int KT = (a.K + 31) / 32;
    const int TH = a.tile_rows > 0 ? a.tile_rows : 16;
    UMV_CHECK(TH <= 16, UMV_ERR_ARG, "gemm: tile_rows %d > 16", TH);
    UMV_CHECK(TH == 16 || (a.M <= 64 && !(a.epilogue & UMV_EPI_SWIGLU)), UMV_ERR_UNSUPPORTED,
              "gemm: %d-row packed tiles are a decode-only layout (M <= 64, no SwiGLU)", TH);
    const int NTT = (a.N + TH - 1) / TH;
    static const int skinny_max = umv_env_int("UMV_GEMM_SKINNY_MAX", 64);   // tuning only: UMV_GEMM_SKINNY_MAX=<M> (rows up to which the weight-streaming kernel is used)
    static const int sk_tiled_min = umv_env_int("UMV_SPLITK_TILED_MIN", 65);   // tuning only: UMV_SPLITK_TILED_MIN=<rows from which split-K runs on the tiled kernel>
    static const int sk_xl = umv_env_int("UMV_SPLITK_TILED_XL", 1);          // UMV_SPLITK_TILED_XL=0: the half-line staging of the 65..128-row split-K tile (A/B, tuning only; bit-identical;
    if (a.k_splits > 1 && (a.M > 64 || a.M >= sk_tiled_min) && TH == 16) {  // 65..128 rows: the 128 x 128 tile (two workgroups per CU) over k_splits K ranges, fp32 partials
        if (sk_xl) return launch_tiled<2, 2, 4, 4, 1, 4, 3>(a, KT, NTT, s);
        return launch_tiled<2, 2, 4, 4, 1, 4, 1>(a, KT, NTT, s);
    }
    if (a.k_splits > 1) {
        // split-K decode GEMM: 4 n-tiles per workgroup share every x fragment (x re-reads from L2 drop 4x against the
        // one-tile workgroups), the K range is cut k_splits ways to keep >= 256 workgroups, partial sums go to fp32
        // (two n-tiles per workgroup at M <= 8 - twice the workgroups, 216-224 is less than one per CU - measured 3.199 vs
        // 3.176 ms per step: no)
        if (a.M <= 16) return launch_skinny<1, 4, 2, true, 0>(a, KT, NTT, s);
        if (a.M <= 32) {
            static const int v32 = umv_env_int("UMV_SKINNY_M32", 0);    // tuning only: UMV_SKINNY_M32=<0|1|2>
            if (v32 == 1) return launch_skinny<2, 4, 1, true, 0>(a, KT, NTT, s);
            if (v32 == 2) return launch_skinny<2, 2, 2, true, 0>(a, KT, NTT, s);
            return launch_skinny<2, 4, 2, true, 0>(a, KT, NTT, s);
        }
        // (33..64 rows on a tiled kernel over the K ranges - 128 x 64 full-line or UMV_SPLITK_TILED_MIN=33 - measured 6.25 / 6.45 ms per
        // 64-sample step against 5.41: the weight-streaming kernel stays)
        return launch_skinny<4, 4, 1, true, 0>(a, KT, NTT, s);
    }
    if (a.M <= 64 && (a.M <= skinny_max || TH != 16)) {
        const bool two = (a.epilogue & UMV_EPI_SWIGLU) || NTT >= 1024;
        if (a.M <= 16) {
            if (a.norm_w && a.M <= 8) return two ? launch_skinny<1, 2, 4, true, 8>(a, KT, NTT, s) : launch_skinny<1, 1, 8, true, 8>(a, KT, NTT, s);
            if (a.norm_w) return two ? launch_skinny<1, 2, 4, true, 16>(a, KT, NTT, s) : launch_skinny<1, 1, 8, true, 16>(a, KT, NTT, s);
            return two ? launch_skinny<1, 2, 4, true, 0>(a, KT, NTT, s) : launch_skinny<1, 1, 8, true, 0>(a, KT, NTT, s);
        }
        // M > 16: every workgroup re-reads all of x from L2, so wide-N GEMMs take 4 n-tiles per workgroup (x : weight bytes
        // = M : 64); UMV_GEMM_SKINNY_NT=2 restores the 2-tile kernels (tuning only)
        static const int nt4_env = umv_env_int("UMV_GEMM_SKINNY_NT", 4), nt4 = nt4_env == 2 ? 0 : nt4_env == 8 ? 8 : 1;
        // (the 128 x 64 full-line tile at 17..32 rows: 32 samples 3.99 -> 4.09 ms per step, 24: 3.80 -> 3.92, 17: 3.62 -> 3.74 - the
        // weight-streaming kernel keeps these rows; profiles/r04_midbatch_xline.txt)
        if (two && nt4 == 8 && TH == 16 && a.M <= 32) return launch_skinny<2, 8, 1, true, 0>(a, KT, NTT, s);
        if (two && nt4 && TH == 16 && a.M <= 32) {
            static const int v32 = umv_env_int("UMV_SKINNY_M32", 0);
            if (v32 == 1) return launch_skinny<2, 4, 1, true, 0>(a, KT, NTT, s);
            if (v32 == 2) return launch_skinny<2, 2, 2, true, 0>(a, KT, NTT, s);
            return launch_skinny<2, 4, 2, true, 0>(a, KT, NTT, s);
        }
        // 33..64 rows on the wide-N GEMMs: the LDS-staged tile 128(n) x 64(m) x 64 reads x once per 128 columns instead of once
        // per 64 and streams gate/up in 73 us against 84 for the 4-tile skinny kernel (tools/stream_tile_bench.py 64); at
        // <= 32 rows the skinny kernel wins (60 vs 64-66 us).  Not with the argmax epilogue (skinny kernels only).
        static const int m64 = umv_env_int("UMV_GEMM_M64_TILED", 2);
        // UMV_GEMM_M64_TILED=2 (default): the same tile with k-steps of 32 and x staged in full 128-byte lines (SCHED = 3, 48 KiB, 3 WG/CU):
        // bit-identical, 64-sample decode step 5.65 -> 5.42 ms, 40 samples 4.82 -> 4.60 ms; 1 = the 64-wide k-step tile, 0 = skinny
        if (two && TH == 16 && m64 == 2 && a.M > 32 && !a.argmax_partial && !a.norm_w && a.K >= 1024) return launch_tiled<4, 1, 2, 4, 1, 4, 3>(a, KT, NTT, s);
        if (two && TH == 16 && m64 && a.M > 32 && !a.argmax_partial && !a.norm_w && a.K >= 1024) return launch_tiled<4, 1, 2, 4, 2, 3>(a, KT, NTT, s);
        if (two && nt4 && TH == 16) return launch_skinny<4, 4, 1, true, 0>(a, KT, NTT, s);
        if (a.M <= 32) return two ? launch_skinny<2, 2, 4, false, 0>(a, KT, NTT, s) : launch_skinny<2, 1, 4, false, 0>(a, KT, NTT, s);
        return two ? launch_skinny<4, 2, 2, false, 0>(a, KT, NTT, s) : launch_skinny<4, 1, 4, false, 0>(a, KT, NTT, s);
    }
    int cfg = umv_gemm_tile_config(a.M, a.N, a.K);
    // SwiGLU pairs the gate / up tiles (2p, 2p + 1) INSIDE a wave's column range: the 288-column tile gives a wave 9 tiles, its pairs
    // would straddle two waves.  (The policy can pick 288 for N = 2 I = k * 288; the model's 37888 is not one of those.)
    if ((a.epilogue & UMV_EPI_SWIGLU) && cfg == 288) cfg = 384;
    {   // the staging variant of the interleaved tiles: full-line x staging (SCHED = 3) unless UMV_GEMM_XLINE=0 (A/B, tuning only).
        // Bit-identical results; end to end on MI355X (tools/stage_profile.py, same box): text-to-image 1530 -> 1443 ms per batch of 4,
        // prefill of 8 images 131.7 -> 130.7 ms, ViT tower 12.70 -> 12.50 ms.  (A 20-launch microbenchmark from a cold chip shows the
        // opposite sign, -2..-8 %: the variant pays a longer prologue and wins only at the clocks a sustained load runs at.)
        static const int xline = umv_env_int("UMV_GEMM_XLINE", 1);      // (2: also the 288-column tile, under evaluation)
        if (xline) cfg = cfg == 266 ? 366 : cfg == 268 ? 368 : cfg == 384 ? 484 : cfg == 270 ? 370 : cfg;
#ifdef UMV_GEMM_ABLATIONS
        if (xline > 1 && cfg == 288) cfg = 388;      // the 288-column tile with full-line staging: measured, not adopted (ablation builds only)
#endif
    }
    {   // round 5: the 4-wave tiles with the accumulators in AGPRs (gemm_w4.hip) take over the 8-wave tiles of the same shape: bit-identical
        // results (same MFMAs, operands and k order).  Sustained loops on MI355X (profiles/r05_w4_policy.txt): gate/up 2048 x 37888 x 3584
        // 458 -> 408 us (1.21 -> 1.36 PF), 8208 rows 1790 -> 1602 us (1.39 PF), down 2048 x 3584 x 18944 254 -> 205 us (1.36 PF), o_proj
        // 8208 x 3584 x 3584 182 -> 169 us; end to end text-to-image 1387 -> 1315 ms per batch of 4, prefill of 8 images 123.1 -> 116.5 ms.
        // At short K (SigLIP, K = 1152) they first LOST - q/k/v 67.8 -> 80.6 us, tower 11.6 -> 12.3 ms - because the general epilogue's
        // 35 k cycles per tile ran on four waves instead of eight; with the lean epilogue (gemm_epilogue.h) they win there too: q/k/v
        // 64.5 -> 59.9 us, fc1 84.9 -> 80.0 us, tower 10.66 -> 10.50 ms at 8 images and 38.0 -> 37.3 ms at 32 (20-repetition runs, twice;
        // profiles/r05_lean_epilogue.txt).  So: every K.  UMV_GEMM_W4=0: the 8-wave tiles everywhere, 3: the 4-wave tiles only at
        // K >= 2048 (the rule before the lean epilogue) - A/B, tuning only.
        static const int w4 = umv_env_int("UMV_GEMM_W4", 1);
        const int c4 = cfg == 366 ? 466 : cfg == 368 ? 468 : cfg == 484 ? 4384 : 0;
        if (w4 && c4 && (w4 != 3 || a.K >= 2048) && umv_gemm_w4_can_take(a, KT, NTT)) return umv_gemm_w4_launch(a, KT, NTT, c4, raster_gn(), s);
    }
    if (cfg == 466 || cfg == 468 || cfg == 4384 || cfg == 94661 || cfg == 94662) return umv_gemm_w4_launch(a, KT, NTT, cfg, raster_gn(), s);
    // experimental weight-streaming shapes of the tiled kernel for 16 < M <= 128 (tuning only, UMV_GEMM_TILE + UMV_GEMM_SKINNY_MAX)
    if (cfg == 332) return launch_tiled<4, 1, 2, 2, 4, 3>(a, KT, NTT, s);      // 128(n) x 32(m) x 128, 3 buffers (120 KiB), 4 waves
    if (cfg == 333) return launch_tiled<4, 1, 2, 2, 2, 4>(a, KT, NTT, s);      // 128(n) x 32(m) x 64, 4 buffers (80 KiB)
    if (cfg == 335) return launch_tiled<4, 1, 2, 2, 2, 3>(a, KT, NTT, s);      // 128(n) x 32(m) x 64, 3 buffers (60 KiB, 2 WG/CU)
    if (cfg == 364) return launch_tiled<4, 1, 2, 4, 2, 3>(a, KT, NTT, s);      // 128(n) x 64(m) x 64, 3 buffers (72 KiB)
    if (cfg == 3128) return launch_tiled<4, 1, 2, 8, 2, 3>(a, KT, NTT, s);     // 128(n) x 128(m) x 64, 3 buffers (96 KiB)
    if (cfg == 256) return launch_tiled<2, 4, 8, 4, 1, 4>(a, KT, NTT, s);      // 256x256x32, 4 buffers (128 KiB)
    if (cfg == 129) return launch_tiled<2, 2, 4, 4, 2, 2>(a, KT, NTT, s);      // 128x128x64, 2 buffers (64 KiB, 2 WG/CU)
    if (cfg == 130) return launch_tiled<2, 2, 4, 4, 1, 4>(a, KT, NTT, s);      // 128x128x32, 4 buffers (64 KiB, 2 WG/CU)
    if (cfg == 288) return launch_tiled<2, 4, 9, 2, 1, 4, 1>(a, KT, NTT, s);   // 288(n)x128(m)x32: N = 1152 / 4608 = 4 / 16 x 288 -> 256 tiles at 8192 / 2048 rows
    if (cfg == 266) return launch_tiled<2, 4, 8, 4, 1, 4, 1>(a, KT, NTT, s);   // 256x256x32, 4 buffers, MFMA / ds_read interleaved by hand
#ifdef UMV_GEMM_ABLATIONS     // measured and not adopted / timing-only variants (UMV_GEMM_ABLATIONS=1 python -m unimedvl_amd.build; profiles/HISTORY.md section 5b)
    if (cfg == 9661) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 1>(a, KT, NTT, s);   // ablations of 266 (timing only)
    if (cfg == 9662) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 2>(a, KT, NTT, s);
    if (cfg == 9663) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 3>(a, KT, NTT, s);
    if (cfg == 9664) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 4>(a, KT, NTT, s);
    if (cfg == 9665) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 5>(a, KT, NTT, s);
    if (cfg == 9666) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 6>(a, KT, NTT, s);
    if (cfg == 9667) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 7>(a, KT, NTT, s);
    if (cfg == 9668) return launch_tiled<2, 4, 8, 4, 1, 4, 1 + 16 * 8>(a, KT, NTT, s);
    if (cfg == 566) return launch_tiled<2, 4, 8, 4, 1, 4, 2>(a, KT, NTT, s);   // the same tiles on v_mfma_f32_32x32x16_bf16 (SCHED = 2)
    if (cfg == 568) return launch_tiled<4, 2, 4, 4, 1, 4, 2>(a, KT, NTT, s);
    if (cfg == 684) return launch_tiled<4, 2, 6, 4, 1, 4, 2>(a, KT, NTT, s);
    if (cfg == 570) return launch_tiled<2, 2, 4, 4, 1, 4, 2>(a, KT, NTT, s);
#endif
    if (cfg == 366) return launch_tiled<2, 4, 8, 4, 1, 4, 3>(a, KT, NTT, s);   // 266 / 268 / 384 / 270 with the x operand staged in full 128-byte lines (SCHED = 3)
    if (cfg == 368) return launch_tiled<4, 2, 4, 4, 1, 4, 3>(a, KT, NTT, s);
    if (cfg == 484) return launch_tiled<4, 2, 6, 4, 1, 4, 3>(a, KT, NTT, s);
    if (cfg == 370) return launch_tiled<2, 2, 4, 4, 1, 4, 3>(a, KT, NTT, s);
#ifdef UMV_GEMM_ABLATIONS
    if (cfg == 388) return launch_tiled<2, 4, 9, 2, 1, 4, 3>(a, KT, NTT, s);   // 288(n) x 128(m) with full-line x staging (18 W tiles on 8 waves: 3 slots each, 6 of them idle)
#endif
    if (cfg == 268) return launch_tiled<4, 2, 4, 4, 1, 4, 1>(a, KT, NTT, s);   // 256(n)x128(m)x32, 8 waves as 4x2, interleaved
    if (cfg == 384) return launch_tiled<4, 2, 6, 4, 1, 4, 1>(a, KT, NTT, s);   // 384(n)x128(m)x32, 8 waves of 96 x 64: N = 1152 = 3 x 384 without padding
    if (cfg == 270) return launch_tiled<2, 2, 4, 4, 1, 4, 1>(a, KT, NTT, s);   // 128x128x32, 4 waves, 4 buffers (64 KiB, 2 WG/CU), interleaved
    if (cfg == 258) return launch_tiled<4, 2, 4, 4, 1, 4>(a, KT, NTT, s);      // 256(n)x128(m)x32, 8 waves as 4x2 (96 KiB)
    return launch_tiled<2, 2, 4, 2, 2, 3>(a, KT, NTT, s);                      // 128(n) x 64(m) x 64, 3 buffers (72 KiB)
}
