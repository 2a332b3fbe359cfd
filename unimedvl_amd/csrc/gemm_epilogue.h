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
    // torch gelu(approximate="tanh"): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3).  With tanh(u) = 1 - 2/(e^(2u)+1) this is
    // x / (1 + e^(-2u)) = x * rcp(1 + 2^(x*(c0 + c1*x^2))): 7 VALU ops, two of them v_exp_f32 / v_rcp_f32 (libm's tanhf is ~40;
    // the textbook form 15).  Abs error ~1e-7, far below the bf16 rounding that follows.  The exponent is clamped so that
    // 2^(..) stays finite: x -> -inf gives -0 like the reference, never inf * 0.
    const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f;   // -2 log2(e) sqrt(2/pi)
    const float c1 = c0 * 0.044715f;
    const float p = __builtin_fmaf(c1, x * x, c0);
    const float e = __builtin_amdgcn_exp2f(fminf(x * p, 126.0f));
    return x * __builtin_amdgcn_rcpf(1.0f + e);
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
// Sampling as an argmax (Gumbel-max): with temp > 0 the value whose key is taken is bf16(logit / temp) - ln(-ln(u)), u in (0, 1] from
// splitmix64(row key + column) - argmax over a row = one draw from softmax(logits / temp) (bagel.py:1297-1299).  The generator and its
// keying are umv_sample_bf16's (elementwise.hip).  u = (r + 0.5) 2^-23 with r the top 23 bits of the hash: strictly inside (0, 1) and exact in
// fp32 (r + 0.5 needs 24 bits), so the noise stays finite (-ln(-ln u) in [-2.8, 16.6]) and a column can only win through its logit - with
// u = 1 allowed (round 5) a column won with probability 2^-24 whatever its logit: ~1 % of the draws over a 152 k vocabulary.
__device__ __forceinline__ uint64_t epi_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t epi_sample_row_key(uint64_t seed, const int64_t* step_ptr, int m) {
    const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : 0ull;
    return epi_splitmix64(seed ^ (step * 0xD1B54A32D192ED03ull) ^ ((uint64_t)m << 32));
}
__device__ __forceinline__ float epi_gumbel_value(float logit, float temp, uint64_t row_key, int n) {
    const float y = rbf(logit / temp);                                    // logits / temperature is a bf16 tensor in the reference
    const uint64_t h = epi_splitmix64(row_key + (uint64_t)n);
    const float u = ((float)(h >> 41) + 0.5f) * (1.0f / 8388608.0f);      // [2^-24, 1 - 2^-24]
    const float q = fmaxf(-__logf(u), 5.9604645e-8f);                     // Exp(1) draw, held at -ln(1 - 2^-24) whatever the fast log returns next to 1
    return y - __logf(q);
}
__device__ __forceinline__ void epi_argmax_tile(uint64_t* __restrict__ partial, int64_t ld_partial, int m, int tile, int lane, bool valid,
                                                int n0, int nend, const float* final, float temp = 0.f, uint64_t seed = 0, const int64_t* step_ptr = nullptr) {
    uint64_t key = 0;
    if (valid) {
        const uint64_t row_key = temp > 0.f ? epi_sample_row_key(seed, step_ptr, m) : 0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + j < nend) {
                const float v = temp > 0.f ? epi_gumbel_value(final[j], temp, row_key, n0 + j) : final[j];
                const uint64_t kj = argmax_key(v, n0 + j);
                key = kj > key ? kj : key;
            }
    }
    uint64_t o = shfl_xor_u64(key, 16);
    key = o > key ? o : key;
    o = shfl_xor_u64(key, 32);
    key = o > key ? o : key;
    if (valid && (lane >> 4) == 0) partial[(int64_t)m * ld_partial + tile] = key;
}

// The bias of columns n0..n0+3 (zero beyond N): one 8-byte load when the four columns exist and the address is 8-byte aligned.
__device__ __forceinline__ void epi_bias4(const EpiCtx& e, int n0, float* b) {
    const bf16_t* bp = e.bias + n0;
    if (n0 + 3 < e.N && ((reinterpret_cast<uintptr_t>(bp) & 7) == 0)) {
        const u32x2 pk = *reinterpret_cast<const u32x2*>(bp);
        b[0] = __uint_as_float(pk.x << 16); b[1] = __uint_as_float(pk.x & 0xFFFF0000u);
        b[2] = __uint_as_float(pk.y << 16); b[3] = __uint_as_float(pk.y & 0xFFFF0000u);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (n0 + j < e.N) ? bf2f(bp[j]) : 0.f;
    }
}

// Finish 4 consecutive n (n0..n0+3) of row `orow` from fp32 accumulators; `bias4` = epi_bias4's values when the caller has
// them already (the tiled kernel loads them once per column group, not once per row).  `final` (optional) receives the
// stored values.
__device__ __forceinline__ void epi_store4(const EpiCtx& e, int64_t orow, int n0, float v0, float v1, float v2, float v3,
                                           float* final = nullptr, const float* bias4 = nullptr) {
    float v[4] = {v0, v1, v2, v3};
    if (e.flags & UMV_EPI_BIAS) {
        float bl[4];
        if (!bias4) epi_bias4(e, n0, bl);
        const float* b = bias4 ? bias4 : bl;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b[j];      // columns beyond N carry a zero bias and are never stored
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
        if (n0 + 3 < e.N && ((reinterpret_cast<uintptr_t>(rr) & 7) == 0)) {   // one 8-byte load
            const u32x2 pk = *reinterpret_cast<const u32x2*>(rr);
            v[0] = rbf(v[0] + __uint_as_float(pk.x << 16)); v[1] = rbf(v[1] + __uint_as_float(pk.x & 0xFFFF0000u));
            v[2] = rbf(v[2] + __uint_as_float(pk.y << 16)); v[3] = rbf(v[3] + __uint_as_float(pk.y & 0xFFFF0000u));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n0 + j < e.N) v[j] = rbf(v[j] + bf2f(rr[j]));
        }
    }
    if (final) {
#pragma unroll
        for (int j = 0; j < 4; ++j) final[j] = v[j];     // already bf16-exact (every branch above ends in rbf)
    }
    bf16_t* o = reinterpret_cast<bf16_t*>(e.out) + orow * e.ldo + n0;
    if (n0 + 3 < e.N && ((reinterpret_cast<uintptr_t>(o) & 7) == 0)) {   // 8-byte aligned: one packed store
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

// ----------------------------------------------------------------------------- wave-tile epilogue through LDS
// The MFMA accumulator layout gives a lane 4 consecutive columns of one row: stored directly, a wave's store instruction
// covers 16 rows x 32 bytes - 16 partial cache lines, and TN*TM such instructions per wave (the store tail of a 256 x 256
// tile measured 15-31 us of an 87-118 us ViT GEMM, 14 % of the 8208 x 18944 x 3584 prefill GEMM).  Here every wave
// transposes its own (TM*16) x (TN*16) tile through its own LDS region (the staging buffers are free after the main loop;
// no cross-wave traffic, so no barrier beyond the caller's "everyone left the main loop"):
//   phase 1  accumulator layout: (* scales) + bias -> bf16 -> activation -> bf16 (or SwiGLU of the gate / up tile pair),
//            ds_write_b64 at [row][16-byte chunk ^ (row & (CH-1))] (XOR swizzle: at most 2-way bank conflicts, no padding);
//   phase 2  row layout: ds_read_b128 (conflict free), + residual with a 16-byte load, one 16-byte store per lane - a wave
//            instruction covers 64/CH whole rows of the wave tile (256 contiguous bytes each at TN = 8).
// Arithmetic and rounding points are those of epi_store4 / epi_swiglu4: results are bit-identical to the direct path.
// sw (per column) / sx (per row), optional: dequantisation scales of the fp8 kernels, applied to the raw accumulator.
// chunk -> physical chunk of a row: XOR with the row for power-of-two chunk counts, rotation by the row otherwise (the
// 384-column tile: 12 / 6 chunks) - either way the 16 rows a ds_write_b64 touches spread over the banks
template <int CH>
__device__ __forceinline__ int epi_swz(int chunk, int row) {
    if constexpr ((CH & (CH - 1)) == 0) return chunk ^ (row & (CH - 1));
    else return (chunk + row) % CH;
}

// L32: the accumulators come from 32x32x16 MFMAs, renamed by the caller into f32x4 quads acc[t][j] that hold, for this lane, row
// (j >> 1) * 32 + (lane & 31) and columns t * 16 + (2 * (j & 1) + (lane >> 5)) * 4 .. + 3 of the wave tile (quad q of the 32 x 32
// tile (u, v) is acc[2u + (q >> 1)][2v + (q & 1)]); otherwise the 16x16x32 layout: row j * 16 + (lane & 15), columns
// t * 16 + (lane >> 4) * 4 .. + 3.  Only the lane -> (row, column group) map differs; arithmetic and roundings are the same.
template <bool L32>
__device__ __forceinline__ int epi_row_of(int j, int lane) { return L32 ? (j >> 1) * 32 + (lane & 31) : j * 16 + (lane & 15); }
template <bool L32>
__device__ __forceinline__ int epi_grp_of(int j, int lane) { return L32 ? 2 * (j & 1) + (lane >> 5) : (lane >> 4); }

template <int TN, int TM, bool L32 = false>
__device__ __forceinline__ void epi_wave_tile_lds(const EpiCtx& e, f32x4 (&acc)[TN][TM], char* wreg, int lane, int m_wave0, int M,
                                                  const int32_t* __restrict__ row_idx, int nt_base, int NTT,
                                                  const bf16_t* bias_tile = nullptr,   // LDS copy of bias[nt_base*16 ..], zero past N
                                                  const float* __restrict__ sw = nullptr, const float* __restrict__ sx = nullptr) {
    const bool swiglu = (e.flags & UMV_EPI_SWIGLU) != 0;
    float sxr[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m_wave0 + epi_row_of<L32>(j, lane);
        sxr[j] = sx ? sx[m < M ? m : M - 1] : 1.f;
    }
    auto finish = [&](auto CHC, int col_base, int n_out) {
        constexpr int CH = decltype(CHC)::value;           // 16-byte chunks per row of the wave tile
        constexpr int ROWB = CH * 16, RPI = 64 / CH, NI = (TM * 16 + RPI - 1) / RPI;   // (64 % CH lanes idle when CH is not a power of two)
        const int row_l = lane / CH, chunk = lane % CH;
        const int col0 = col_base + chunk * 8;
        if (col0 >= n_out || row_l >= RPI) return;
        const bool full = col0 + 7 < n_out;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = i * RPI + row_l;
            const int m = m_wave0 + row;
            if (row >= TM * 16 || m >= M) continue;
            u32x4 v = *reinterpret_cast<const u32x4*>(wreg + row * ROWB + (epi_swz<CH>(chunk, row) << 4));
            const int64_t orow = row_idx ? (int64_t)row_idx[m] : (int64_t)m;
            bf16_t* o = reinterpret_cast<bf16_t*>(e.out) + orow * e.ldo + col0;
            if (e.flags & UMV_EPI_RESIDUAL) {
                const bf16_t* rr = e.residual + orow * e.ldr + col0;
                uint32_t rw[4];
                if (full && ((reinterpret_cast<uintptr_t>(rr) & 15) == 0)) {
                    const u32x4 rv = *reinterpret_cast<const u32x4*>(rr);
                    rw[0] = rv.x; rw[1] = rv.y; rw[2] = rv.z; rw[3] = rv.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t lo = (col0 + 2 * q < n_out) ? rr[2 * q] : 0, hi = (col0 + 2 * q + 1 < n_out) ? rr[2 * q + 1] : 0;
                        rw[q] = lo | (hi << 16);
                    }
                }
                uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    vw[q] = pack2bf(__uint_as_float(vw[q] << 16) + __uint_as_float(rw[q] << 16),
                                    __uint_as_float(vw[q] & 0xFFFF0000u) + __uint_as_float(rw[q] & 0xFFFF0000u));
                v.x = vw[0]; v.y = vw[1]; v.z = vw[2]; v.w = vw[3];
            }
            if (full && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                *reinterpret_cast<u32x4*>(o) = v;
            } else {
                const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (col0 + q < n_out) o[q] = (bf16_t)(vw[q >> 1] >> ((q & 1) * 16));
            }
        }
    };
    if (swiglu) {
        constexpr int CH = TN;                    // TN/2 output tiles of 16 columns = 32 bytes each
        static_for<0, TN / 2>([&](auto P) {
            constexpr int p = decltype(P)::value;
            static_for<0, TM>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const int row = epi_row_of<L32>(j, lane), g = epi_grp_of<L32>(j, lane);
                float v[4];
                const float gg[4] = {acc[2 * p][j].x, acc[2 * p][j].y, acc[2 * p][j].z, acc[2 * p][j].w};
                const float uu[4] = {acc[2 * p + 1][j].x, acc[2 * p + 1][j].y, acc[2 * p + 1][j].z, acc[2 * p + 1][j].w};
                if (sw) {
                    const int ng = (nt_base + 2 * p) * 16 + g * 4, nu = ng + 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a_ = rbf(gg[q] * sw[min(ng + q, e.N - 1)] * sxr[j]), b_ = rbf(uu[q] * sw[min(nu + q, e.N - 1)] * sxr[j]);
                        v[q] = rbf(rbf(silu_f(a_)) * b_);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = rbf(rbf(silu_f(rbf(gg[q]))) * rbf(uu[q]));
                }
                u32x2 pk;
                pk.x = pack2bf(v[0], v[1]);
                pk.y = pack2bf(v[2], v[3]);
                const int chunk = 2 * p + (g >> 1);
                *reinterpret_cast<u32x2*>(wreg + row * (CH * 16) + (epi_swz<CH>(chunk, row) << 4) + (g & 1) * 8) = pk;
            });
        });
        finish(std::integral_constant<int, CH>{}, (nt_base >> 1) * 16, e.N / 2);
    } else {
        constexpr int CH = 2 * TN;
        static_for<0, TN>([&](auto T) {
            constexpr int t = decltype(T)::value;
            // the column group of a quad depends on the lane alone (16x16x32) or on the lane and the parity of j (32x32x16):
            // bias / scales of the one or two groups this lane meets in column tile t
            constexpr int NG = L32 ? 2 : 1;
            float b4[NG][4], s4[NG][4];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int g = epi_grp_of<L32>(gi, lane);
                const int n0 = (nt_base + t) * 16 + g * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) { b4[gi][q] = 0.f; s4[gi][q] = 1.f; }
                if (e.flags & UMV_EPI_BIAS) {
                    if (bias_tile) {
                        const u32x2 pk = *reinterpret_cast<const u32x2*>(bias_tile + t * 16 + g * 4);
                        b4[gi][0] = __uint_as_float(pk.x << 16); b4[gi][1] = __uint_as_float(pk.x & 0xFFFF0000u);
                        b4[gi][2] = __uint_as_float(pk.y << 16); b4[gi][3] = __uint_as_float(pk.y & 0xFFFF0000u);
                    } else if (n0 < e.N) {
                        epi_bias4(e, n0, b4[gi]);
                    }
                }
                if (sw) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s4[gi][q] = sw[min(n0 + q, e.N - 1)];
                }
            }
            static_for<0, TM>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int gi = L32 ? (j & 1) : 0;
                const int row = epi_row_of<L32>(j, lane), g = epi_grp_of<L32>(j, lane);
                float v[4] = {acc[t][j].x, acc[t][j].y, acc[t][j].z, acc[t][j].w};
                if (sw) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] * s4[gi][q] * sxr[j];
                }
                if (e.flags & UMV_EPI_BIAS) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += b4[gi][q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = rbf(v[q]);
                if (e.flags & UMV_EPI_GELU_TANH) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = rbf(gelu_tanh_f(v[q]));
                }
                if (e.flags & UMV_EPI_SILU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = rbf(silu_f(v[q]));
                }
                u32x2 pk;
                pk.x = pack2bf(v[0], v[1]);
                pk.y = pack2bf(v[2], v[3]);
                const int chunk = 2 * t + (g >> 1);
                *reinterpret_cast<u32x2*>(wreg + row * (CH * 16) + (epi_swz<CH>(chunk, row) << 4) + (g & 1) * 8) = pk;
            });
        });
        finish(std::integral_constant<int, CH>{}, nt_base * 16, e.N);
    }
}


// ----------------------------------------------------------------------------- lean wave-tile epilogue (round 5)
// epi_wave_tile_lds above decides everything at run time, per quad and per row: which flags are set, whether a row / a chunk is
// inside the matrix, whether a pointer is 16-byte aligned.  Unrolled over a 128 x 128 wave tile that is ~20 000 instructions with
// ~1200 branches, and the tile-level trace of the 4-wave kernel (profiles/r05_w4_tile_trace.txt) shows what it costs: ~35 000
// cycles per 256 x 256 tile - three quarters of the whole k loop at K = 1152, a fifth at K = 3584 - where the store path takes the
// same 128 KiB per CU from every CU at once in ~12 000 (tools/store_bench.hip).
// The lean form serves the flag combinations the model uses (KIND, compile time) on outputs whose rows are 16-byte aligned and
// addressable with 32-bit offsets (epi_lean_kind); everything else keeps the general function.  Same two phases through the wave's
// own LDS region, same arithmetic and rounding points (bit-identical results), but
//   * no per-element decisions: rows beyond M and chunks beyond N are dropped by the buffer range check of
//     buffer_store_dwordx4 / answered with zeros by buffer_load_dwordx4 (dense: num_records = M rows; row-indexed: an
//     out-of-range offset for the lanes concerned), so the loops carry no branches;
//   * LDS addresses are one lane constant XOR a compile-time constant (power-of-two chunk counts) plus an immediate offset;
//   * pack2bf(rbf(x)) is pack2bf(x): a value that is only stored is rounded once;
//   * the residual rows of the whole tile are requested before the first one is needed.
// KIND bits: 1 bias, 2 GELU(tanh) [with bias], 4 residual, 8 SwiGLU.
__host__ __device__ inline int epi_lean_kind(const umv_gemm_args& a) {
    const int f = a.epilogue;
    if (!(f == 0 || f == UMV_EPI_BIAS || f == (UMV_EPI_BIAS | UMV_EPI_GELU_TANH) || f == UMV_EPI_RESIDUAL || f == (UMV_EPI_BIAS | UMV_EPI_RESIDUAL) ||
          f == UMV_EPI_SWIGLU))
        return -1;
    const int kind = ((f & UMV_EPI_BIAS) ? 1 : 0) | ((f & UMV_EPI_GELU_TANH) ? 2 : 0) | ((f & UMV_EPI_RESIDUAL) ? 4 : 0) | ((f & UMV_EPI_SWIGLU) ? 8 : 0);
    const int n_out = (f & UMV_EPI_SWIGLU) ? a.N / 2 : a.N;
    if ((f & UMV_EPI_SWIGLU) && (a.N & 31)) return -1;
    if ((n_out & 7) || (a.ldo & 7) || (reinterpret_cast<uintptr_t>(a.out) & 15) || a.ldo < n_out) return -1;
    const int64_t rows = a.row_idx ? a.x_rows : (int64_t)a.M;
    const int64_t lim = (int64_t)1 << 31;
    if (rows <= 0 || rows * a.ldo * 2 >= lim) return -1;
    if ((f & UMV_EPI_RESIDUAL) && ((a.ldr & 7) || (reinterpret_cast<uintptr_t>(a.residual) & 15) || a.ldr < n_out || rows * a.ldr * 2 >= lim)) return -1;
    return kind;
}

template <int CH>
__device__ __forceinline__ int epi_wrap(int p) { return p >= CH ? p - CH : p; }

template <int TN, int TM, int KIND>
__device__ __forceinline__ void epi_wave_tile_lean(const umv_gemm_args& a, f32x4 (&acc)[TN][TM], char* wreg, int lane, int m_wave0, int nt_base,
                                                   const bf16_t* bias_tile /* LDS: bias[nt_base * 16 ..], zero past N */) {
    constexpr bool BIAS = (KIND & 1) != 0, GELU = (KIND & 2) != 0, RES = (KIND & 4) != 0, SWI = (KIND & 8) != 0;
    constexpr int CH = SWI ? TN : 2 * TN;                  // 16-byte chunks per row of the wave's output tile
    constexpr int ROWB = CH * 16, RPI = 64 / CH, NI = (TM * 16 + RPI - 1) / RPI;
    constexpr bool P2 = (CH & (CH - 1)) == 0;
    constexpr int NQ = SWI ? TN / 2 : TN;                  // column groups of phase 1 (tiles, or gate / up tile pairs)
    static_assert(CH <= 16 || !P2, "row & (CH - 1) == lane & (CH - 1) needs CH <= 16");
    const int r = lane & 15, g = lane >> 4, gh = g >> 1;
    const int n_out = SWI ? a.N / 2 : a.N;
    const int col_base = SWI ? (nt_base >> 1) * 16 : nt_base * 16;
    const uint32_t ldo2 = (uint32_t)a.ldo * 2u, ldr2 = (uint32_t)a.ldr * 2u;
    // ---- where this lane's rows of phase 2 go (row-indexed outputs: ask for the indices now)
    const int row_l = lane / CH, chunk = lane % CH;
    const bool lane_ok = row_l < RPI && col_base + chunk * 8 < n_out;
    const uint32_t colb = (uint32_t)(col_base + chunk * 8) * 2u;
    int32_t ridx[NI];
    if (a.row_idx) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = m_wave0 + i * RPI + row_l;
            ridx[i] = a.row_idx[m < a.M ? m : a.M - 1];
        }
    }
    // ---- phase 1: accumulator layout -> bf16 -> LDS, row (j * 16 + r), chunk 2q + gh, 8-byte half g & 1
    const int L = P2 ? r * ROWB + ((gh ^ (r & (CH - 1))) << 4) + (g & 1) * 8 : r * ROWB + (g & 1) * 8;
    const int c0m = (gh + r) % CH;
    static_for<0, NQ>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BIAS) {
            const u32x2 pk = *reinterpret_cast<const u32x2*>(bias_tile + q * 16 + g * 4);
            b4[0] = __uint_as_float(pk.x << 16); b4[1] = __uint_as_float(pk.x & 0xFFFF0000u);
            b4[2] = __uint_as_float(pk.y << 16); b4[3] = __uint_as_float(pk.y & 0xFFFF0000u);
        }
        static_for<0, TM>([&](auto J) {
            constexpr int j = decltype(J)::value;
            float v[4];
            if constexpr (SWI) {
                const float gg[4] = {acc[2 * q][j].x, acc[2 * q][j].y, acc[2 * q][j].z, acc[2 * q][j].w};
                const float uu[4] = {acc[2 * q + 1][j].x, acc[2 * q + 1][j].y, acc[2 * q + 1][j].z, acc[2 * q + 1][j].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = rbf(silu_f(rbf(gg[k]))) * rbf(uu[k]);     // (the product is rounded by the pack below)
            } else {
                v[0] = acc[q][j].x; v[1] = acc[q][j].y; v[2] = acc[q][j].z; v[3] = acc[q][j].w;
                if constexpr (BIAS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += b4[k];
                }
                if constexpr (GELU) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = gelu_tanh_f(rbf(v[k]));
                }
            }
            u32x2 pk;
            pk.x = pack2bf(v[0], v[1]);
            pk.y = pack2bf(v[2], v[3]);
            int off;
            if constexpr (P2) off = (L ^ (q << 5)) + j * 16 * ROWB;
            else off = L + j * 16 * ROWB + (epi_wrap<CH>(c0m + (2 * q + 16 * j) % CH) << 4);
            *reinterpret_cast<u32x2*>(wreg + off) = pk;
        });
    });
    // ---- phase 2: whole rows; an instruction covers RPI rows of ROWB bytes
    const int64_t nrows = a.row_idx ? 0 : (int64_t)a.M;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.row_idx ? 0x7FFFFFFF : (int)(nrows * ldo2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(RES ? a.residual : (const bf16_t*)a.out), 0,
                                                                            a.row_idx ? 0x7FFFFFFF : (int)(nrows * (RES ? ldr2 : ldo2)), 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;
    uint32_t vo[NI], vr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        bool ok = lane_ok;
        if (i * RPI + RPI > TM * 16) ok = ok && (i * RPI + row_l < TM * 16);       // (chunk counts that do not divide 64: the last instruction overshoots)
        if (a.row_idx) {
            ok = ok && (m_wave0 + i * RPI + row_l < a.M);
            vo[i] = ok ? (uint32_t)ridx[i] * ldo2 + colb : OOB;
            if constexpr (RES) vr[i] = ok ? (uint32_t)ridx[i] * ldr2 + colb : OOB;
        } else {
            vo[i] = ok ? (uint32_t)(m_wave0 + i * RPI + row_l) * ldo2 + colb : OOB;
            if constexpr (RES) vr[i] = ok ? (uint32_t)(m_wave0 + i * RPI + row_l) * ldr2 + colb : OOB;
        }
    }
    u32x4 rv[RES ? NI : 1];
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < NI; ++i) rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, vr[i], 0, 0);
    }
    const int rl = row_l & (CH - 1);
    const int R = P2 ? row_l * ROWB + ((chunk ^ rl) << 4) : row_l * ROWB;
    const int c1 = (chunk + row_l) % CH;
    static_for<0, NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        int off;
        if constexpr (P2) off = (R ^ (((i * RPI) & (CH - 1)) << 4)) + i * RPI * ROWB;
        else off = R + i * RPI * ROWB + (epi_wrap<CH>(c1 + (i * RPI) % CH) << 4);
        if constexpr (!P2) off = row_l < RPI ? off : 0;                          // idle lanes read a valid address
        u32x4 v = *reinterpret_cast<const u32x4*>(wreg + off);
        if constexpr (RES) {
            const uint32_t rw[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
            uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                vw[k] = pack2bf(__uint_as_float(vw[k] << 16) + __uint_as_float(rw[k] << 16),
                                __uint_as_float(vw[k] & 0xFFFF0000u) + __uint_as_float(rw[k] & 0xFFFF0000u));
            v.x = vw[0]; v.y = vw[1]; v.z = vw[2]; v.w = vw[3];
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, vo[i], 0, 0);
    });
}

// dispatch on the run-time kind (wave-uniform); returns false when the general function must be used
template <int TN, int TM>
__device__ __forceinline__ bool epi_wave_tile_lean_any(int kind, const umv_gemm_args& a, f32x4 (&acc)[TN][TM], char* wreg, int lane, int m_wave0,
                                                       int nt_base, const bf16_t* bias_tile) {
    switch (kind) {
    case 0: epi_wave_tile_lean<TN, TM, 0>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true;
    case 1: epi_wave_tile_lean<TN, TM, 1>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true;
    case 3: epi_wave_tile_lean<TN, TM, 3>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true;
    case 4: epi_wave_tile_lean<TN, TM, 4>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true;
    case 5: epi_wave_tile_lean<TN, TM, 5>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true;
    case 8:
        if constexpr (TN % 2 == 0) { epi_wave_tile_lean<TN, TM, 8>(a, acc, wreg, lane, m_wave0, nt_base, bias_tile); return true; }
        return false;
    default: return false;
    }
}
