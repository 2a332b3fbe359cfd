// Pieces shared by the tiled GEMM kernels of libunimedvl_hip (gemm.hip, gemm_w4.hip): the workgroup -> tile order, the
// instruction slots of the interleaved schedules, and the launcher of the 4-wave kernels.  Internal to csrc/.
#pragma once
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include <stdlib.h>

// A/B and tuning knobs are read from the environment ONCE per process: `static const int v = umv_env_int("NAME", default);` at the
// point of use (the initialisation of a function-local static is thread-safe)
static inline int umv_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// read r of NRD goes right after MFMA number (r * SPAN) / NRD, SPAN = 3/4 of the step's MFMAs: evenly spread over the first three
// quarters, first one after the first MFMA, so that the last quarter's MFMAs cover the latency of the last reads before the
// step's closing s_waitcnt lgkmcnt(0)
__host__ __device__ constexpr int umv_interleave_slot(int i, int nmma, int nrd) {
    const int span = (nmma * 3 / 4 >= nrd) ? nmma * 3 / 4 : nmma;
    for (int r = 0; r < nrd; ++r)
        if ((r * span) / nrd == i) return r;
    return -1;
}
// piece p of n_pieces goes behind MFMA floor((2p + 1) * n_mma / (2 * n_pieces)): evenly spread, never behind the last MFMA
__host__ __device__ constexpr int umv_dma_slot(int i, int n_mma, int n_pieces) {
    for (int p = 0; p < n_pieces; ++p)
        if (((2 * p + 1) * n_mma) / (2 * n_pieces) == i) return p;
    return -1;
}

// XCD-aware tile order (blockIdx round-robins over the 8 XCDs): every XCD gets a contiguous run of tiles; inside the run,
// strips of gn n-blocks, m-block next, n-block within the strip fastest, so that the ~32 tiles an XCD works on at a time share
// both operands' k-slices in its L2.  M super-blocks of ms m-blocks: all XCDs finish one before the next (x stays in the
// memory-side cache while the strips of W stream past).  See gemm.hip::gemm_tiled_kernel for the measurements.
__device__ __forceinline__ void umv_tile_order(int mblocks, int nblocks, int gn, int ms, int block, int& mblk, int& nblk) {
    const int sb_tiles = ms * nblocks;
    const int sb = block / sb_tiles;
    const int mb0 = sb * ms, mb_n = min(ms, mblocks - mb0);
    const int nwg = mb_n * nblocks;
    int bid = block - sb * sb_tiles;
    {
        const int q = nwg / 8, rem = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
    }
    const int per = mb_n * gn, strip = bid / per, rem = bid - strip * per;
    const int w = min(gn, nblocks - strip * gn);
    mblk = mb0 + rem / w;
    nblk = strip * gn + rem % w;
}

// m-blocks per super-block: ~64 MB of x rows above 16k rows, one super-block below (host side)
static inline int umv_tile_superblock(int mblocks, int BM, int K) {
    int ms = (int)(((int64_t)64 << 20) / ((int64_t)BM * K * 2));
    ms = ms < 8 ? 8 : ms;
    if ((int64_t)mblocks * BM < 16384) ms = mblocks;
    if (ms > mblocks || ms * 3 / 2 >= mblocks) ms = mblocks;
    else ms = (mblocks + (mblocks + ms - 1) / ms - 1) / ((mblocks + ms - 1) / ms);
    return ms;
}

// gemm_w4.hip: 4-wave tiles with the accumulators in AGPRs (cfg 466 / 468 / 4384); bf16 output, no split-K, operands within
// 2 GiB of their base pointers (umv_gemm_w4_can_take)
int umv_gemm_lean_epilogue(const umv_gemm_args& a);      // gemm.hip: >= 0 = the lean epilogue kind of this call, -1 = general
bool umv_gemm_w4_can_take(const umv_gemm_args& a, int KT, int NTT);
int umv_gemm_w4_launch(const umv_gemm_args& a, int KT, int NTT, int cfg, int gn, hipStream_t s);

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
// 16 e4m3 values x one power-of-two scale -> 16 bf16 (exact: an e4m3 value times 2^e is a bf16 value)
__device__ __forceinline__ void cvt_fp8x16(u32x4 q, float scale, bf16x8& lo, bf16x8& hi) {
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
