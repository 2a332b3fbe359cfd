// Tiled bf16 GEMM with ONE wave per SIMD and the accumulators in the AGPR half of the register file (gfx950).
//
//   out[m,n] = epi(sum_k x[m,k] W[n,k])        M > 64: prefill, ViT, flow passes (qwen2_navit.py:541-543,617-620,
//                                              modeling_qwen2.py:234-235, siglip_navit.py:216-218,243,256-258)
//
// Same operands, LDS image, rings and counted-wait protocol as gemm.hip's SCHED = 3 tile (W: 1 KiB fragment-order pieces of
// the packed image; x: 8 rows x 128 bytes = full cache lines, row-major image XOR-swizzled by the row), but a workgroup is
// WN x WM = 4 waves and a wave owns TN x TM = 8 x 8 (or 8 x 4, 12 x 4) MFMA tiles:
//   * a 128 x 128 wave tile needs 16 fragment reads per 64 MFMAs where the 8-wave kernel's 128 x 64 tiles need 12 per 32:
//     a third less LDS read traffic per flop, half the waves at the barrier, nobody to share the SIMD's matrix pipe with;
//   * its 256 accumulator registers cannot live in VGPRs next to 2 x 16 fragments, and hipcc (ROCm 7.2) cannot allocate them
//     through "+a" constraints without spilling into the counted-vmcnt loop.  Here the accumulators
//     are LITERAL registers a[4i : 4i+3] in the instruction text.  The compiler never sees them: it allocates only the
//     fragment / address VGPRs, the kernel zeroes a[0:255] itself and moves them out 128 at a time for the epilogue.
//     (What keeps this sound: the compiler uses AGPRs on its own only to spill, and this kernel's VGPR pressure is < 200 of
//     256; tools/check_w4_isa.py greps the ISA for any v_accvgpr it did not write.)
// Same MFMAs on the same operands in the same k order as the 8-wave tiles: results are bit-identical to them.
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include "gemm_epilogue.h"
#include "gemm_internal.h"

__device__ __attribute__((aligned(16))) const uint32_t g_zero_page_w4[4] = {0, 0, 0, 0};

typedef __attribute__((address_space(3))) void* lds_ptr_w4_t;

template <int OFF>
__device__ __forceinline__ void w4_lds_read_frag(bf16x8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// acc IDX += a x b with the accumulator named in the instruction text
template <int IDX>
__device__ __forceinline__ void w4_mfma(const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(IDX * 4), "n"(IDX * 4 + 3));
}
// ABL = 2 (timing study): s_memtime stamps; the value arrives like a scalar load, the caller consumes it behind an lgkmcnt(0)
__device__ __forceinline__ void w4_stamp(uint64_t& t) { asm volatile("s_memtime %0" : "=s"(t)); }
template <int R>
__device__ __forceinline__ float w4_acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}

// NWS = slots of the W ring (3: W(t+3) is staged while tile t computes; 4: one more k-step of look-ahead, the tile's bias then comes
// from global memory - the rings take all 160 KiB).  ABL (TIMING ONLY, wrong results): 1 = the K advance wraps every 8 k-steps, so
// that every piece after the first pass is an L2 hit; 2 = the first 8 workgroups log five s_memtime stamps per k-step of steps 16..79
// (body entry, my pieces landed, barrier passed, last MFMA issued, fragments landed) to (uint64_t*)a.w_scale - tools/w4_trace.py.
template <int WN, int WM, int TN, int TM, int NWS = 3, int ABL = 0>
__global__ __launch_bounds__(WN * WM * 64) void gemm_w4_kernel(umv_gemm_args a, int KT, int NTT, int mblocks, int nblocks, int gn, int ms) {
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int WTILES = BN / 16;
    constexpr int WPW = (WTILES + NW - 1) / NW, XPP = (BM / 8) / NW, XPB = XPP / 2, NP = WPW + XPB;
    constexpr int WSLOT = WTILES * 1024, XSLOT = BM * 128, XBASE = NWS * WSLOT;
    constexpr int STAGE_BYTES = NWS * WSLOT + 3 * XSLOT;
    constexpr bool BIAS_LDS = STAGE_BYTES + BN * 2 <= 160 * 1024;
    static_assert(NWS == 3 || NWS == 4, "W ring depth");
    constexpr int NRD = TN + TM, NMMA = TN * TM, NACC = NMMA * 4;
    static_assert(NW == 4, "one wave per SIMD");
    static_assert(WTILES % NW == 0 && (BM / 8) % NW == 0 && XPP % 2 == 0, "even split of the staging pieces over the waves");
    static_assert(NACC <= 256 && NRD <= NMMA && NP <= NMMA, "accumulators fit the AGPR file; at most one read / piece per MFMA");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    // the accumulators: a[0 : NACC-1] = 0 (the clobber list is what tells the compiler that this kernel owns AGPRs at all)
    asm volatile("" ::: "a0", "a255");
    static_for<0, NACC>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(i));
    });
    int mblk, nblk;
    umv_tile_order(mblocks, nblocks, gn, ms, (int)blockIdx.x, mblk, nblk);
    const int m0 = mblk * BM;
    const int nt_blk = nblk * (BN / 16);
    const int nt_base = nt_blk + wn * TN;
    bf16_t* bias_lds = reinterpret_cast<bf16_t*>(smem + STAGE_BYTES);
    if (BIAS_LDS && (a.epilogue & UMV_EPI_BIAS) && tid < BN) {
        const int n = nt_blk * 16 + tid;
        bias_lds[tid] = n < a.N ? a.bias[n] : (bf16_t)0;
    }
    if constexpr (BIAS_LDS && BN > NW * 64) {
        if ((a.epilogue & UMV_EPI_BIAS) && tid + NW * 64 < BN) {
            const int n = nt_blk * 16 + tid + NW * 64;
            bias_lds[tid + NW * 64] = n < a.N ? a.bias[n] : (bf16_t)0;
        }
    }
    const int KTL = KT;
    const int nsteps = KTL;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_w4);
    const int kx_rel = a.K;
    const bf16_t* curW[WPW];
    int bumpW[WPW];
    const bf16_t* curX[XPP];
    const int xchunk = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int nt = nt_blk + wave * WPW + i;
        const bool ok = nt < NTT;
        curW[i] = ok ? a.wp + ((int64_t)nt * KT) * 512 + lane * 8 : zero;
        bumpW[i] = ok ? 512 : 0;
    }
#pragma unroll
    for (int i = 0; i < XPP; ++i) {
        const int m = m0 + (wave * XPP + i) * 8 + (lane >> 3);
        const int mm = m < a.M ? m : a.M - 1;                // rows past M are clamped (their outputs are masked)
        const int64_t row = a.row_idx ? (int64_t)a.row_idx[mm] : (int64_t)mm;
        curX[i] = a.x + row * a.ldx + xchunk * 8;
    }
    auto dstW = [&](int slot, int i) -> char* { return smem + slot * WSLOT + (wave * WPW + i) * 1024; };
    auto dstX = [&](int slot, int i) -> char* { return smem + XBASE + slot * XSLOT + (wave * XPP + i) * 1024; };
    const bf16_t* pw[WPW];
    const bf16_t* px[XPB];
    auto prep_w = [&](int kt) {                 // the W pieces of k-tile kt: zero page past the K range
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            pw[i] = kt < KTL ? curW[i] : zero;
            curW[i] += bumpW[i];
            if constexpr (ABL == 1) { if ((kt & 7) == 7) curW[i] -= 8 * bumpW[i]; }
        }
    };
    auto prep_x = [&](int q, auto HALF) {       // pieces [HALF * XPB, +XPB) of k-step pair q: per-lane zero fill at the K tail
        constexpr int hf = decltype(HALF)::value;
        const int k0 = q * 64;
        if (k0 + 64 <= kx_rel) {
#pragma unroll
            for (int i = 0; i < XPB; ++i) {
                px[i] = curX[hf * XPB + i];
                curX[hf * XPB + i] += 64;
                if constexpr (ABL == 1) { if ((q & 3) == 3) curX[hf * XPB + i] -= 4 * 64; }
            }
        } else {
#pragma unroll
            for (int i = 0; i < XPB; ++i) { px[i] = (k0 + xchunk * 8 < kx_rel) ? curX[hf * XPB + i] : zero; curX[hf * XPB + i] += 64; }
        }
    };
    // prologue, in the order the loop would have issued it.  NWS = 3: x pair 0, W(0), x pair 1, W(1), first half of x pair 2, W(2);
    // NWS = 4: x pair 0, W(0), first half of pair 1, W(1), second half, W(2), first half of pair 2, W(3)
    auto issue_x = [&](int q, auto HALF) {
        constexpr int hf = decltype(HALF)::value;
        prep_x(q, HALF);
#pragma unroll
        for (int i = 0; i < XPB; ++i) __builtin_amdgcn_global_load_lds((const void*)px[i], (lds_ptr_w4_t)dstX(q % 3, hf * XPB + i), 16, 0, 0);
    };
    auto issue_w = [&](int t) {
        prep_w(t);
#pragma unroll
        for (int i = 0; i < WPW; ++i) __builtin_amdgcn_global_load_lds((const void*)pw[i], (lds_ptr_w4_t)dstW(t % NWS, i), 16, 0, 0);
    };
    constexpr std::integral_constant<int, 0> H0{};
    constexpr std::integral_constant<int, 1> H1{};
    issue_x(0, H0); issue_x(0, H1); issue_w(0);
    if constexpr (NWS == 3) {
        issue_x(1, H0); issue_x(1, H1); issue_w(1);
        issue_x(2, H0); issue_w(2);
    } else {
        issue_x(1, H0); issue_w(1);
        issue_x(1, H1); issue_w(2);
        issue_x(2, H0); issue_w(3);
    }
    bf16x8 wfA[TN], xfA[TM], wfB[TN], xfB[TM];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_w4_t)smem;
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
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWS == 3 ? 3 * XPB + 2 * WPW : 3 * NP) : "memory");      // x pair 0 and W(0) landed; what was issued behind them may fly
    UMV_BARRIER();
    static_for<0, TN>([&](auto T) {
        constexpr int t = decltype(T)::value;
        w4_lds_read_frag<t * 1024>(wfA[t], lds0 + woff);
    });
    static_for<0, TM>([&](auto J) {
        constexpr int j = decltype(J)::value;
        w4_lds_read_frag<j * 2048>(xfA[j], lds0 + xoff0);
    });
    land(wfA, xfA);
    auto body = [&](auto EVEN, int step, bf16x8(&wc)[TN], bf16x8(&xc)[TM], bf16x8(&wnx)[TN], bf16x8(&xnx)[TM]) {
        constexpr bool even = decltype(EVEN)::value;
        uint64_t ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
        if constexpr (ABL == 2) w4_stamp(ts0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NWS - 2) * NP) : "memory");      // tile step + 1 has landed (mine); the NWS - 2 groups behind it may fly
        if constexpr (ABL == 2) w4_stamp(ts1);
        UMV_BARRIER();                                                  // ... everyone's; and the slot of W tile `step` is free
        if constexpr (ABL == 2) w4_stamp(ts2);
        const int q = (step + 5) >> 1;                                  // the x pair this body stages half of
        if constexpr (even) prep_x(q, std::integral_constant<int, 1>{});
        else prep_x(q, std::integral_constant<int, 0>{});
        prep_w(step + NWS);
        const int sw = step % NWS, sx = q % 3;
        const uint32_t wa = lds0 + ((step + 1) % NWS) * WSLOT + woff;
        const uint32_t xa = lds0 + (((step + 1) >> 1) % 3) * XSLOT + (even ? xoff1 : xoff0);     // tile step + 1 is the odd half in an even body
        static_for<0, NMMA>([&](auto I) {
            constexpr int i = decltype(I)::value, t = i / TM, j = i % TM;
            w4_mfma<i>(wc[t], xc[j]);
            constexpr int rd = umv_interleave_slot(i, NMMA, NRD);
            if constexpr (rd >= 0 && rd < TN) w4_lds_read_frag<(rd < TN ? rd : 0) * 1024>(wnx[rd < TN ? rd : 0], wa);
            else if constexpr (rd >= TN) w4_lds_read_frag<(rd >= TN ? rd - TN : 0) * 2048>(xnx[rd >= TN ? rd - TN : 0], xa);
            constexpr int pc = umv_dma_slot(i, NMMA, NP);
            if constexpr (pc >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (pc < XPB)
                    __builtin_amdgcn_global_load_lds((const void*)px[pc < XPB ? pc : 0], (lds_ptr_w4_t)dstX(sx, (even ? XPB : 0) + pc), 16, 0, 0);
                else
                    __builtin_amdgcn_global_load_lds((const void*)pw[pc >= XPB ? pc - XPB : 0], (lds_ptr_w4_t)dstW(sw, pc - XPB), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if constexpr (ABL == 2) w4_stamp(ts3);
        land(wnx, xnx);
        if constexpr (ABL == 2) {
            w4_stamp(ts4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts0), "+s"(ts1), "+s"(ts2), "+s"(ts3), "+s"(ts4)::"memory");
            if (blockIdx.x < 8 && step >= 16 && step < 80 && lane == 0) {
                uint64_t* tr = reinterpret_cast<uint64_t*>(const_cast<float*>(a.w_scale)) + (((int)blockIdx.x * NW + wave) * 64 + (step - 16)) * 5;
                tr[0] = ts0; tr[1] = ts1; tr[2] = ts2; tr[3] = ts3; tr[4] = ts4;
            }
        }
    };
    for (int step = 0; step < nsteps; step += 2) {
        body(std::true_type{}, step, wfA, xfA, wfB, xfB);
        if (step + 1 < nsteps) body(std::false_type{}, step + 1, wfB, xfB, wfA, xfA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> v_accvgpr_read wait states by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    UMV_BARRIER();      // every wave has read its last fragments: the staging buffers are free

    // epilogue: the accumulators leave the AGPRs 4 m-tiles at a time (TN x 4 x 4 = 128 registers at TN = 8) and go through the
    // wave's own LDS region as whole rows (gemm_epilogue.h); same arithmetic and roundings as the 8-wave tiles
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    constexpr int JC = 4;
    static_assert(TM % JC == 0, "m-tiles per epilogue chunk");
    char* wreg = smem + wave * (TN * JC * 512);
    static_for<0, TM / JC>([&](auto H) {
        constexpr int h = decltype(H)::value;
        f32x4 acc[TN][JC];
        static_for<0, TN>([&](auto T) {
            constexpr int t = decltype(T)::value;
            static_for<0, JC>([&](auto J) {
                constexpr int jj = decltype(J)::value, idx = (t * TM + h * JC + jj) * 4;
                acc[t][jj] = (f32x4){w4_acc_read<idx>(), w4_acc_read<idx + 1>(), w4_acc_read<idx + 2>(), w4_acc_read<idx + 3>()};
            });
        });
        epi_wave_tile_lds<TN, JC, false>(e, acc, wreg, lane, m0 + wm * TM * 16 + h * JC * 16, a.M, a.row_idx, nt_base, NTT,
                                        BIAS_LDS ? bias_lds + wn * TN * 16 : nullptr);
    });
}

template <int WN, int WM, int TN, int TM, int NWS = 3, int ABL = 0>
static int launch_w4(const umv_gemm_args& a, int KT, int NTT, int gn, hipStream_t s) {
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr size_t rings = (size_t)NWS * (BN / 16) * 1024 + (size_t)3 * BM * 128;
    constexpr size_t lds = rings + BN * 2 <= 160 * 1024 ? rings + (BN * 2 + 15) / 16 * 16 : rings;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((size_t)WN * WM * TN * 4 * 512 <= lds, "epilogue regions fit the staging area");
    static bool attr_set[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr_set)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<WN, WM, TN, TM, NWS, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int mblocks = (a.M + BM - 1) / BM, nblocks = (a.N + BN - 1) / BN;
    const int ms = umv_tile_superblock(mblocks, BM, a.K);
    hipLaunchKernelGGL((gemm_w4_kernel<WN, WM, TN, TM, NWS, ABL>), dim3(mblocks * nblocks), dim3(WN * WM * 64), lds, s, a, KT, NTT, mblocks, nblocks, gn, ms);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// cfg: 466 = 256(n) x 256(m), 468 = 256(n) x 128(m), 484 -> 4384 = 384(n) x 128(m); bf16 output, no split-K (the caller checks)
int umv_gemm_w4_launch(const umv_gemm_args& a, int KT, int NTT, int cfg, int gn, hipStream_t s) {
    if (cfg == 466) return launch_w4<2, 2, 8, 8>(a, KT, NTT, gn, s);
    if (cfg == 4664) return launch_w4<2, 2, 8, 8, 4>(a, KT, NTT, gn, s);          // W ring of 4 slots
    if (cfg == 4684) return launch_w4<2, 2, 8, 4, 4>(a, KT, NTT, gn, s);
#ifdef UMV_GEMM_ABLATIONS
    if (cfg == 94661) return launch_w4<2, 2, 8, 8, 3, 1>(a, KT, NTT, gn, s);      // timing only: every piece an L2 hit
    if (cfg == 94664) return launch_w4<2, 2, 8, 8, 4, 1>(a, KT, NTT, gn, s);
    if (cfg == 94662) return launch_w4<2, 2, 8, 8, 3, 2>(a, KT, NTT, gn, s);      // s_memtime trace (a.w_scale = the log)
#endif
    if (cfg == 468) return launch_w4<2, 2, 8, 4>(a, KT, NTT, gn, s);
    if (cfg == 4384) return launch_w4<2, 2, 12, 4>(a, KT, NTT, gn, s);
    umv_set_error("gemm_w4: unknown tile configuration %d", cfg);
    return UMV_ERR_ARG;
}
