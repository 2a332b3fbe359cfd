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
//     through "+a" constraints without spilling into the counted-vmcnt loop (profiles/HISTORY.md section 5b, round 4).  Here the accumulators
//     are LITERAL registers a[4i : 4i+3] in the instruction text.  The compiler never sees them: it allocates only the
//     fragment / address VGPRs, the kernel zeroes a[0:255] itself and moves them out 128 at a time for the epilogue.
//     (What keeps this sound: the compiler uses AGPRs on its own only to spill, and this kernel's VGPR pressure is < 200 of
//     256; tests/test_host_cpu.py::test_w4_kernels_own_their_agprs greps the ISA for any v_accvgpr it did not write.)
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
__host__ __device__ constexpr int w4_filler_at(int slot, int nf, int span) {
    for (int k = 0; k < nf; ++k)
        if (k * span / nf == slot) return k;
    return -1;
}
// the compiler must treat a fragment as rewritten (an asm ds_read filled it behind its back, a counted wait has just retired it)
__device__ __forceinline__ void w4_touch(bf16x8& f) { asm volatile("" : "+v"(f)); }
// ABL = 2 (timing study): s_memtime stamps; the value arrives like a scalar load, the caller consumes it behind an lgkmcnt(0)
__device__ __forceinline__ void w4_stamp(uint64_t& t) { asm volatile("s_memtime %0" : "=s"(t)); }
template <int R>
__device__ __forceinline__ float w4_acc_read() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
    return v;
}

// ABL (TIMING ONLY, wrong results; UMV_GEMM_ABLATIONS builds): 1 = the K advance wraps every 8 k-steps (every piece an L2 hit);
// 2 = the first 8 workgroups log s_memtime stamps per k-step of steps 16..79 to (uint64_t*)a.w_scale - tools/w4_trace.py.
// RAGK: K % 32 != 0 - the x chunks of the last k-step that lie beyond K are staged as zeros (out-of-range buffer offsets).
//
// Schedule of one k-step ("body" s; the fragments of tile s are in registers, read during body s-1):
//     MFMA 0 .. QB-1        tile s, groups of TM MFMAs per W fragment; before group t a counted lgkmcnt wait for W fragment t
//     s_waitcnt vmcnt(NP) lgkmcnt(0); s_barrier      my pieces of tile s+1 have landed / everyone's; everyone's reads of tile s done
//     MFMA QB .. NMMA-1     with, between them: the 16 fragment reads of tile s+1 (x first, then W in the order the next body's
//                           groups need them) and the NP LDS-DMA pieces of W(s+3) and of half an x pair
//  -> nothing but MFMAs at the boundary between two bodies: the matrix pipe keeps running while the waves drift back into step at
//     the barrier in the MIDDLE of a body (round 5 trace of the barrier-first form: 1650 cycles per k-step for 1024 of MFMAs,
//     ~420 of them between the last MFMA of a step and the first of the next: profiles/r05_w4_v1_kstep_trace.txt).
// The loop is unrolled over the period of the rings (6 k-steps), so every LDS address is a constant, and the pieces are
// buffer_load_dwordx4 ... offen lds with ONE offset VGPR per piece for the whole kernel; K advances in the scalar offset
// (clamped at the last k-tile / pair: the trailing bodies re-stage data nobody reads).
template <int WN, int WM, int TN, int TM, int ABL = 0, bool RAGK = false>
__global__ __launch_bounds__(WN * WM * 64) void gemm_w4_kernel(umv_gemm_args a, int KT, int NTT, int mblocks, int nblocks, int gn, int ms, int lean) {
    constexpr int NW = WN * WM;
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr int WTILES = BN / 16;
    constexpr int WPW = WTILES / NW, XPP = (BM / 8) / NW, XPB = XPP / 2, NP = WPW + XPB;
    constexpr int WSLOT = WTILES * 1024, XSLOT = BM * 128, XBASE = 3 * WSLOT;
    constexpr int STAGE_BYTES = 3 * WSLOT + 3 * XSLOT;
    constexpr int NRD = TN + TM, NMMA = TN * TM, NACC = NMMA * 4;
    constexpr int QB = 2 * TM;                       // the barrier sits behind the first two groups
    static_assert(NW == 4, "one wave per SIMD");
    static_assert(WTILES % NW == 0 && (BM / 8) % NW == 0 && XPP % 2 == 0, "even split of the staging pieces over the waves");
    static_assert(NACC <= 256 && NRD + NP <= NMMA - QB, "accumulators fit the AGPR file; at most one read / piece per MFMA behind the barrier");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wn = wave % WN, wm = wave / WN;
    uint64_t tw0 = 0, tw1 = 0, tw2 = 0, tw3 = 0, tw4 = 0;       // ABL = 2: tile-level stamps (entry, first body, loop end, first epilogue chunk, end)
    if constexpr (ABL == 2) w4_stamp(tw0);
    asm volatile("" ::: "a0", "a255");      // (the clobber list is what tells the compiler that this kernel owns AGPRs at all)
    int mblk, nblk;
    umv_tile_order(mblocks, nblocks, gn, ms, (int)blockIdx.x, mblk, nblk);
    const int m0 = mblk * BM;
    const int nt_blk = nblk * (BN / 16);
    const int nt_base = nt_blk + wn * TN;
    bf16_t* bias_lds = reinterpret_cast<bf16_t*>(smem + STAGE_BYTES);
    if ((a.epilogue & UMV_EPI_BIAS) && tid < BN) {
        const int n = nt_blk * 16 + tid;
        bias_lds[tid] = n < a.N ? a.bias[n] : (bf16_t)0;
    }
    if constexpr (BN > NW * 64) {
        if ((a.epilogue & UMV_EPI_BIAS) && tid + NW * 64 < BN) {
            const int n = nt_blk * 16 + tid + NW * 64;
            bias_lds[tid + NW * 64] = n < a.N ? a.bias[n] : (bf16_t)0;
        }
    }
    const int nsteps = KT;
    const int q_last = (a.K - 1) >> 6;               // the last k-step pair that holds data
    // ---- the staging pieces: one 32-bit offset per piece, relative to a.wp / a.x, valid for every k-step
    // W: n-tiles past NTT are clamped to the last one (their columns are never stored); x: rows past M are clamped likewise
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t offW[WPW], offX[XPP];
    const int xchunk = (lane & 7) ^ ((lane >> 3) & 7);
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int nt = min(nt_blk + wave * WPW + i, NTT - 1);
        offW[i] = (uint32_t)nt * (uint32_t)KT * 1024u + lane * 16;
    }
#pragma unroll
    for (int i = 0; i < XPP; ++i) {
        const int m = m0 + (wave * XPP + i) * 8 + (lane >> 3);
        const int mm = m < a.M ? m : a.M - 1;
        const int64_t row = a.row_idx ? (int64_t)a.row_idx[mm] : (int64_t)mm;
        offX[i] = (uint32_t)((row * a.ldx + xchunk * 8) * 2);
    }
    // RAGK: lanes whose 8 k of the last pair lie beyond K read out of range (zeros)
    const bool xtail = RAGK && (q_last * 64 + xchunk * 8 >= a.K);
    // scalar K offsets of the group the next body stages
    // (char* and a cast at the call: a lambda RETURNING an address_space(3) pointer makes the host pass drop the kernel's stub)
    auto lds_of = [&](int byte) -> char* { return smem + byte; };
    auto stage_w = [&](int kt, auto SLOT, auto I) {
        constexpr int slot = decltype(SLOT)::value, i = decltype(I)::value;
        int ktc = min(kt, KT - 1);
        if constexpr (ABL == 1) ktc &= 7;
        const uint32_t off = offW[i];          // (passed directly, the array element makes the host pass drop the kernel stub)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_w4_t)lds_of(slot * WSLOT + (wave * WPW + i) * 1024), 16, off, ktc * 1024, 0, 0);
    };
    auto stage_x = [&](int q, auto SLOT, auto HALF, auto I) {
        constexpr int slot = decltype(SLOT)::value, hf = decltype(HALF)::value, i = decltype(I)::value;
        int qc = min(q, q_last);
        uint32_t off = offX[hf * XPB + i];
        if constexpr (RAGK) off = (xtail && qc == q_last) ? 0x80000000u : off;
        if constexpr (ABL == 1) qc &= 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr_w4_t)lds_of(XBASE + slot * XSLOT + (wave * XPP + hf * XPB + i) * 1024), 16, off, qc * 128, 0, 0);
    };
    // prologue, in the order the loop would have issued it: x pair 0, W(0), x pair 1, W(1), first half of x pair 2, W(2)
    static_for<0, 3>([&](auto T) {
        constexpr int t = decltype(T)::value;
        static_for<0, XPB>([&](auto I) { stage_x(t, T, std::integral_constant<int, 0>{}, I); });
        if constexpr (t < 2) static_for<0, XPB>([&](auto I) { stage_x(t, T, std::integral_constant<int, 1>{}, I); });
        static_for<0, WPW>([&](auto I) { stage_w(t, T, I); });
    });
    // the accumulators: a[0 : NACC-1] = 0, behind the prologue's loads (256 instructions under ~2 us of first-piece latency)
    static_for<0, NACC>([&](auto I) {
        constexpr int i = decltype(I)::value;
        asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(i));
    });
    bf16x8 wfA[TN], xfA[TM], wfB[TN], xfB[TM];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_w4_t)smem;
    const uint32_t wbase = lds0 + wn * TN * 1024 + lane * 16;
    // x fragment of k half h: row (wm * TM + j) * 16 + r, chunk (4h + g) ^ (r & 7)
    const uint32_t xbase0 = lds0 + XBASE + (wm * TM * 16 + r) * 128 + ((g ^ (r & 7)) << 4), xbase1 = xbase0 ^ 64;
    const uint32_t xbase0_hi = xbase0 + 65536, xbase1_hi = xbase1 + 65536;
    // fragment read number rd of tile (slot sw, pair slot sx, k half h): x fragments first, then the W fragments in group order
    auto read_frag = [&](auto RD, auto SW, auto SX, auto HALF, bf16x8(&wf)[TN], bf16x8(&xf)[TM]) {
        constexpr int rd = decltype(RD)::value, sw = decltype(SW)::value, sx = decltype(SX)::value, h = decltype(HALF)::value;
        if constexpr (rd < TM) {
            constexpr int off = sx * XSLOT + rd * 2048;          // (the instruction's offset field has 16 bits)
            if constexpr (off < 65536) w4_lds_read_frag<off>(xf[rd], h ? xbase1 : xbase0);
            else w4_lds_read_frag<off - 65536>(xf[rd], h ? xbase1_hi : xbase0_hi);
        } else {
            static_assert(2 * WSLOT + (TN - 1) * 1024 < 65536, "W fragment offsets fit the instruction");
            w4_lds_read_frag<sw * WSLOT + (rd - TM) * 1024>(wf[rd - TM], wbase);
        }
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * XPB + 2 * WPW) : "memory");      // x pair 0 and W(0) landed; what was issued behind them may fly
    UMV_BARRIER();
    static_for<0, NRD>([&](auto RD) { read_frag(RD, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, wfA, xfA); });
    if constexpr (ABL == 2) w4_stamp(tw1);

    // body U (compile time: position in the period of 6) of k-step `step`
    auto body = [&](auto UC, int step, bf16x8(&wc)[TN], bf16x8(&xc)[TM], bf16x8(&wnx)[TN], bf16x8(&xnx)[TM]) {
        constexpr int u = decltype(UC)::value;
        constexpr bool even = (u & 1) == 0;
        constexpr int sw_next = (u + 1) % 3, sx_next = ((u + 1) >> 1) % 3;        // slots of tile step + 1
        constexpr int sw_stage = u % 3, sx_stage = ((u + 5) >> 1) % 3;            // slots of W(step + 3) and of x pair (step + 5) >> 1
        uint64_t ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
        if constexpr (ABL == 2) w4_stamp(ts0);
        static_for<0, NMMA>([&](auto I) {
            constexpr int i = decltype(I)::value, t = i / TM, j = i % TM;
            if constexpr (j == 0 && i < QB) {         // group t starts: W fragment t (and, for t = 0, every x fragment) has landed
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN - 1 - t) : "memory");
                w4_touch(wc[t]);
                if constexpr (t == 0) static_for<0, TM>([&](auto J) { w4_touch(xc[decltype(J)::value]); });
            }
            if constexpr (i == QB) {
                if constexpr (ABL == 2) w4_stamp(ts1);
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
                static_for<QB / TM, TN>([&](auto T) { w4_touch(wc[decltype(T)::value]); });
                UMV_BARRIER();
                if constexpr (ABL == 2) w4_stamp(ts2);
            }
            w4_mfma<i>(wc[t], xc[j]);
            if constexpr (i >= QB) {
                // behind the barrier: one filler every second MFMA - reads and pieces alternate until the pieces run out
                constexpr int f = i - QB;             // filler slot
                constexpr int NF = NRD + NP;
                constexpr int span = NMMA - QB - 2;   // the last two MFMAs stay bare
                constexpr int k = w4_filler_at(f, NF, span);      // filler k (0 .. NF-1) goes behind MFMA QB + k * span / NF
                if constexpr (k >= 0) {
                    // fillers 0, 2, 4 ... are reads while pieces remain (piece p is filler 2p + 1), then only reads
                    constexpr bool is_piece = (k & 1) && (k / 2 < NP);
                    if constexpr (is_piece) {
                        constexpr int pc = k / 2;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (pc < XPB) stage_x((step + 5) >> 1, std::integral_constant<int, sx_stage>{}, std::integral_constant<int, even ? 1 : 0>{}, std::integral_constant<int, pc < XPB ? pc : 0>{});
                        else stage_w(step + 3, std::integral_constant<int, sw_stage>{}, std::integral_constant<int, pc >= XPB ? pc - XPB : 0>{});
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        constexpr int rd = k < 2 * NP ? k / 2 : k - NP;
                        read_frag(std::integral_constant<int, rd>{}, std::integral_constant<int, sw_next>{}, std::integral_constant<int, sx_next>{},
                                  std::integral_constant<int, even ? 1 : 0>{}, wnx, xnx);      // tile step + 1 is the odd k half of its pair in an even body
                    }
                }
            }
        });
        if constexpr (ABL == 2) {
            w4_stamp(ts3);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts0), "+s"(ts1), "+s"(ts2), "+s"(ts3)::"memory");
            if (blockIdx.x < 8 && step >= 16 && step < 80 && lane == 0) {
                uint64_t* tr = reinterpret_cast<uint64_t*>(const_cast<float*>(a.w_scale)) + (((int)blockIdx.x * NW + wave) * 64 + (step - 16)) * 5;
                tr[0] = ts0; tr[1] = ts1; tr[2] = ts2; tr[3] = ts3; tr[4] = ts3;
            }
        }
    };
    for (int base = 0; base < nsteps; base += 6) {
        body(std::integral_constant<int, 0>{}, base, wfA, xfA, wfB, xfB);
        if (base + 1 >= nsteps) break;
        body(std::integral_constant<int, 1>{}, base + 1, wfB, xfB, wfA, xfA);
        if (base + 2 >= nsteps) break;
        body(std::integral_constant<int, 2>{}, base + 2, wfA, xfA, wfB, xfB);
        if (base + 3 >= nsteps) break;
        body(std::integral_constant<int, 3>{}, base + 3, wfB, xfB, wfA, xfA);
        if (base + 4 >= nsteps) break;
        body(std::integral_constant<int, 4>{}, base + 4, wfA, xfA, wfB, xfB);
        if (base + 5 >= nsteps) break;
        body(std::integral_constant<int, 5>{}, base + 5, wfB, xfB, wfA, xfA);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> v_accvgpr_read wait states by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    UMV_BARRIER();      // every wave has read its last fragments and all pieces have landed: the staging buffers are free
    if constexpr (ABL == 2) w4_stamp(tw2);

    // epilogue: the accumulators leave the AGPRs 4 m-tiles at a time (TN x 4 x 4 = 128 registers at TN = 8) and go through the
    // wave's own LDS region as whole rows (gemm_epilogue.h); same arithmetic and roundings as the 8-wave tiles
    EpiCtx e{a.bias, a.residual, a.ldr, a.out, a.ldo, a.N, a.epilogue};
    constexpr int JC = TN > 8 ? 2 : 4;      // (TN = 12: 192 live accumulator copies made hipcc spill ~270 registers - into AGPRs - inside the epilogue)
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
        // lean >= 0: the branch-free form for this call's flag combination (gemm_epilogue.h::epi_lean_kind), else the general one
        if (!(lean >= 0 && epi_wave_tile_lean_any<TN, JC>(lean, a, acc, wreg, lane, m0 + wm * TM * 16 + h * JC * 16, nt_base, bias_lds + wn * TN * 16)))
            epi_wave_tile_lds<TN, JC, false>(e, acc, wreg, lane, m0 + wm * TM * 16 + h * JC * 16, a.M, a.row_idx, nt_base, NTT,
                                            bias_lds + wn * TN * 16);
        if constexpr (ABL == 2 && h == 0) w4_stamp(tw3);
    });
    if constexpr (ABL == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the stores of this wave have been accepted
        w4_stamp(tw4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tw0), "+s"(tw1), "+s"(tw2), "+s"(tw3), "+s"(tw4)::"memory");
        // tile-level log behind the k-step log (8 x NW x 64 x 5 words): 6 words per wave of the first 64 workgroups and of the last 64
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        const int slot = b < 64 ? b : (b >= nb - 64 ? 64 + (b - (nb - 64)) : -1);
        if (slot >= 0 && lane == 0) {
            uint64_t* tr = reinterpret_cast<uint64_t*>(const_cast<float*>(a.w_scale)) + 8 * NW * 64 * 5 + (slot * NW + wave) * 6;
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            tr[0] = tw0; tr[1] = tw1; tr[2] = tw2; tr[3] = tw3; tr[4] = tw4; tr[5] = hw;
        }
    }
}

template <int WN, int WM, int TN, int TM, int ABL = 0>
static int launch_w4(const umv_gemm_args& a, int KT, int NTT, int gn, hipStream_t s) {
    constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
    constexpr size_t lds = (size_t)3 * (BN / 16) * 1024 + (size_t)3 * BM * 128 + (BN * 2 + 15) / 16 * 16;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert((size_t)WN * WM * TN * 4 * 512 <= lds, "epilogue regions fit the staging area");
    const int mblocks = (a.M + BM - 1) / BM, nblocks = (a.N + BN - 1) / BN;
    const int ms = umv_tile_superblock(mblocks, BM, a.K);
    auto go = [&](auto RAG) {
        constexpr bool ragk = decltype(RAG)::value;
        static bool attr_set[UMV_MAX_DEVICES] = {};
        if (umv_first_on_device(attr_set))
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<WN, WM, TN, TM, ABL, ragk>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm_w4_kernel<WN, WM, TN, TM, ABL, ragk>), dim3(mblocks * nblocks), dim3(WN * WM * 64), lds, s, a, KT, NTT, mblocks, nblocks, gn, ms,
                           umv_gemm_lean_epilogue(a));
    };
    if (a.K % 32) go(std::true_type{});
    else go(std::false_type{});
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// Shapes the 32-bit piece offsets can address: the packed weight image and every x row the call can touch lie within 2 GiB of
// their base pointers (x_rows = rows of the buffer row_idx points into; unknown -> the caller keeps the 8-wave tiles)
bool umv_gemm_w4_can_take(const umv_gemm_args& a, int KT, int NTT) {
    if ((a.epilogue & UMV_EPI_OUT_F32) || a.k_splits > 1) return false;
    if ((int64_t)NTT * KT * 1024 >= ((int64_t)1 << 31)) return false;
    const int64_t rows = a.row_idx ? a.x_rows : (int64_t)a.M;
    if (rows <= 0 || rows * a.ldx * 2 + 256 >= ((int64_t)1 << 31)) return false;
    return true;
}

// cfg: 466 = 256(n) x 256(m), 468 = 256(n) x 128(m), 4384 = 384(n) x 128(m); bf16 output, no split-K (umv_gemm_w4_can_take)
int umv_gemm_w4_launch(const umv_gemm_args& a, int KT, int NTT, int cfg, int gn, hipStream_t s) {
    if (!umv_gemm_w4_can_take(a, KT, NTT)) {
        umv_set_error("gemm_w4: shape / mode not addressable by the 4-wave tiles (fp32 output, K split, or operands beyond 2 GiB)");
        return UMV_ERR_UNSUPPORTED;
    }
    if (cfg == 466) return launch_w4<2, 2, 8, 8>(a, KT, NTT, gn, s);
    if (cfg == 468) return launch_w4<2, 2, 8, 4>(a, KT, NTT, gn, s);
    if (cfg == 4384) return launch_w4<2, 2, 12, 4>(a, KT, NTT, gn, s);
#ifdef UMV_GEMM_ABLATIONS
    if (cfg == 94661) return launch_w4<2, 2, 8, 8, 1>(a, KT, NTT, gn, s);      // timing only: every piece an L2 hit
    if (cfg == 94662) return launch_w4<2, 2, 8, 8, 2>(a, KT, NTT, gn, s);      // s_memtime trace (a.w_scale = the log)
#endif
    umv_set_error("gemm_w4: unknown tile configuration %d", cfg);
    return UMV_ERR_ARG;
}
