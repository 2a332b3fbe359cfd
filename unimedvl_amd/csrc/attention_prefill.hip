// Prefill attention for libunimedvl_hip (gfx950): the nsplit == 1 path of umv_attn_varlen for hd 128 / 72
// (flash_attn_varlen_func at qwen2_navit.py:605-614, siglip_navit.py:232-241).
//
// attention.hip's attn_kernel lets every wave stream its own K / V^T fragments from L2: with 64-128 q-tiles per
// (segment, kv head) that is 64-128x the K/V bytes through L2->L1 (ViT: 2.4 GB per layer = 12 TB/s, the measured ceiling
// of that path).  Here the 4 waves of a workgroup take TQ q-tiles each (4*TQ tiles of the same segment / kv head) and
// share every 32- or 64-key stage through LDS:
//   * LDS-DMA gathers the K and V^T fragments straight into MFMA fragment order (every lane supplies its own source
//     address, the destination is lane-linear, so the consumers' ds_read_b128 are conflict free), double buffered, one
//     barrier per stage;
//   * a fragment read from LDS feeds TQ MFMAs (one per q-tile of the wave): half the LDS bytes per flop at TQ = 2;
//   * interior blocks skip the per-element causal / length masks;
//   * round 5, the shipped form (LAZY): the softmax runs against a lazy reference point (attn_softmax_lazy below) and the q-tiles pack
//     the (token, head) pairs densely (16 per tile at any group size).  Same softmax, P rounded at another scale than attn_kernel's.
// With UMV_ATTN_LAZY=0 the kernels keep the exact running maximum: per row the arithmetic and its order are then those of attn_kernel
// (both call common.h::attn_softmax_block) and the results are bit-identical to it; the lazy kernels (TQ = 1, 2, either packing) are
// bit-identical among themselves (tools/attn_ab.py, tests/test_kernel_branches_gpu.py::test_attn_kernel_variants_bit_identical).
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include <stdlib.h>
#include <type_traits>

#define ATTN_LAZY_TAU 8.0f       // lazy softmax reference: exponents stay <= ATTN_LAZY_TAU (P <= 256)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int B, int E, class F>
__device__ __forceinline__ void static_for_attn(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for_attn<B + 1, E>(f);
    }
}
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;

// 32-key blocks per LDS stage and the occupancy asked of the register allocator.  The kernel is latency bound (every wave runs
// K reads -> QK^T -> softmax chain -> V reads -> PV in series; MFMA and VALU pipes are each < 35 % busy), so resident waves are
// what buys throughput: TQ = 2 uses one block per stage (hd 128: 2 x 16 KiB per workgroup, 162 VGPRs = 3 waves per SIMD;
// hd 72: 2 x 11 KiB, held to 128 VGPRs = 4 waves per SIMD: 102 -> 93 us per ViT layer).  TQ = 1, i.e. small grids (the
// 34-token text prefill: 128 workgroups), has no co-resident workgroup to hide the one-stage prefetch distance behind and
// keeps two blocks per stage (32-key stages cost it 47 -> 126 us per layer).  Same block order either way: bit-identical.
constexpr int ATTN_PREFILL_NB(int hd, int tq) { return (tq == 2) ? 1 : 2; }
constexpr int ATTN_PREFILL_WAVES(int hd, int tq) { return (hd <= 96 && tq == 2) ? 4 : 1; }   // minimum waves per SIMD asked of the register allocator
// (a build that spills here is WRONG, not just slow: scratch traffic counts in vmcnt and the loop's waits are counted by hand.
// tests/test_host_cpu.py::test_attention_prefill_isa_has_no_scratch compiles this file and checks every kernel.)

__device__ __forceinline__ bf16x8 attn_mask_keys(bf16x8 v, int nvalid) {   // keep the first nvalid (0..8) elements
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = j < nvalid ? v[j] : (short)0;
    return o;
}


// Online softmax of one 32-key block against a LAZY reference point, for the TQ q-tiles of a wave at once (round 5).
// P = 2^(s c - m_ref) with m_ref <= the row's running maximum <= m_ref + ATTN_LAZY_TAU: the reference moves (and O, l are rescaled by
// alpha) only when some exponent exceeds ATTN_LAZY_TAU = 8 (P <= 256; the softmax is shift invariant, so only the rounding of P changes:
// mean error against exact fp32 attention 9.2e-5 vs 8.7e-5 with the exact running maximum).  With the exact maximum the rescale fires
// in most blocks - the maximum of n keys still moves with probability 32 / n per block and row, and a wave has 16 TQ rows.  The
// trigger test runs on each lane's own eight exponents: no cross-lane step, no alpha, no 16 DT multiplies in the common path; the
// row sums stay per-lane partials (reduced once, at the end).  State per row: nm = -m_ref (0 while unset), thr = ATTN_LAZY_TAU (-inf while
// unset: any finite score of a row's first block sets the reference), l.  When the reference moves by d, the exponents already
// computed are shifted (e - d) instead of recomputed from the scores: one rounding of ~1e-6 in the exponent, and the scores need
// not stay live.  Returns true (wave-uniform) when the reference moved; alpha is written only then.
#ifndef UMV_ATTN_PAIR_DEBUG
#define UMV_ATTN_PAIR_DEBUG 0
#endif
template <int TQ, bool MASKED, bool STATS, int DBG = 0>
__device__ __forceinline__ bool attn_softmax_lazy(const f32x4 (&st)[TQ][2], int kb, int g, const int (&limit)[TQ], const int (&my_end)[TQ], float c,
                                                  float (&nm_run)[TQ], float (&thr_run)[TQ], float (&l_run)[TQ], float (&alpha)[TQ], bf16x8 (&pf)[TQ],
                                                  uint32_t (&cnt)[2]) {
    const umv_f32x2_v c2 = {c, c};
    umv_f32x2_v e[TQ][4];
    float emax[TQ];
    bool trig = false;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[t * 4 + r] = st[u][t][r];
                if constexpr (MASKED) {
                    const int key = kb + g * 8 + t * 4 + r;
                    v[t * 4 + r] = (key <= limit[u] && key < my_end[u]) ? v[t * 4 + r] : -INFINITY;
                }
            }
        const umv_f32x2_v n2 = {nm_run[u], nm_run[u]};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) e[u][q4] = __builtin_elementwise_fma((umv_f32x2_v){v[2 * q4], v[2 * q4 + 1]}, c2, n2);   // (-inf) c + n = -inf
        // Every instruction of this function is COMPILER-VISIBLE (round 6): the exponents are fma results, which the compiler knows to be
        // canonical, so fmaxf folds into v_max3_f32 without the v_max_f32 x, x it needs on raw MFMA results (the reason common.h's
        // attn_softmax_block takes its maxima as inline asm) - and the hazard recogniser sees every producer and consumer of the
        // v_permlane*_swap steps below (VALU write -> swap read needs wait states that it cannot place around inline asm).
        const float ea = fmaxf(fmaxf(e[u][0][0], e[u][0][1]), e[u][1][0]), eb = fmaxf(fmaxf(e[u][1][1], e[u][2][0]), e[u][2][1]);
        emax[u] = fmaxf(fmaxf(ea, eb), fmaxf(e[u][3][0], e[u][3][1]));
        trig = trig || emax[u] > thr_run[u];
    }
    if constexpr (DBG == 6 && TQ == 2) {      // bisect: ONE compare for both tiles (thr = -inf while unset: emax - thr = +inf, or NaN for an all-masked row)
        trig = fmaxf(emax[0] - thr_run[0], emax[1] - thr_run[1]) > 0.f;
    }
    const bool moved = DBG == 4 ? true : __any(trig);
    if (moved) {
        // the rows' cross-lane maxima first, then the selects, branch-free
        float mxr[TQ];
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            if constexpr (DBG == 2) mxr[u] = fmaxf(fmaxf(emax[u], __shfl_xor(emax[u], 16, 64)), fmaxf(__shfl_xor(emax[u], 32, 64), __shfl_xor(emax[u], 48, 64)));
            else mxr[u] = xor32_max(xor16_max(emax[u]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DBG == 3) { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int u = 0; u < TQ; ++u) {
            const float mx = mxr[u];
            const bool unset = thr_run[u] == -INFINITY;
            // only the rows that crossed their OWN trigger level move (all lanes of a row agree after the cross-lane maximum): a row's
            // result does not depend on which other rows share its tile - the tile packing below may change without changing a bit
            const bool need = mx > thr_run[u];
            const float d = need ? mx : 0.f;                 // the reference rises by d (an unset row: to its first maximum)
            const float a2 = umv_exp2(-d);
            alpha[u] = (need && !unset) ? a2 : 1.0f;         // (an unset row has O = 0, l = 0)
            l_run[u] *= alpha[u];
            nm_run[u] -= d;
            thr_run[u] = need ? ATTN_LAZY_TAU : thr_run[u];
            const umv_f32x2_v d2 = {d, d};
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) e[u][q4] = e[u][q4] - d2;
            // diagnostics of the parity tests (umv_attn_args.stats, the STATS kernels only): how often each form of the rare path ran.
            // Wave-uniform counters, added to memory once at the end of the kernel
            if constexpr (STATS) {
                cnt[0] += __any(need && !unset) ? 1u : 0u;
                cnt[1] += __any(need && unset) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        umv_f32x2_v pv[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) pv[q4] = (umv_f32x2_v){umv_exp2(e[u][q4][0]), umv_exp2(e[u][q4][1])};
        const umv_f32x2_v s2 = (pv[0] + pv[1]) + (pv[2] + pv[3]);
        l_run[u] += s2[0] + s2[1];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const uint32_t w = pack2bf(pv[q4][0], pv[q4][1]);
            pf[u][2 * q4] = (short)(w & 0xFFFFu);
            pf[u][2 * q4 + 1] = (short)(w >> 16);
        }
    }
    return moved;
}

// LAZY: 0 = exact running maximum, 1 = lazy reference (one softmax call per q-tile), 2 = lazy, both tiles in one call.  STATS: the kernel
// also counts its rare-path events into umv_attn_args.stats (a separate instantiation, selected when stats != NULL: the counters cost the
// hd-72 two-tile kernel the last of its 128 registers; same arithmetic, and the tests compare its output with the plain kernel's bit for bit)
// PAGED: K / V^T are page pools behind umv_attn_args.page_table (a separate instantiation of the shipped lazy kernels: the slab kernels are
// not touched; same arithmetic in the same block order, so a paged call gives the slab call's bits).  A stage's DMA offset becomes
// page base + offset inside the page; a 32-key block (and a two-block stage) never straddles a 256-key page.
template <int HD, int TQ, int LAZY, bool STATS = false, bool PAGED = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STATS ? 1 : ATTN_PREFILL_WAVES(HD, TQ)))) void attn_prefill_kernel(umv_attn_args a, float scale_log2e, int dense) {
    constexpr int KS = (HD + 31) / 32;
    constexpr int DT = (HD + 15) / 16;
    constexpr int FK = 2 * KS, FB = FK + DT;   // fragments (1 KiB each) per 32-key block: K then V^T
    constexpr int NB = ATTN_PREFILL_NB(HD, TQ);    // 32-key blocks per stage
    constexpr int STAGE = NB * FB * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int G = a.nq / a.nkv;
    const int QPT = 16 / G > 0 ? 16 / G : 1;
    // the 16 columns of a q-tile are (token, head) pairs of one kv head.  QPT packing: whole tokens, 16 / G of them (G = 7: 14 columns
    // used); DENSE packing (round 5): pairs run on across tile boundaries, PPT = 16 per tile - 12.5 % fewer tiles at G = 7.  Per row
    // nothing changes (same keys, same blocks, its own softmax reference): the two packings give the same bits.
    const int PPT = dense ? 16 : G * QPT;
    const int s = blockIdx.z;
    const int kh = blockIdx.y;
    const int qt_wg = blockIdx.x * 4 * TQ;      // first q-tile of the workgroup
    const int q0 = a.cu_q[s];
    const int Lq = a.cu_q[s + 1] - q0;
    const int Lk = a.kv_len[s];
    const int npairs = Lq * G;
    if (qt_wg * PPT >= npairs || Lk <= 0) return;   // uniform over the workgroup

    int qi[TQ], head[TQ], limit[TQ], my_end[TQ], min_limit[TQ];
    bool rvalid[TQ];
    bf16x8 qf[TQ][KS];
    int wave_end = 0;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int qt = qt_wg + wave * TQ + u;
        const int p0 = qt * PPT;                   // first (token, head) pair of the tile
        const bool active = p0 < npairs;
        const int pr = p0 + j;
        qi[u] = pr / G;
        head[u] = kh * G + (pr - qi[u] * G);
        rvalid[u] = active && (j < PPT) && (pr < npairs);
        const int first_tok = p0 / G, last_tok = min(Lq - 1, (p0 + PPT - 1) / G);
        limit[u] = a.causal ? (Lk - Lq + qi[u]) : (Lk - 1);     // bottom-right aligned causal mask
        min_limit[u] = a.causal ? (Lk - Lq + first_tok) : (Lk - 1);
        int e = Lk;
        if (a.causal) e = min(Lk, Lk - Lq + last_tok + 1);
        my_end[u] = active ? e : 0;
        wave_end = max(wave_end, my_end[u]);
        const bf16_t* qp = a.q + (int64_t)(q0 + (rvalid[u] ? qi[u] : 0)) * (a.q_row_stride ? a.q_row_stride : (int64_t)a.nq * HD) + (rvalid[u] ? head[u] : 0) * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            qf[u][ks] = (rvalid[u] && d < HD) ? ldg_frag(qp + d) : zero_frag();
        }
    }
    int blk_end = Lk;
    if (a.causal) {
        const int last_blk = min(Lq - 1, ((qt_wg + 4 * TQ - 1) * PPT + PPT - 1) / G);
        blk_end = min(Lk, Lk - Lq + last_blk + 1);
    }
    const int nstages = (blk_end + 32 * NB - 1) / (32 * NB);
    const int64_t kstride = a.k_key_stride ? a.k_key_stride : HD;     // packed K (k_key_stride > 0): rows cu_q[s] .. of a [T, ...] buffer
    const bf16_t* kbase = a.k_slab + (a.k_key_stride ? (int64_t)q0 * a.k_key_stride : (PAGED ? 0 : s * a.k_seg_stride)) + kh * a.k_head_stride;
    const bf16_t* vbase = a.vt_slab + (PAGED ? 0 : s * a.v_seg_stride) + kh * a.v_head_stride;
    const int32_t* ptab = PAGED ? a.page_table + (int64_t)s * a.page_table_stride : nullptr;
    const uint32_t kpage_bytes = PAGED ? (uint32_t)(a.k_seg_stride * 2) : 0u, vpage_bytes = PAGED ? (uint32_t)(a.v_seg_stride * 2) : 0u;

    // ---- staging: fragment f of a stage (K fragments (t, ks) of its 32-key blocks, then their V^T fragments) is fetched by wave f % 4
    // with ONE buffer_load_dwordx4 ... lds: the lane's byte offset inside this (segment, kv head)'s K rows / V^T rows is a kernel
    // constant, the stage advances a scalar offset, and everything outside the data - keys beyond Lk, the d >= HD lanes of a padded
    // fragment, the end of the V^T rows - is answered with zeros by the buffer's range check (round 5: the per-fragment 64-bit
    // address arithmetic of the global_load form was ~100 of the ~300 VALU instructions a wave issued per stage, and the kernel is
    // VALU-issue bound: profiles/r05_attn_stage_trace.txt).  V^T columns in [Lk, cap) may hold stale values: the tail mask below
    // zeroes them (as before); a fragment that runs past the end of a V^T row reads the next row's head - keys >= Lk, masked too.
    constexpr int NFR = (NB * FB + 3) / 4;         // fragments per wave and stage
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kbase), 0,
                                                                         PAGED ? 0x7FFFFFFF : (int)min((int64_t)Lk * kstride * 2, (int64_t)0x7FFFFFFF), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vbase), 0,
                                                                         PAGED ? 0x7FFFFFFF : (int)min((int64_t)HD * a.v_d_stride * 2, (int64_t)0x7FFFFFFF), 0x00020000);
    uint32_t voff[NFR];
#pragma unroll
    for (int i = 0; i < NFR; ++i) {
        const int f = wave + 4 * i;                // (wave-uniform)
        const int ff = f % FB;
        if (ff < FK) {         // K fragment (t, ks): row i of tile t <-> key kb + (i>>2)*8 + t*4 + (i&3)
            const int t = ff / KS, ks = ff - t * KS;
            const int d = ks * 32 + g * 8;
            voff[i] = d < HD ? (uint32_t)((((j >> 2) * 8 + t * 4 + (j & 3)) * kstride + d) * 2) : 0x80000000u;
        } else {               // V^T fragment dt: row d = dt*16 + j, keys kb + g*8 .. +8
            const int d = (ff - FK) * 16 + j;
            voff[i] = d < HD ? (uint32_t)(((int64_t)d * a.v_d_stride + g * 8) * 2) : 0x80000000u;
        }
    }
    const uint32_t kstep = (uint32_t)(32 * kstride * 2);            // bytes per 32-key block of K rows
    auto stage = [&](int sidx, int buf) {
#pragma unroll
        for (int i = 0; i < NFR; ++i) {
            const int f = wave + 4 * i;
            if (f < NB * FB) {
                const int b = f / FB, ff = f - b * FB;
                const uint32_t blk = (uint32_t)(sidx * NB + b);
                const uint32_t off = voff[i];          // (a local: passing the array element makes the host pass drop the kernel's stub)
                char* dst = smem + buf * STAGE + f * 1024;
                uint32_t ks_off = blk * kstep, vs_off = blk * 64u;
                if constexpr (PAGED) {                 // (uniform scalar load; the table's 64-byte lines hold 16 pages = 128 blocks)
                    const uint32_t pg = (uint32_t)ptab[blk >> (UMV_KV_PAGE_LOG2 - 5)], inpage = blk & ((UMV_KV_PAGE >> 5) - 1);
                    ks_off = pg * kpage_bytes + inpage * kstep;
                    vs_off = pg * vpage_bytes + inpage * 64u;
                }
                if (ff < FK) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (attn_lds_ptr_t)dst, 16, off, ks_off, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (attn_lds_ptr_t)dst, 16, off, vs_off, 0, 0);
            }
        }
    };

    f32x4 o[TQ][DT];
    float m_run[TQ], l_run[TQ];
    float nm_run[TQ], thr_run[TQ];    // LAZY: -reference (0 while unset), trigger level (-inf while unset)
    uint32_t cnt[2] = {0u, 0u};       // LAZY: rare-path events of this wave (uniform), see umv_attn_args.stats
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        m_run[u] = -INFINITY;
        l_run[u] = 0.f;
        nm_run[u] = 0.f;
        thr_run[u] = -INFINITY;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

#ifdef UMV_ATTN_TRACE       // timing study (UMV_ATTN_TRACE=1 python -m unimedvl_amd.build; tools/attn_trace.py): s_memtime stamps per stage -> a.workspace
#define ATTN_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(t)); __builtin_amdgcn_sched_barrier(0); } while (0)
    uint64_t ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
#else
#define ATTN_STAMP(t) do { } while (0)
#endif
    stage(0, 0);
    for (int sidx = 0; sidx < nstages; ++sidx) {
        ATTN_STAMP(ts0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my part of stage sidx has landed
        UMV_BARRIER();                            // ... everyone's; and everyone is done with stage sidx-1
        if (sidx + 1 < nstages) stage(sidx + 1, (sidx + 1) & 1); // overlaps the math below
        ATTN_STAMP(ts1);
        const char* sb = smem + (sidx & 1) * STAGE;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int kb = (sidx * NB + b) * 32;
            if (kb >= wave_end) continue;
            const char* fb = sb + b * FB * 1024 + lane * 16;
            // ---- S^T = K Q^T : one K fragment from LDS feeds the wave's TQ tiles
            f32x4 st[TQ][2];
#pragma unroll
            for (int u = 0; u < TQ; ++u) st[u][0] = st[u][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(fb + (t * KS + ks) * 1024);
#pragma unroll
                    for (int u = 0; u < TQ; ++u) st[u][t] = mfma16(kf, qf[u][ks], st[u][t]);
#if UMV_ATTN_PAIR_DEBUG
                    if constexpr (LAZY == 7) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // bisect: keep the next ds_read away from these MFMAs
#endif
                }
            ATTN_STAMP(ts2);
            bf16x8 pf[TQ];
            float alpha[TQ];
            bool rescale = false;
            // Online softmax on the RAW scores (common.h::attn_softmax_block, shared with attn_kernel: same bits); the masked form
            // of a boundary block is a code path of its own (merging the two forms after the masks cost the interior path 12
            // register copies per tile).
            auto softmax_tile = [&](auto U, auto MASKED) {
                constexpr int u = decltype(U)::value;
                constexpr bool masked = decltype(MASKED)::value;
                float v[8];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[t * 4 + r] = st[u][t][r];
                        if constexpr (masked) {
                            const int key = kb + g * 8 + t * 4 + r;
                            v[t * 4 + r] = (key <= limit[u] && key < my_end[u]) ? v[t * 4 + r] : -INFINITY;
                        }
                    }
                attn_softmax_block(v, scale_log2e, m_run[u], l_run[u], alpha[u], pf[u]);
            };
            // both tiles of the wave in ONE basic block per form: the two softmax chains are independent, and side by side the
            // scheduler interleaves them (a branch per tile left each chain's ~25-deep dependency latency exposed: the stage trace
            // showed 1200 cycles for ~90 instructions)
            bool all_interior = true;
#pragma unroll
            for (int u = 0; u < TQ; ++u) all_interior = all_interior && (kb + 32 <= my_end[u] && kb + 31 <= min_limit[u]);   // wave uniform
            if constexpr (LAZY >= 2) {
                // both q-tiles of the wave in ONE call (one trigger test, one rare path): UMV_ATTN_VARIANT_PAIR, see the note below
                constexpr int DBG = LAZY - 2;
                if constexpr (DBG == 1) {
                    if constexpr (TQ == 2) attn_mfma_guard(st[0][0], st[0][1], st[1][0], st[1][1]);
                }
                rescale = all_interior ? attn_softmax_lazy<TQ, false, STATS, DBG>(st, kb, g, limit, my_end, scale_log2e, nm_run, thr_run, l_run, alpha, pf, cnt)
                                       : attn_softmax_lazy<TQ, true, STATS, DBG>(st, kb, g, limit, my_end, scale_log2e, nm_run, thr_run, l_run, alpha, pf, cnt);
                if (!rescale) {
#pragma unroll
                    for (int u = 0; u < TQ; ++u) alpha[u] = 1.0f;
                }
            } else if constexpr (LAZY == 1) {
                // one call per q-tile.  The two tiles of a wave in ONE call (LAZY >= 2: one trigger test, one rare path for both) is NOT shipped:
                // that kernel gives wrong rows in the SECOND tile of some waves now and then, never the same twice.  Round 6 bisected it
                // (profiles/r06_attn_pair_nondeterminism.txt, tools/attn_pair_debug.py): not the inline-asm maxima round 5 suspected (the
                // function is compiler-visible throughout now), not the v_permlane*_swap steps (a ds_bpermute form fails too), not MFMA ->
                // VALU wait states (tools/mfma_raw_probe.hip: 7 needed, hipcc places 8, at any occupancy); it needs >= 2 waves per SIMD, it
                // goes away when the rare path is unconditional, and the per-lane state dump shows tile 1's scores of lanes 48-63 garbage
                // in one block.  Cause not found; the form is compiled in UMV_ATTN_PAIR_DEBUG builds only.  The per-tile form below is
                // held to a determinism stress test on the bench shapes (tests/test_attn_lazy_gpu.py::test_lazy_kernels_are_deterministic).
                static_for_attn<0, TQ>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    float a1[1] = {1.0f};
                    const bool mv = all_interior
                        ? attn_softmax_lazy<1, false, STATS>(reinterpret_cast<const f32x4 (&)[1][2]>(st[u]), kb, g, reinterpret_cast<const int (&)[1]>(limit[u]),
                                                      reinterpret_cast<const int (&)[1]>(my_end[u]), scale_log2e, reinterpret_cast<float (&)[1]>(nm_run[u]),
                                                      reinterpret_cast<float (&)[1]>(thr_run[u]), reinterpret_cast<float (&)[1]>(l_run[u]), a1,
                                                      reinterpret_cast<bf16x8 (&)[1]>(pf[u]), cnt)
                        : attn_softmax_lazy<1, true, STATS>(reinterpret_cast<const f32x4 (&)[1][2]>(st[u]), kb, g, reinterpret_cast<const int (&)[1]>(limit[u]),
                                                     reinterpret_cast<const int (&)[1]>(my_end[u]), scale_log2e, reinterpret_cast<float (&)[1]>(nm_run[u]),
                                                     reinterpret_cast<float (&)[1]>(thr_run[u]), reinterpret_cast<float (&)[1]>(l_run[u]), a1,
                                                     reinterpret_cast<bf16x8 (&)[1]>(pf[u]), cnt);
                    alpha[u] = mv ? a1[0] : 1.0f;
                    rescale = rescale || mv;
                });
            } else {
            if (all_interior) {
                if constexpr (TQ == 2) attn_mfma_guard(st[0][0], st[0][1], st[1][0], st[1][1]);      // raw accumulators go to asm maxima (common.h)
                else attn_mfma_guard(st[0][0], st[0][1]);
                static_for_attn<0, TQ>([&](auto U) { softmax_tile(U, std::false_type{}); });
            }
            else static_for_attn<0, TQ>([&](auto U) { softmax_tile(U, std::true_type{}); });
#pragma unroll
            for (int u = 0; u < TQ; ++u) rescale = rescale || __any(alpha[u] != 1.0f);
            }
            // ---- O^T += V^T P^T : one V^T fragment from LDS feeds the TQ tiles
            // Straight-line code: the rescale and the tail mask are hoisted out of the dt loop as wave-uniform branches, so the
            // fragment reads are in flight before the first MFMA (with the branches inside the loop every fragment was a
            // ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs basic block of its own: 8 exposed LDS latencies per block).
            // Every MFMA is the builtin; the file is compiled with -mllvm -amdgpu-mfma-vgpr-form (build.py), which keeps all
            // MFMA destinations in VGPRs: by default the compiler parks the 64 O accumulators in AGPRs and moves them out and
            // back around every rescale (PMC then: 15 VALU instructions per MFMA).  Hazards and waits are the compiler's.
            ATTN_STAMP(ts3);
            constexpr int H0 = (DT + 1) / 2;      // V^T fragments in two halves: half the registers, the second half's reads fly
            const bool tail = kb + 32 > Lk;       // behind the first half's MFMAs.  tail: the last, partial block of the segment
            const int nvalid = min(8, max(0, Lk - (kb + g * 8)));
            bf16x8 vfa[H0];
#pragma unroll
            for (int dt = 0; dt < H0; ++dt) vfa[dt] = *reinterpret_cast<const bf16x8*>(fb + (FK + dt) * 1024);
            if (rescale) {
#pragma unroll
                for (int u = 0; u < TQ; ++u)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        o[u][dt].x *= alpha[u]; o[u][dt].y *= alpha[u]; o[u][dt].z *= alpha[u]; o[u][dt].w *= alpha[u];
                    }
            }
            if (tail) {
#pragma unroll
                for (int dt = 0; dt < H0; ++dt) vfa[dt] = attn_mask_keys(vfa[dt], nvalid);
            }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 vfb[DT - H0];
#pragma unroll
            for (int dt = H0; dt < DT; ++dt) vfb[dt - H0] = *reinterpret_cast<const bf16x8*>(fb + (FK + dt) * 1024);
#pragma unroll
            for (int dt = 0; dt < H0; ++dt)
#pragma unroll
                for (int u = 0; u < TQ; ++u) o[u][dt] = mfma16(vfa[dt], pf[u], o[u][dt]);
            if (tail) {
#pragma unroll
                for (int dt = H0; dt < DT; ++dt) vfb[dt - H0] = attn_mask_keys(vfb[dt - H0], nvalid);
            }
#pragma unroll
            for (int dt = H0; dt < DT; ++dt)
#pragma unroll
                for (int u = 0; u < TQ; ++u) o[u][dt] = mfma16(vfb[dt - H0], pf[u], o[u][dt]);
        }
#ifdef UMV_ATTN_TRACE
        ATTN_STAMP(ts4);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts0), "+s"(ts1), "+s"(ts2), "+s"(ts3), "+s"(ts4)::"memory");
        if (a.workspace && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && sidx >= 4 && sidx < 20 && lane == 0) {
            uint64_t* tr = reinterpret_cast<uint64_t*>(a.workspace) + ((blockIdx.x * 4 + wave) * 16 + (sidx - 4)) * 5;
            tr[0] = ts0; tr[1] = ts1; tr[2] = ts2; tr[3] = ts3; tr[4] = ts4;
        }
#endif
    }
#if UMV_ATTN_PAIR_DEBUG
    if constexpr (LAZY != 0 && TQ == 2) {      // debug builds: the per-lane softmax state at the end of the key loop -> workspace [wg][wave][u][lane][4]
        if (a.workspace && a.nsplit == 1) {
            const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            float* dbg = reinterpret_cast<float*>(a.workspace) + ((int64_t)(wg * 4 + wave) * TQ) * 64 * 4;
#pragma unroll
            for (int u = 0; u < TQ; ++u) {
                float* d4 = dbg + (u * 64 + lane) * 4;
                d4[0] = nm_run[u]; d4[1] = thr_run[u]; d4[2] = l_run[u]; d4[3] = o[u][0].x;
            }
        }
    }
#endif
    // LAZY: the row sums were kept as per-lane partials; reduced here, in front of the first divergent store
    if constexpr (LAZY != 0) {
#pragma unroll
        for (int u = 0; u < TQ; ++u) l_run[u] = xor32_sum(xor16_sum(l_run[u]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        if (!rvalid[u]) continue;
        const float inv = l_run[u] > 0.f ? 1.0f / l_run[u] : 0.f;
        bf16_t* op = a.out + ((int64_t)(q0 + qi[u]) * a.nq + head[u]) * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g * 4;
            if (d + 3 < HD) {
                u32x2 pk;
                pk.x = pack2bf(o[u][dt].x * inv, o[u][dt].y * inv);
                pk.y = pack2bf(o[u][dt].z * inv, o[u][dt].w * inv);
                *reinterpret_cast<u32x2*>(op + d) = pk;
            }
        }
    }
    if constexpr (STATS) {
        if (a.stats && lane == 0) {
            if (cnt[0]) atomicAdd(a.stats + 0, cnt[0]);
            if (cnt[1]) atomicAdd(a.stats + 1, cnt[1]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------- host side
// The process-wide policy is read from the environment ONCE (thread-safe initialisation of a function-local static, immutable
// afterwards): UMV_ATTN_SHARED=0 = the per-wave streaming kernel everywhere, UMV_ATTN_TQ=1|2 pins the q-tiles per wave,
// UMV_ATTN_DENSE=0 = whole-token tiles, UMV_ATTN_LAZY=0|1|2 = the softmax form.  A call can replace it for itself through
// umv_attn_args.variant (UMV_ATTN_VARIANT_FORCE | ...: the parity tests drive every kernel variant in one process that way).
struct AttnPolicy { int shared, tq, dense, lazy; };
static const AttnPolicy& attn_env_policy() {
    static const AttnPolicy p = [] {
        AttnPolicy q;
        const char* e = getenv("UMV_ATTN_SHARED"); q.shared = (e && atoi(e) == 0) ? 0 : 1;
        e = getenv("UMV_ATTN_TQ"); q.tq = e ? atoi(e) : 0;
        e = getenv("UMV_ATTN_DENSE"); q.dense = e ? atoi(e) : 1;
        e = getenv("UMV_ATTN_LAZY"); q.lazy = e ? atoi(e) : 1;
        if (q.lazy < 0 || q.lazy > 2) q.lazy = 1;
        return q;
    }();
    return p;
}
static AttnPolicy attn_policy(int variant) {
    if (!(variant & UMV_ATTN_VARIANT_FORCE)) return attn_env_policy();
    AttnPolicy q;
    q.shared = (variant & UMV_ATTN_VARIANT_STREAM) ? 0 : 1;
    q.tq = (variant & UMV_ATTN_VARIANT_TQ1) ? 1 : (variant & UMV_ATTN_VARIANT_TQ2) ? 2 : 0;
    q.dense = (variant & UMV_ATTN_VARIANT_WHOLE_TOKENS) ? 0 : 1;
    q.lazy = (variant & UMV_ATTN_VARIANT_EXACT) ? 0 : (variant & UMV_ATTN_VARIANT_PAIR) ? 2 : 1;
    return q;
}
bool umv_attn_prefill_enabled(int variant) { return attn_policy(variant).shared != 0; }

template <int HD, int TQ, int LAZY, bool STATS = false, bool PAGED = false>
static int launch_prefill(const umv_attn_args& a, int qtiles, float scale_log2e, int dense, hipStream_t s) {
    constexpr int KS = (HD + 31) / 32, DT = (HD + 15) / 16;
    constexpr int lds = 2 * ATTN_PREFILL_NB(HD, TQ) * (2 * KS + DT) * 1024;
    static bool attr[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prefill_kernel<HD, TQ, LAZY, STATS, PAGED>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid((qtiles + 4 * TQ - 1) / (4 * TQ), a.nkv, a.nseg);
    hipLaunchKernelGGL((attn_prefill_kernel<HD, TQ, LAZY, STATS, PAGED>), grid, dim3(256), lds, s, a, scale_log2e, dense);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// two q-tiles per wave only when that still leaves >= 2 workgroups per CU (measured: LLM prefill 496 -> 413 us, flow pass
// 96 -> 83, ViT 152 -> 144; but 34-token text prefill 52 -> 73 us with only 96 workgroups)
static bool prefill_two_qtiles(const AttnPolicy& p, int qtiles, int nkv, int nseg) {
    return p.tq == 2 || (p.tq != 1 && (long)((qtiles + 7) / 8) * nkv * nseg >= 448);      // (448 = 512 x 14 / 16: the flow pass of B = 4 text-to-image, 480 workgroups of dense tiles, keeps TQ = 2: 29 vs 33 us)
}

// q-tiles of the LDS-shared kernels: dense packing of the (token, head) pairs when 16 is not a multiple of the group size (G = 7: 16
// pairs per tile instead of 14); UMV_ATTN_DENSE=0 keeps whole tokens per tile (A/B only; same bits either way)
static bool attn_dense(const AttnPolicy& p, int G) { return p.dense != 0 && (16 % G) != 0 && G < 16; }
static int prefill_qtiles(const AttnPolicy& p, int max_q, int G) {
    const int QPT = 16 / G > 0 ? 16 / G : 1;
    return attn_dense(p, G) ? (max_q * G + 15) / 16 : (max_q + QPT - 1) / QPT;
}

// Which prefill-attention kernel umv_attn_varlen sends a (nsplit = 1) call to under the process policy: 0 = the per-wave streaming
// attn_kernel, 1 / 2 = attn_prefill_kernel<hd, TQ>.  Exported so that tests can assert that a case reaches the TQ = 2 kernels.
extern "C" int umv_attn_prefill_tq(int nseg, int nq, int nkv, int hd, int max_q) {
    if (nkv <= 0 || nq % nkv || nseg <= 0 || max_q <= 0) return 0;
    const AttnPolicy& p = attn_env_policy();
    const int G = nq / nkv;
    const int QPT = 16 / G > 0 ? 16 / G : 1;
    if (!((max_q + QPT - 1) / QPT >= 4 && (hd == 128 || hd == 72) && p.shared)) return 0;
    return prefill_two_qtiles(p, prefill_qtiles(p, max_q, G), nkv, nseg) ? 2 : 1;
}

// The LDS-shared kernels address K and V^T through buffer resources: 32-bit byte offsets and a range check that is clamped to
// 2^31 - 1 bytes.  A (segment, kv head) whose K rows or V^T rows span more than that cannot be taken (keys beyond the clamp would read
// as zeros, unmasked): umv_attn_varlen keeps such a call on the per-wave kernel, which uses 64-bit addresses.
bool umv_attn_prefill_can_take(const umv_attn_args& a) {
    if (a.page_table)       // paged pools: 32-bit DMA offsets reach 2 GiB from the pool base - the caller promises page ids below that (kvcache.PagedCache
        return a.hd == 128 && a.k_key_stride == 0 && attn_policy(a.variant).lazy == 1 && !a.stats;      // checks); hd 128 lazy kernels only
    const int64_t kstride = a.k_key_stride ? a.k_key_stride : a.hd;
    const int64_t kbytes = ((int64_t)a.max_kv + 64) * kstride * 2;      // (+ the blocks a stage may run past the last key)
    const int64_t vbytes = (int64_t)a.hd * a.v_d_stride * 2;
    return kbytes < (int64_t)0x7FFFFFFF && vbytes < (int64_t)0x7FFFFFFF;
}

int umv_attn_prefill_launch(const umv_attn_args& a, int /*qtiles of the per-wave kernel*/, float scale_log2e, hipStream_t s) {
    const AttnPolicy p = attn_policy(a.variant);
    const int G = a.nq / a.nkv;
    const int dense = attn_dense(p, G) ? 1 : 0;
    const int qtiles = prefill_qtiles(p, a.max_q, G);
    const bool two = prefill_two_qtiles(p, qtiles, a.nkv, a.nseg);
    if (a.page_table)
        return two ? launch_prefill<128, 2, 1, false, true>(a, qtiles, scale_log2e, dense, s) : launch_prefill<128, 1, 1, false, true>(a, qtiles, scale_log2e, dense, s);
    if (a.stats && p.lazy == 1) {      // the counting instantiations of the shipped (per-tile lazy) kernels
        if (a.hd == 128) return two ? launch_prefill<128, 2, 1, true>(a, qtiles, scale_log2e, dense, s) : launch_prefill<128, 1, 1, true>(a, qtiles, scale_log2e, dense, s);
        return two ? launch_prefill<72, 2, 1, true>(a, qtiles, scale_log2e, dense, s) : launch_prefill<72, 1, 1, true>(a, qtiles, scale_log2e, dense, s);
    }
    const int key = (a.hd == 128 ? 0 : 8) + (two ? 4 : 0) + p.lazy;
    switch (key) {
        case 0: return launch_prefill<128, 1, 0>(a, qtiles, scale_log2e, dense, s);
        case 1: case 2: return launch_prefill<128, 1, 1>(a, qtiles, scale_log2e, dense, s);     // (one tile per wave: the paired form is the per-tile form)
        case 4: return launch_prefill<128, 2, 0>(a, qtiles, scale_log2e, dense, s);
        case 5: return launch_prefill<128, 2, 1>(a, qtiles, scale_log2e, dense, s);
        case 6:
#if UMV_ATTN_PAIR_DEBUG
            switch ((a.variant >> 8) & 7) {          // debug builds only: bits 8..10 of variant pick a bisecting form of the paired kernel
                case 1: return launch_prefill<128, 2, 3>(a, qtiles, scale_log2e, dense, s);
                case 2: return launch_prefill<128, 2, 4>(a, qtiles, scale_log2e, dense, s);
                case 3: return launch_prefill<128, 2, 5>(a, qtiles, scale_log2e, dense, s);
                case 4: return launch_prefill<128, 2, 6>(a, qtiles, scale_log2e, dense, s);
                case 5: return launch_prefill<128, 2, 7>(a, qtiles, scale_log2e, dense, s);
                case 6: return launch_prefill<128, 2, 8>(a, qtiles, scale_log2e, dense, s);
                default: return launch_prefill<128, 2, 2>(a, qtiles, scale_log2e, dense, s);
            }
#else
            return launch_prefill<128, 2, 1>(a, qtiles, scale_log2e, dense, s);      // the paired form exists in UMV_ATTN_PAIR_DEBUG builds only (see the note in the kernel)
#endif
        case 8: return launch_prefill<72, 1, 0>(a, qtiles, scale_log2e, dense, s);
        case 9: case 10: return launch_prefill<72, 1, 1>(a, qtiles, scale_log2e, dense, s);
        case 12: return launch_prefill<72, 2, 0>(a, qtiles, scale_log2e, dense, s);
        default: return launch_prefill<72, 2, 1>(a, qtiles, scale_log2e, dense, s);     // (13, 14: the paired form needs more than the 128 registers of four waves per SIMD at hd 72 - it exists for hd 128 only)
    }
}
