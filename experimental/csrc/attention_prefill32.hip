// Prefill attention on the 32x32x16 matrix instruction (gfx950): the nsplit == 1, hd 128 path of umv_attn_varlen
// (flash_attn_varlen_func at qwen2_navit.py:605-614 - image-span / text prefill, the flow passes of generate_image).
//
// attn_prefill_kernel (attention_prefill.hip) builds S^T from 16x16x32 MFMAs: every q column's 32 scores of a key block sit
// in four lanes, so each block pays two cross-lane maxima and two cross-lane sums per q-tile on top of 8 VALU per score, and
// a wave issues 16 small MFMAs per 16 columns - the PMC counters read 7.9 VALU instructions per MFMA, MfmaUtil 19 %.
// Here one wave owns 32 q columns:
//   * S^T[32 keys][32 q] = K Q^T is 8 v_mfma_f32_32x32x16_bf16 (A = a K fragment from LDS, B = the wave's Q fragment in
//     registers); a lane then holds 16 of its column's 32 scores, the other 16 are in lane ^ 32: ONE permlane32 swap for
//     the running maximum, and the running sum stays a per-lane partial until the epilogue;
//   * the K rows of a block are gathered with key bits 2 / 3 swapped, so that after v_cvt_pk_bf16_f32 registers 0..7 and
//     8..15 of a lane ARE the B fragments (8 consecutive keys) of the two 16-key slabs of P^T - no shuffle between the
//     two GEMMs, and the V^T fragments are plain 16-byte runs of the V^T slab;
//   * O^T[128 d][32 q] += V^T P^T is 8 more MFMAs (4 d-blocks x 2 slabs), accumulators 64 VGPRs per wave.
// 16 MFMAs of 32 cycles per 32 x 32 block against ~80 VALU instructions.  K / V^T blocks reach LDS by LDS-DMA in fragment
// order (every lane supplies its own source address, destination lane-linear: conflict-free ds_read_b128), three stages,
// one barrier per 32-key block, counted vmcnt waits; NW = 4 or 8 waves share a stage (8 when the grid is large enough).
// Per q column the arithmetic does not depend on NW or on what the other columns hold: results are bit-identical across
// the variants and independent of the batch composition.
#include "common.h"
#include "unimedvl_hip.h"
#include "unimedvl_hip_experimental.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void* ap32_lds_ptr_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __attribute__((aligned(16))) uint32_t g_ap32_zero_page[4] = {0, 0, 0, 0};

// The MFMAs are inline asm with the accumulator tied in place: with the builtin, the two code paths of the loop below (pipelined /
// straight) let the register allocator put O's 64 registers somewhere else on one path and copy them back at the merge - 64
// v_mov_b64 per key block.  What the compiler then no longer covers: (a) MFMA -> VALU read of the result needs 18 wait states
// (16-pass MFMA): every such read sits behind the top of the next iteration (wait + barrier + >= 40 staging instructions) or
// behind the explicit s_nop before the epilogue; (b) MFMA -> dependent MFMA on the same accumulator is interlocked by hardware;
// (c) a VALU write of an operand (the zero-initialised / rescaled O, the packed P) right in front of the MFMA needs 2 wait
// states: the s_nop 1 that opens every string (without it: NaN in O's first row block on some waves of some launches).
__device__ __forceinline__ void mfma32_acc(f32x16& acc, bf16x8 a, bf16x8 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_new(f32x16& acc, bf16x8 a, bf16x8 b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}

// s_barrier as an asm with a memory clobber: the builtin does not stop the compiler from hoisting an LDS read of the block the
// barrier publishes to above it (seen: the first K / V^T fragment reads of an iteration, i.e. stale fragments now and then)
__device__ __forceinline__ bf16x8 ap32_mask_keys(bf16x8 v, int nvalid) {   // keep the first nvalid (0..8) elements
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = j < nvalid ? v[j] : (short)0;
    return o;
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void ap32_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        ap32_for<B + 1, E>(f);
    }
}

// The online softmax of one 32-key block of a q column, cut into 16 slices so that the pipelined loop can place one slice
// behind every MFMA (slices 0-3: running maximum of the raw scores; 4: the column maximum across the two half-waves, the new
// running maximum and alpha; 5-12: two weights each - exp2(s * scale - m), partial row sum, bf16 pair; 13: l and m).
// The straight path calls the slices back to back: same instructions, same order, same bits.
constexpr float AP32_LAZY = 6.0f;
struct Ap32Softmax {
    float mx, m_new, m_use, alpha, ps;
    template <int Q>
    __device__ __forceinline__ void slice(const f32x16& st, uint32_t (&pk)[8], float& m_run, float& l_run, float scale_log2e) {
        if constexpr (Q < 4) {
            if constexpr (Q == 0) mx = -INFINITY;
            mx = fmaxf(fmaxf(mx, st[4 * Q]), fmaxf(st[4 * Q + 1], fmaxf(st[4 * Q + 2], st[4 * Q + 3])));
        } else if constexpr (Q == 4) {
            mx = xor32_max(mx) * scale_log2e;                 // scale > 0: the maximum of the scaled scores
            // The reference point of the exponentials only moves when the column maximum has grown by more than 2^AP32_LAZY
            // (a per-column decision: no other column enters it).  Any reference point gives the same quotient; the weights of
            // a block are then <= 2^AP32_LAZY instead of <= 1 - bf16 keeps the same relative precision, the sums are fp32 -
            // and O is rescaled a few times per column instead of at almost every block of a 32-column tile.
            const float cand = fmaxf(m_run, mx);
            const bool grow = !(cand - m_run <= AP32_LAZY);   // also true for m_run = -inf (first block)
            m_new = grow ? cand : m_run;
            m_use = (m_new == -INFINITY) ? 0.f : m_new;
            alpha = grow ? ((m_run == -INFINITY) ? 0.f : umv_exp2(m_run - m_use)) : 1.0f;
            ps = 0.f;
        } else if constexpr (Q < 13) {
            constexpr int r = 2 * (Q - 5);
            const float p0 = umv_exp2(__builtin_fmaf(st[r], scale_log2e, -m_use));       // exp2(-inf) = 0
            const float p1 = umv_exp2(__builtin_fmaf(st[r + 1], scale_log2e, -m_use));
            ps += p0;
            ps += p1;
            pk[r >> 1] = pack2bf(p0, p1);
        } else if constexpr (Q == 13) {
            l_run = l_run * alpha + ps;
            m_run = m_new;
        }
    }
};

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_prefill32_kernel(umv_attn_args a, float scale_log2e) {
    constexpr int KC = HD / 16;          // k-steps of S^T = K Q^T
    constexpr int DB = HD / 32;          // 32-row blocks of O^T
    constexpr int FK = KC, FV = DB * 2, FB = FK + FV;   // 1 KiB fragments per 32-key block: K (k-step kc), V^T (slab, d-block)
    constexpr int PPW = FB / NW;                          // DMA pieces per wave and group
    constexpr int KRING = 0, VRING = 3 * FK * 1024, DUMP = VRING + 3 * FV * 1024;
    static_assert(FB % NW == 0 && FK == FV, "fragments must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // K ring (3 blocks), V^T ring (3 blocks), 1 KiB per wave for idle pieces
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int G = a.nq / a.nkv;
    const int QPT = 32 / G;              // q positions per 32-column tile (G <= 32)
    const int s = blockIdx.z;
    const int kh = blockIdx.y;
    const int qt_wg = blockIdx.x * NW;
    const int q0 = a.cu_q[s];
    const int Lq = a.cu_q[s + 1] - q0;
    const int Lk = a.kv_len[s];
    if (qt_wg * QPT >= Lq || Lk <= 0) return;   // uniform over the workgroup

    const int ql = c / G, hg = c - ql * G;
    const int head = kh * G + hg;
    const int qt = qt_wg + wave;
    const bool active = qt * QPT < Lq;
    const int qi = qt * QPT + ql;
    const bool rvalid = active && (c < G * QPT) && (qi < Lq);
    const int limit = a.causal ? (Lk - Lq + qi) : (Lk - 1);              // bottom-right aligned causal mask
    const int min_limit = a.causal ? (Lk - Lq + qt * QPT) : (Lk - 1);
    int my_end = Lk;
    if (a.causal) my_end = min(Lk, Lk - Lq + min(Lq - 1, qt * QPT + QPT - 1) + 1);
    if (!active) my_end = 0;
    int blk_end = Lk;
    if (a.causal) {
        const int last_q = min(Lq - 1, (qt_wg + NW - 1) * QPT + QPT - 1);
        blk_end = min(Lk, Lk - Lq + last_q + 1);
    }
    const int nblk = (blk_end + 31) / 32;       // key blocks of the workgroup ...
    const int my_nblk = (my_end + 31) / 32;     // ... and of this wave's tile (causal: earlier tiles end earlier)

    // ---- Q fragments (B operand of K Q^T): lane (column c, half h) holds dims 16 kc + 8 h .. + 8 of its q row
    bf16x8 qf[KC];
    const bf16_t* qp = a.q + (int64_t)(q0 + (rvalid ? qi : 0)) * (a.q_row_stride ? a.q_row_stride : (int64_t)a.nq * HD) + head * HD + 8 * h;

    // ---- staging.  Group g = {K block g + 1, V^T block g - 1}: what iteration g of the pipelined loop reads.  Every LDS-DMA
    // instruction copies 1 KiB that is (nearly) contiguous in memory - gathering the MFMA fragments directly (16 bytes per lane
    // from 32 different rows) costs one L1 access per lane, and the L1 tag rate then bounds the whole kernel (PMC: 58 cache
    // accesses per DMA instruction, ~250 us on the 8 x 1026 span whatever the MFMA / VALU schedule):
    //   * K block: LDS image [32 keys][16 chunks of 16 bytes]; piece i = keys 4 i .. 4 i + 3 (four whole 256-byte rows);
    //   * V^T block: LDS image [HD rows][4 chunks]; piece i = rows 16 i .. 16 i + 15, 64 bytes of each.
    // The fragment reads then walk the images with a row stride of 256 / 64 bytes; bank conflicts are avoided by XOR-ing the
    // chunk index with (key & 15) resp. ((d >> 2) & 3) - on the SOURCE side, the DMA destination being lane-linear.  A block that
    // does not exist is replaced by a copy of the zero page into the wave's idle slot: every group is PPW pieces per wave, so
    // the counted waits hold.
    const int64_t kstride = a.k_key_stride ? a.k_key_stride : HD;
    const bf16_t* kbase = a.k_slab + (a.k_key_stride ? (int64_t)q0 * a.k_key_stride : s * a.k_seg_stride) + kh * a.k_head_stride;
    const bf16_t* vbase = a.vt_slab + s * a.v_seg_stride + kh * a.v_head_stride;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_ap32_zero_page);
    const int cap = (int)a.v_d_stride;
    char* const dump = smem + DUMP + wave * 1024;
    const int krow_l = lane >> 4, kchunk_l = lane & 15;      // K piece: this lane's row within the piece / destination chunk
    const int vrow_l = lane >> 2, vchunk_l = lane & 3;       // V^T piece
    // Running source pointers (one 64-bit add per piece and block); only a block that straddles Lk / the slab capacity
    // recomputes its addresses with the clamps (wave-uniform branch).  issue(g, KS, VS): KS / VS = ring slots of K block g + 1 and
    // V^T block g - 1, compile-time in the loop (its body is unrolled over the ring period).
    const bf16_t* kp[PPW];
    const bf16_t* vp[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int f = wave + i * NW;
        const int key = 4 * f + krow_l;                               // K piece f: key 0..31 within a block
        kp[i] = kbase + (int64_t)key * kstride + 8 * (kchunk_l ^ (key & 15));
        const int d = 16 * (f - FK) + vrow_l;                         // V^T piece f - FK
        vp[i] = vbase + (int64_t)d * a.v_d_stride + 8 * (vchunk_l ^ ((d >> 2) & 3)) - 64;   // block -2 (never read)
    }
    auto issue = [&](int g, auto KS, auto VS) {
        constexpr int ks = decltype(KS)::value, vs = decltype(VS)::value;
        const int kblk = g + 1, vblk = g - 1;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int f = wave + i * NW;
            const bf16_t* p = zero;
            char* dst = dump;
            if (f < FK) {
                if (kblk < nblk) {
                    p = kp[i];
                    if (kblk * 32 + 32 > Lk) {                            // the segment's last block: rows past Lk - 1 read row Lk - 1
                        const int key = 4 * f + krow_l;
                        p = kbase + (int64_t)min(kblk * 32 + key, Lk - 1) * kstride + 8 * (kchunk_l ^ (key & 15));
                    }
                    dst = smem + KRING + (ks * FK + f) * 1024;
                }
                kp[i] += 32 * kstride;
            } else {
                const int v = f - FK;
                if (vblk >= 0 && vblk < nblk) {
                    p = vp[i];
                    if (vblk * 32 + 32 > cap) {                           // columns past the slab: the zero page
                        const int d = 16 * v + vrow_l;
                        if (vblk * 32 + 8 * (vchunk_l ^ ((d >> 2) & 3)) + 8 > cap) p = zero;
                    }
                    dst = smem + VRING + (vs * FV + v) * 1024;
                }
                vp[i] += 32;
            }
            __builtin_amdgcn_global_load_lds((const void*)p, (ap32_lds_ptr_t)dst, 16, 0, 0);
        }
    };
    // fragment reads.  K, k-step kc: lane (row rho = c, half h) reads dims 16 kc + 8 h .. of key pi(rho) (pi = swap of bits 2 / 3:
    // the register order of S^T then is the key order P^T needs) = chunk 2 kc + h of LDS row pi(rho);  V^T, (sl, db): lane reads
    // keys 16 sl + 8 h .. of row d = 32 db + c = chunk 2 sl + h.
    const int pi = (c & 0x13) | ((c & 4) << 1) | ((c & 8) >> 1);
    int koff[KC], voff[2];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) koff[kc] = pi * 256 + (((2 * kc + h) ^ (pi & 15)) << 4);
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) voff[sl] = c * 64 + (((2 * sl + h) ^ ((c >> 2) & 3)) << 4);      // + db * 2048

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;   // l_run: this lane's half of the row sum
    float alpha_pending = 1.0f;             // alpha of the block whose P V has not been added yet

    // the V^T ring starts as zeros: iteration 0 multiplies "block -1" (P = 0) with slot 2, and the idle iterations that round the
    // loop up to the ring period do the same with slots no block may ever have reached
    for (int t = threadIdx.x; t < 3 * FV * 64; t += NW * 64) *reinterpret_cast<u32x4*>(smem + VRING + t * 16) = (u32x4){0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    issue(-1, I0{}, I0{});
    issue(0, I1{}, I0{});
    issue(1, I2{}, I0{});
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[kc] = ldg_frag(qp + 16 * kc);     // always a readable row (row 0 of the segment for idle columns)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        if (!rvalid) qf[kc] = zero_frag();
        asm volatile("" : "+v"(qf[kc]));
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    const char* const lk = smem + KRING;
    const char* const lv = smem + VRING;
    auto qk0 = [&](f32x16& st) {                 // S^T of key block 0 (ring slot 0)
        const char* kb_ = lk;
        mfma32_new(st, *reinterpret_cast<const bf16x8*>(kb_ + koff[0]), qf[0]);
#pragma unroll
        for (int kc = 1; kc < KC; ++kc) mfma32_acc(st, *reinterpret_cast<const bf16x8*>(kb_ + koff[kc]), qf[kc]);
    };
    auto mask_scores = [&](int blk, f32x16& st) {   // register r <-> key kb + (r & 7) + 8 h + 16 (r >> 3)
        const int kb = blk * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb + (r & 7) + 8 * h + 16 * (r >> 3);
            st[r] = (key <= limit && key < my_end) ? st[r] : -INFINITY;
        }
    };
    // The last, partial block of a segment: V^T columns past Lk may hold anything (0 x NaN).  The wave that staged a V^T piece
    // zeroes those columns in LDS before the barrier that publishes the block - P V itself is one code path for every block.
    auto mask_v_tail = [&](int vblk, int slot) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int f = wave + i * NW;
            if (f >= FK) {
                const int v = f - FK, d = 16 * v + vrow_l;
                bf16x8* q = reinterpret_cast<bf16x8*>(smem + VRING + (slot * FV + v) * 1024 + lane * 16);
                *q = ap32_mask_keys(*q, min(8, max(0, Lk - (vblk * 32 + 8 * (vchunk_l ^ ((d >> 2) & 3))))));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // Iteration i of the pipelined loop works on three blocks at once - S^T of block i + 1, the softmax of block i, P V of block
    // i - 1 - which do not depend on each other: one softmax slice and one fragment read follow every MFMA, and the MFMAs
    // alternate between the two GEMMs (a dependent accumulator is only touched every other MFMA).  ONE code path for every
    // iteration (two paths that both update O make the register allocator copy O's 64 registers at the merge): a block that
    // does not exist for this wave is neutralised by data - its scores are masked to -inf (weights 0, maximum and sum
    // untouched), S^T of a block past the end is computed from whatever the ring holds and never read, P of "block -1" is 0
    // against a zeroed V^T buffer.
    f32x16 S[3];
    uint32_t P[3][8];
    auto iter = [&](int i, auto R) {          // R = i % 3 (compile time: every ring slot below is an immediate offset)
        constexpr int r = decltype(R)::value, r1 = (r + 1) % 3, r2 = (r + 2) % 3;
        f32x16& s_cur = S[r];
        f32x16& s_nxt = S[r1];
        uint32_t (&p_prev)[8] = P[r2];
        uint32_t (&p_cur)[8] = P[r];
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");      // my pieces of group i have landed (group i + 1 may fly)
        if (i >= 1 && i <= nblk && i * 32 > Lk) mask_v_tail(i - 1, r2);  // wave uniform; at most once per workgroup
        asm volatile("s_barrier" ::: "memory");                         // ... everyone's; and everyone is done with iteration i - 1
        issue(i + 2, std::integral_constant<int, r>{}, std::integral_constant<int, r1>{});   // K block i + 3, V^T block i + 1
        if (__any(alpha_pending != 1.0f)) {                             // O is at the reference point of block i - 2: move it to block i - 1's
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha_pending;
        }
        const int kb = i * 32;
        if (!(kb + 32 <= my_end && kb + 31 <= min_limit)) mask_scores(i, s_cur);
        Ap32Softmax sm;
        const char* kb_ = lk + r1 * FK * 1024;                         // block i + 1
        const char* vb = lv + r2 * FV * 1024;                          // block i - 1
        bf16x8 pf[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            u32x4 t = {p_prev[4 * sl], p_prev[4 * sl + 1], p_prev[4 * sl + 2], p_prev[4 * sl + 3]};
            pf[sl] = __builtin_bit_cast(bf16x8, t);
        }
        bf16x8 fk[2], fv[2];
        fk[0] = *reinterpret_cast<const bf16x8*>(kb_ + koff[0]);
        fv[0] = *reinterpret_cast<const bf16x8*>(vb + voff[0]);
        __builtin_amdgcn_sched_barrier(0);
        ap32_for<0, 16>([&](auto K) {
            constexpr int k = decltype(K)::value, j = k >> 1;
            if constexpr (k + 2 < 16) {      // the fragment of MFMA k + 2
                constexpr int j2 = (k + 2) >> 1;
                if constexpr ((k & 1) == 0) fk[j2 & 1] = *reinterpret_cast<const bf16x8*>(kb_ + koff[j2]);
                else fv[j2 & 1] = *reinterpret_cast<const bf16x8*>(vb + voff[j2 / DB] + (j2 % DB) * 2048);
            }
            if constexpr (k == 0) mfma32_new(s_nxt, fk[0], qf[0]);
            else if constexpr ((k & 1) == 0) mfma32_acc(s_nxt, fk[j & 1], qf[j]);
            else mfma32_acc(o[j % DB], fv[j & 1], pf[j / DB]);
            sm.template slice<k>(s_cur, p_cur, m_run, l_run, scale_log2e);
            __builtin_amdgcn_sched_barrier(0);
        });
        alpha_pending = sm.alpha;
    };

#pragma unroll
    for (int r = 0; r < 8; ++r) P[0][r] = P[1][r] = P[2][r] = 0u;
    qk0(S[0]);
    for (int i = 0; i <= nblk; i += 3) {      // whole ring periods: up to two idle iterations at the end (everything masked)
        iter(i, I0{});
        iter(i + 1, I1{});
        iter(i + 2, I2{});
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 3" ::: "memory");     // the idle pieces of the last groups; MFMA -> VALU wait states
#pragma unroll
    for (int db = 0; db < DB; ++db) asm volatile("" : "+v"(o[db]));
    if (!rvalid) return;
    const float l = xor32_sum(l_run);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16_t* op = a.out + ((int64_t)(q0 + qi) * a.nq + head) * HD + 4 * h;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {      // registers 4 g4 .. + 4 <-> d = 32 db + 8 g4 + 4 h + 0..3
            u32x2 w;
            w.x = pack2bf(o[db][4 * g4] * inv, o[db][4 * g4 + 1] * inv);
            w.y = pack2bf(o[db][4 * g4 + 2] * inv, o[db][4 * g4 + 3] * inv);
            *reinterpret_cast<u32x2*>(op + 32 * db + 8 * g4) = w;
        }
}

template <int HD, int NW>
static int launch_prefill32(const umv_attn_args& a, int qtiles, float scale_log2e, hipStream_t s) {
    constexpr int lds = 3 * (HD / 16 + HD / 32 * 2) * 1024 + NW * 1024;
    static bool attr[UMV_MAX_DEVICES] = {};
    if (umv_first_on_device(attr))
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_prefill32_kernel<HD, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    dim3 grid((qtiles + NW - 1) / NW, a.nkv, a.nseg);
    hipLaunchKernelGGL((attn_prefill32_kernel<HD, NW>), grid, dim3(NW * 64), lds, s, a, scale_log2e);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}

// EXPERIMENTAL entry point (include/unimedvl_hip_experimental.h): same arguments as umv_attn_varlen, nsplit = 1, hd = 128 only.
// UMV_ATTN32_NW=4|8 pins the waves per workgroup (A/B only).
extern "C" int umv_attn_prefill32(const umv_attn_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap, UMV_ERR_ARG, "attn32: null args");
    const umv_attn_args& a = *ap;
    UMV_CHECK(a.q && a.out && a.cu_q && a.kv_len && a.k_slab && a.vt_slab, UMV_ERR_ARG, "attn32: null pointer");
    UMV_CHECK(a.hd == 128 && a.nsplit == 1, UMV_ERR_UNSUPPORTED, "attn32: hd = 128, nsplit = 1 only (hd %d, nsplit %d)", a.hd, a.nsplit);
    UMV_CHECK(a.nkv > 0 && a.nq % a.nkv == 0 && a.nq / a.nkv <= 32, UMV_ERR_ARG, "attn32: bad head counts nq=%d nkv=%d", a.nq, a.nkv);
    UMV_CHECK((a.v_d_stride % 8) == 0 && (a.q_row_stride % 8) == 0 && (a.k_key_stride % 8) == 0 && (a.k_head_stride % 8) == 0, UMV_ERR_ARG,
              "attn32: strides must be multiples of 8 elements");
    if (a.nseg == 0 || a.max_q == 0) return UMV_OK;
    const int G = a.nq / a.nkv;
    const int QPT = 32 / G;
    const int qtiles = (a.max_q + QPT - 1) / QPT;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)a.hd);
    static int nw_env = -1;
    if (nw_env < 0) { const char* e = getenv("UMV_ATTN32_NW"); nw_env = e ? atoi(e) : 0; }
    // 8 waves share a K / V^T stage when that still gives every CU two workgroups
    const bool eight = nw_env == 8 || (nw_env != 4 && (long)((qtiles + 7) / 8) * a.nkv * a.nseg >= 512);
    hipStream_t s = (hipStream_t)stream;
    return eight ? launch_prefill32<128, 8>(a, qtiles, scale_log2e, s) : launch_prefill32<128, 4>(a, qtiles, scale_log2e, s);
}
