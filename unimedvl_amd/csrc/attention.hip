// Varlen GQA attention over KV slabs (flash-style, online softmax, fp32 statistics).
//
// One wavefront owns a tile of 16 "rows".  A row is a (query, q-head-of-the-GQA-group)
// pair, so the G = nq/nkv heads that share a KV head share every K/V fragment load:
//   prefill, G=7 : 2 queries x 7 heads per tile        decode (Lq=1): the 7 heads
//   ViT,     G=1 : 16 queries per tile
// Per 32-key block (MFMA v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   S^T[key][row] = K . Q^T   A = K rows (key order permuted so that each lane ends up
//                             with 8 CONSECUTIVE keys), B = Q^T (held in registers)
//   O^T[d][row]  += V^T . P^T A = V^T fragment = 16 B contiguous along keys (the slab keeps
//                             V transposed), B = P^T = the lane's own 8 probabilities
// so P never moves between lanes, the O rescale factor is lane-local, and no LDS is used.
// nsplit > 1 splits the key range over workgroups (decode) and a combine kernel merges.
#include "common.h"
#include "attention_combine.h"
#include "../../include/unimedvl_hip.h"

// attention_prefill.hip: the nsplit == 1 path with K / V^T shared through LDS (bit-identical results)
bool umv_attn_prefill_enabled(int variant);
bool umv_attn_prefill_can_take(const umv_attn_args& a);
int umv_attn_prefill_launch(const umv_attn_args& a, int qtiles, float scale_log2e, hipStream_t s);

__device__ __forceinline__ bf16x8 mask_keys(bf16x8 v, int nvalid) {
    // keep the first nvalid (0..8) bf16 elements, zero the rest
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = j < nvalid ? v[j] : (short)0;
    return o;
}

// WS > 1 (decode: ONE q-tile per (segment, kv head, split)): the workgroup's waves all own that q-tile and split the keys of the
// workgroup's range among themselves in runs of 32-key blocks; their (O, m, l) triples meet in LDS (merge in wave order 0 .. wsplit-1,
// wave w finishing the d-tiles dt = w, w + wsplit, ..) and the workgroup writes ONE partial (or the final rows).  PAGED = the K / V^T
// operands are page pools addressed through umv_attn_args.page_table (pages of UMV_KV_PAGE = 256 keys; a 32-key block never straddles one).
template <int HD, int WS, bool PAGED>
__global__ __launch_bounds__(256) void attn_kernel(umv_attn_args a, float scale_log2e) {
    constexpr int wsplit = WS;
    constexpr int KS = (HD + 31) / 32;   // k-steps over the head dim for S
    constexpr int DT = (HD + 15) / 16;   // 16-wide output d tiles
    extern __shared__ __attribute__((aligned(16))) float attn_sm[];      // wsplit > 1: [wsplit][DT][64] f32x4 + [wsplit][16][2]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int G = a.nq / a.nkv;
    const int QPT = 16 / G > 0 ? 16 / G : 1;
    const int s = blockIdx.z;
    const int kh = blockIdx.y % a.nkv;
    const int split = blockIdx.y / a.nkv;
    const int qt = wsplit > 1 ? (int)blockIdx.x : blockIdx.x * (int)(blockDim.x >> 6) + wave;      // 1..4 q-tiles (waves) per workgroup, see the launcher
    const int q0 = a.cu_q[s];
    const int Lq = a.cu_q[s + 1] - q0;
    const int Lk = a.kv_len[s];
    if (qt * QPT >= Lq) return;

    const int ql = j / G, hg = j % G;
    const int qi = qt * QPT + ql;
    const bool rvalid = (j < G * QPT) && (qi < Lq);
    const int head = kh * G + hg;
    // causal = bottom-right aligned: query qi sees keys <= Lk - Lq + qi
    const int limit = a.causal ? (Lk - Lq + qi) : (Lk - 1);

    // Q fragments (B operand): lane (j,g) holds Q[row j][ks*32 + g*8 .. +8]
    bf16x8 qf[KS];
    {
        const bf16_t* qp = a.q + (int64_t)(q0 + (rvalid ? qi : 0)) * (a.q_row_stride ? a.q_row_stride : (int64_t)a.nq * HD) + head * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            qf[ks] = (rvalid && d < HD) ? ldg_frag(qp + d) : zero_frag();
        }
    }
    // key range of this split, in 32-key blocks
    int kb_begin = 0, kb_end = Lk;
    if (a.nsplit > 1) {
        int chunk = ((Lk + a.nsplit - 1) / a.nsplit + 31) & ~31;
        kb_begin = split * chunk;
        kb_end = min(Lk, kb_begin + chunk);
    }
    if (a.causal) {  // no key beyond the tile's largest limit
        int last_q = min(Lq - 1, qt * QPT + QPT - 1);
        kb_end = min(kb_end, Lk - Lq + last_q + 1);
    }
    if (wsplit > 1) {     // this wave's run of 32-key blocks inside the workgroup's range
        const int nblk = (max(kb_end - kb_begin, 0) + 31) >> 5;
        const int per = (nblk + wsplit - 1) / wsplit;
        const int b0 = kb_begin + wave * per * 32;
        kb_end = min(kb_end, b0 + per * 32);
        kb_begin = b0;
    }
    const int64_t kstride = a.k_key_stride ? a.k_key_stride : HD;     // packed K (k_key_stride > 0): rows cu_q[s] .. of a [T, ...] buffer
    // slab form: one base per (segment, kv head); paged form: the base of the page holding block kb (key / column offsets then count
    // from the page start: kpage0 = first key of that page)
    // (PAGED is a template parameter: the slab kernels carry no trace of it - a run-time test cost the decode step's attention 9.4 vs 8.9 us)
    const int32_t* ptab = PAGED ? a.page_table + (int64_t)s * a.page_table_stride : nullptr;
    const bf16_t* kbase = a.k_slab + (a.k_key_stride ? (int64_t)q0 * a.k_key_stride : (PAGED ? 0 : s * a.k_seg_stride)) + kh * a.k_head_stride;
    const bf16_t* vbase = a.vt_slab + (PAGED ? 0 : s * a.v_seg_stride) + kh * a.v_head_stride;
    auto k_of = [&](int kb, int& kpage0) -> const bf16_t* {
        if constexpr (!PAGED) { kpage0 = 0; return kbase; }
        else {
            kpage0 = kb & ~(UMV_KV_PAGE - 1);
            return kbase + (int64_t)ptab[kb >> UMV_KV_PAGE_LOG2] * a.k_seg_stride;
        }
    };
    auto v_of = [&](int kb, int& kpage0) -> const bf16_t* {
        if constexpr (!PAGED) { kpage0 = 0; return vbase; }
        else {
            kpage0 = kb & ~(UMV_KV_PAGE - 1);
            return vbase + (int64_t)ptab[kb >> UMV_KV_PAGE_LOG2] * a.v_seg_stride;
        }
    };

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // K fragments of a 32-key block; A row i=(lane&15) of tile t  <->  key kb + (i>>2)*8 + t*4 + (i&3)
    constexpr bool PREFETCH = HD <= 128;   // register budget: K(next) + V(cur) in flight while S/softmax run
    auto load_k = [&](int kb, bf16x8 (&kf)[2][PREFETCH ? KS : 1]) {
        int kp0;
        const bf16_t* kb_base = k_of(kb, kp0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = kb + (j >> 2) * 8 + t * 4 + (j & 3);
            const bf16_t* kp = kb_base + (int64_t)(min(key, Lk - 1) - kp0) * kstride;   // keys past Lk are masked below: any valid row will do
#pragma unroll
            for (int ks = 0; ks < (PREFETCH ? KS : 1); ++ks) {
                const int d = ks * 32 + g * 8;
                kf[t][ks] = (d < HD) ? ldg_frag(kp + d) : zero_frag();
            }
        }
    };
    constexpr int KSP = PREFETCH ? KS : 1, DTP = PREFETCH ? DT : 1;
    bf16x8 kcur[2][KSP], knext[2][KSP];
    if constexpr (PREFETCH) {
        if (kb_begin < kb_end) load_k(kb_begin, kcur);
    }
    for (int kb = kb_begin; kb < kb_end; kb += 32) {
        // issue this block's V^T loads and the next block's K loads before any math
        bf16x8 vf[DTP];
        int vp0;
        const bf16_t* vb_base = v_of(kb, vp0);
        if constexpr (PREFETCH) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + j;
                vf[dt] = (d < HD) ? ldg_frag(vb_base + (int64_t)d * a.v_d_stride + (kb - vp0) + g * 8) : zero_frag();
            }
            if (kb + 32 < kb_end) load_k(kb + 32, knext);
        }
        // ---- S^T = K Q^T
        f32x4 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (PREFETCH) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) st[t] = mfma16(kcur[t][ks], qf[ks], st[t]);
            } else {   // large head_dim: stream the K fragments through the MFMA chain
                const int key = kb + (j >> 2) * 8 + t * 4 + (j & 3);
                int kp0;
                const bf16_t* kp = k_of(kb, kp0) + (int64_t)(min(key, Lk - 1) - kp0) * kstride;   // keys past Lk are masked below: any valid row will do
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d = ks * 32 + g * 8;
                    bf16x8 kf = (d < HD) ? ldg_frag(kp + d) : zero_frag();
                    st[t] = mfma16(kf, qf[ks], st[t]);
                }
            }
        }
        // lane (row j, g) now holds scores of keys kb + g*8 + t*4 + r
        float sc[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + g * 8 + t * 4 + r;
                sc[t * 4 + r] = (key <= limit && key < kb_end) ? st[t][r] : -INFINITY;
            }
        float alpha;
        bf16x8 pf;
        attn_softmax_block(sc, scale_log2e, m_run, l_run, alpha, pf);      // (shared with the exact-maximum prefill kernels, UMV_ATTN_LAZY=0: same bits)
        // ---- O^T += V^T P^T ; A = V^T[d = dt*16 + (lane&15)][kb + g*8 .. +8]
        const bool partial = kb + 32 > Lk;
        const int nvalid = min(8, max(0, Lk - (kb + g * 8)));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x8 v;
            if constexpr (PREFETCH) {
                v = vf[dt];
            } else {
                const int d = dt * 16 + j;
                v = (d < HD) ? ldg_frag(vb_base + (int64_t)d * a.v_d_stride + (kb - vp0) + g * 8) : zero_frag();
            }
            if (partial) v = mask_keys(v, nvalid);
            // (pinning o[] in VGPRs with an asm MFMA, as attention_prefill.hip does, is bit-identical here too but buys nothing:
            // this kernel's waves walk two or three blocks and are latency bound - decode 3.235 vs 3.235 ms, B=32 4.50 vs 4.51)
            f32x4 acc = o[dt];
            acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
            o[dt] = mfma16(v, pf, acc);
        }
        if constexpr (PREFETCH) {
            if (kb + 32 < kb_end) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kcur[t][ks] = knext[t][ks];
            }
        }
    }
    if constexpr (WS > 1) {
        // merge the waves' (O, m, l) in LDS: wave order 0 .. WS-1 (fixed: the result does not depend on which wave finished first)
        f32x4* smO = reinterpret_cast<f32x4*>(attn_sm);                       // [WS][DT][64]
        float* smML = attn_sm + WS * DT * 64 * 4;                             // [WS][16][2]
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) smO[(wave * DT + dt) * 64 + lane] = o[dt];
        if (g == 0) { smML[(wave * 16 + j) * 2] = m_run; smML[(wave * 16 + j) * 2 + 1] = l_run; }
        __syncthreads();
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < WS; ++w) M = fmaxf(M, smML[(w * 16 + j) * 2]);
        float wgt[WS], L = 0.f;
#pragma unroll
        for (int w = 0; w < WS; ++w) {
            const float mw = smML[(w * 16 + j) * 2];
            wgt[w] = (mw == -INFINITY) ? 0.f : umv_exp2(mw - M);
            L += wgt[w] * smML[(w * 16 + j) * 2 + 1];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            if (dt % WS != wave) continue;          // this wave finishes (and stores) the d-tiles dt = wave, wave + WS, ..
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WS; ++w) {
                const f32x4 ow = smO[(w * DT + dt) * 64 + lane];
                acc.x += wgt[w] * ow.x; acc.y += wgt[w] * ow.y; acc.z += wgt[w] * ow.z; acc.w += wgt[w] * ow.w;
            }
            o[dt] = acc;
        }
        m_run = M;
        l_run = L;
    }
    if (!rvalid) return;
    const int64_t tok = q0 + qi;
    if (a.nsplit == 1) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* op = a.out + (tok * a.nq + head) * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g * 4;
            if (d + 3 < HD && (WS == 1 || dt % WS == wave)) {
                u32x2 pk;
                pk.x = pack2bf(o[dt].x * inv, o[dt].y * inv);
                pk.y = pack2bf(o[dt].z * inv, o[dt].w * inv);
                *reinterpret_cast<u32x2*>(op + d) = pk;
            }
        }
    } else {
        float* wsO = reinterpret_cast<float*>(a.workspace);
        const int64_t slotw = (tok * a.nq + head) * a.nsplit + split;
        float* po = wsO + slotw * (HD + 4);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int d = dt * 16 + g * 4;
            if (d + 3 < HD && (WS == 1 || dt % WS == wave)) *reinterpret_cast<f32x4*>(po + d) = o[dt];
        }
        if (g == 0 && (WS == 1 || wave == 0)) { po[HD] = m_run; po[HD + 1] = l_run; }
    }
}


extern "C" size_t umv_attn_workspace_bytes(int nseg, int nq, int hd, int max_q, int nsplit) {
    if (nsplit <= 1) return 0;
    return (size_t)nseg * max_q * nq * nsplit * (hd + 4) * sizeof(float);
}

extern "C" int umv_attn_varlen(const umv_attn_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap, UMV_ERR_ARG, "attn: null args");
    const umv_attn_args& a = *ap;
    UMV_CHECK(a.q && a.out && a.cu_q && a.kv_len && a.k_slab && a.vt_slab, UMV_ERR_ARG, "attn: null pointer");
    UMV_CHECK(a.nkv > 0 && a.nq % a.nkv == 0 && a.nq / a.nkv <= 16, UMV_ERR_ARG, "attn: bad head counts nq=%d nkv=%d", a.nq, a.nkv);
    UMV_CHECK(a.nsplit >= 1 && a.nsplit <= 32 && (a.nsplit == 1 || a.workspace), UMV_ERR_ARG, "attn: nsplit=%d (1..32) needs workspace", a.nsplit);
    UMV_CHECK((a.v_d_stride % 8) == 0, UMV_ERR_ARG, "attn: slab capacity must be a multiple of 8");
    UMV_CHECK(a.q_row_stride >= 0 && a.k_key_stride >= 0 && (a.q_row_stride % 8) == 0 && (a.k_key_stride % 8) == 0 && (a.k_head_stride % 8) == 0,
              UMV_ERR_ARG, "attn: q_row_stride / k_key_stride / k_head_stride must be non-negative multiples of 8 elements (16-byte fragment loads)");
    UMV_CHECK(a.k_key_stride == 0 || (a.nsplit == 1 && !a.causal), UMV_ERR_UNSUPPORTED,
              "attn: packed K (k_key_stride > 0) is the cache-less self-attention form: nsplit = 1, non-causal, kv_len[s] = cu_q[s+1] - cu_q[s]");
    if (a.nseg == 0 || a.max_q == 0) return UMV_OK;
    const int G = a.nq / a.nkv;
    const int QPT = 16 / G > 0 ? 16 / G : 1;
    const int qtiles = (a.max_q + QPT - 1) / QPT;
    // one wave per q-tile, up to four per workgroup: a decode step (max_q = 1) has ONE q-tile per (segment, kv head, split), so its
    // workgroups are single waves (with 256 threads three of the four waves of each of the 544 workgroups only exited)
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)a.hd);
    hipStream_t s = (hipStream_t)stream;
    if (a.page_table) {
        UMV_CHECK(a.page_table_stride > 0 && a.k_key_stride == 0, UMV_ERR_ARG, "attn: page_table needs page_table_stride > 0 and goes with pooled K (no k_key_stride)");
        UMV_CHECK(a.v_d_stride == UMV_KV_PAGE, UMV_ERR_ARG, "attn: paged V^T rows are UMV_KV_PAGE = %d keys long (v_d_stride %lld)", UMV_KV_PAGE, (long long)a.v_d_stride);
    }
    // decode (one q-tile per (segment, kv head, split)) with wave_split = 2 / 4: the workgroup's waves split its keys, LDS merge
    const int ws = (qtiles == 1 && (a.wave_split == 2 || a.wave_split == 4) && a.hd == 128) ? a.wave_split : 1;
    // one wave per q-tile, up to four per workgroup: a decode step (max_q = 1) has ONE q-tile per (segment, kv head, split), so its
    // workgroups are single waves (with 256 threads three of the four waves of each of the 544 workgroups only exited)
    const int wpb = ws > 1 ? ws : (qtiles < 4 ? qtiles : 4);
    dim3 grid(ws > 1 ? qtiles : (qtiles + wpb - 1) / wpb, a.nkv * a.nsplit, a.nseg), block(64 * wpb);
    if (a.nsplit == 1 && qtiles >= 4 && (a.hd == 128 || a.hd == 72) && umv_attn_prefill_enabled(a.variant) && umv_attn_prefill_can_take(a))
        return umv_attn_prefill_launch(a, qtiles, scale_log2e, s);
    const bool pg = a.page_table != nullptr;
    UMV_CHECK(!pg || a.hd == 128, UMV_ERR_UNSUPPORTED, "attn: paged KV is built for head_dim 128 (the LLM's cache); got %d", a.hd);
#define UMV_ATTN_LAUNCH(HD_, WS_, LDS_)                                                                               \
    do {                                                                                                               \
        if (pg) hipLaunchKernelGGL((attn_kernel<HD_, WS_, true>), grid, block, LDS_, s, a, scale_log2e);               \
        else hipLaunchKernelGGL((attn_kernel<HD_, WS_, false>), grid, block, LDS_, s, a, scale_log2e);                 \
    } while (0)
    if (a.hd == 128 && ws == 4) UMV_ATTN_LAUNCH(128, 4, 4 * 8 * 1024 + 4 * 16 * 2 * 4);
    else if (a.hd == 128 && ws == 2) UMV_ATTN_LAUNCH(128, 2, 2 * 8 * 1024 + 2 * 16 * 2 * 4);
    else if (a.hd == 128) UMV_ATTN_LAUNCH(128, 1, 0);
#undef UMV_ATTN_LAUNCH
    else if (a.hd == 72)
        hipLaunchKernelGGL((attn_kernel<72, 1, false>), grid, block, 0, s, a, scale_log2e);
    else if (a.hd == 512 && a.nsplit == 1)   // VAE mid-block attention, single head of 512 (autoencoder.py:50-62)
        hipLaunchKernelGGL((attn_kernel<512, 1, false>), grid, block, 0, s, a, scale_log2e);
    else
        UMV_CHECK(false, UMV_ERR_UNSUPPORTED, "attn: head_dim %d unsupported (128, 72, 512)", a.hd);
    UMV_LAUNCH_CHECK();
    if (a.nsplit > 1)   // grid over the static bound nseg*max_q tokens; rows beyond cu_q[nseg]*nq exit on device
        return umv_attn_combine_launch((const float*)a.workspace, a.out, a.cu_q, a.nseg, a.nq, a.hd, a.nsplit,
                                       (int64_t)a.nseg * a.max_q * a.nq, s);
    return UMV_OK;
}

