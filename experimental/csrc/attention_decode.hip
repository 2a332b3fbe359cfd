// Decode attention with the q/k RMSNorm + RoPE + KV append folded in (gfx950): one token per segment.
// Replaces, for a decode step, umv_qkv_post + the split-KV pass of umv_attn_varlen
// (qwen2_navit.py:544-600 followed by flash_attn_varlen_func at :605-614); umv's combine kernel still merges the splits.
//
// One wave per (segment, kv head, key split).  Its 16 MFMA rows are the G = nq/nkv query heads of the group (rows >= G
// idle), exactly as in attn_kernel.  What is new:
//   * the wave reads the RAW fused-QKV row of its segment: lane (j, g) holds the four 8-element chunks ks*32 + g*8 of
//     head j - the MFMA B-fragment layout - so Qwen2RMSNorm (two shuffles across g for the row sum) and RoPE (chunk ks
//     pairs with chunk ks+2 of the SAME lane: rotate_half is lane-local) run in registers, `und` bf16 chain.  The row may
//     also arrive as the fp32 partial sums of a split-K QKV GEMM (PART): summed in split order, + bias, rounded to bf16 -
//     the values the GEMM epilogue / umv_qkv_post would have produced;
//   * the new key / value never round-trip through the slab inside the kernel: the wave that owns the last key writes
//     K / V^T for later steps and PATCHES the fragments of its last 32-key block in registers (the lanes whose MFMA row /
//     key octet is the new position), so the scores and the P.V product come out of the same MFMAs as before;
//   * the grid is one wave per SIMD at most (a few hundred waves), so the kernel lasts as long as ONE wave's chain of
//     dependent memory round trips.  All of a wave's first two 32-key blocks (K and V^T fragments, 128 VGPRs) are
//     requested together with the prologue operands, before anything is waited on: a split of <= 64 keys (every context
//     up to 2048 keys at nsplit = 32) pays one memory latency instead of three.
#include <type_traits>

#include "common.h"
#include "attention_combine.h"
#include "unimedvl_hip_experimental.h"

__device__ __forceinline__ bf16x8 adec_mask_keys(bf16x8 v, int nvalid) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = j < nvalid ? v[j] : (short)0;
    return o;
}

// Qwen2RMSNorm + RoPE of one 128-wide head held as 4 chunks of 8 (chunk ks = elements ks*32 + g*8 .. +8), bf16 chain of
// qkv_post_kernel: n = bf16(w * bf16(x * rstd)); out = bf16(bf16(n * cos) + bf16(+-n_pair * sin))
__device__ __forceinline__ void adec_norm_rope(const bf16x8 (&x)[4], const bf16x8 (&w)[4], const bf16x8 (&c)[4], const bf16x8 (&s)[4],
                                               float eps, bf16x8 (&out)[4]) {
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf2f((bf16_t)x[ks][e]);
            ss += f * f;
        }
    ss = xor16_sum(ss);   // the 4 lanes (g = 0..3) that share row j
    ss = xor32_sum(ss);
    const float rstd = rsqrt_ieee(ss / 128.0f + eps);
    float n[4][8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) n[ks][e] = rbf(bf2f((bf16_t)w[ks][e]) * rbf(bf2f((bf16_t)x[ks][e]) * rstd));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float n1 = n[ks][e], n2 = n[ks + 2][e];
            out[ks][e] = (short)f2bf(rbf(n1 * bf2f((bf16_t)c[ks][e])) + rbf(-n2 * bf2f((bf16_t)s[ks][e])));
            out[ks + 2][e] = (short)f2bf(rbf(n2 * bf2f((bf16_t)c[ks + 2][e])) + rbf(n1 * bf2f((bf16_t)s[ks + 2][e])));
        }
}

// 8 consecutive columns of the QKV row from split-K partial sums: bf16(((0 + P[0]) + P[1] + ...) + bias), the order of
// qkv_post_kernel
// NS > 0: that many partial sums, known at compile time so that all their loads are requested before the first add
// (a run-time loop costs one memory round trip per split: 24 us instead of 11 for the kernel); NS < 0: n of them
template <int NS, bool BIAS>
__device__ __forceinline__ bf16x8 adec_sum_frag(const float* p, int n, int64_t stride, const bf16_t* bias) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (NS > 0) {
        f32x4 lo[NS], hi[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            lo[s] = *reinterpret_cast<const f32x4*>(p + s * stride);
            hi[s] = *reinterpret_cast<const f32x4*>(p + s * stride + 4);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            acc[0] += lo[s].x; acc[1] += lo[s].y; acc[2] += lo[s].z; acc[3] += lo[s].w;
            acc[4] += hi[s].x; acc[5] += hi[s].y; acc[6] += hi[s].z; acc[7] += hi[s].w;
        }
    } else {
        for (int s = 0; s < n; ++s) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(p + s * stride);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(p + s * stride + 4);
            acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w;
            acc[4] += hi.x; acc[5] += hi.y; acc[6] += hi.z; acc[7] += hi.w;
        }
    }
    bf16x8 o;
    if constexpr (BIAS) {
        const bf16x8 b = ldg_frag(bias);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f((bf16_t)b[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (short)f2bf(acc[e]);
    return o;
}

// one column of the QKV row.  NS == 0: the bf16 row; > 0: that many split-K partial sums; < 0: a.n_splits of them
template <int NS, bool BIAS>
__device__ __forceinline__ float adec_raw(const umv_attn_decode_args& a, int s, int64_t col) {
    if constexpr (NS == 0) {
        return bf2f(a.qkv[(int64_t)s * a.ld_qkv + col]);
    } else {
        const float* p = a.qkv_partials + (int64_t)s * a.ld_qkv + col;
        float v = 0.f;
        if constexpr (NS > 0) {
            float t[NS];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) t[sp] = p[sp * a.split_stride];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) v += t[sp];
        } else {
            for (int sp = 0; sp < a.n_splits; ++sp) v += p[sp * a.split_stride];
        }
        if constexpr (BIAS) v += bf2f(a.qkv_bias[col]);
        return rbf(v);
    }
}

template <int NS>
__global__ __launch_bounds__(64) void attn_decode_fused_kernel(umv_attn_decode_args a, float scale_log2e) {
    constexpr int HD = 128, KS = 4, DT = 8;
    const int lane = threadIdx.x;
    const int j = lane & 15, g = lane >> 4;
    const int G = a.nq / a.nkv;
    const int s = blockIdx.z;
    const int kh = blockIdx.y % a.nkv;
    const int split = blockIdx.y / a.nkv;
    const int Lk = a.kv_len[s];          // includes the token of this step
    const int kpos = Lk - 1;             // its cache slot
    const bool rvalid = j < G;
    const int head = kh * G + (rvalid ? j : 0);
    const int pos = a.tok_pos[s];

    // ---- prologue: raw q (per row j) and raw k (same for every row), norm weights, cos / sin at `pos`
    bf16x8 qraw[KS], kraw[KS], qw[KS], kw[KS], cs[KS], sn[KS];
    float vnew[DT];
    if constexpr (NS != 0) {
        // one uniform branch on "has bias" around the whole prologue: a branch per fragment would serialise the loads
        auto load_rows = [&](auto bias_tag) {
            constexpr bool BIAS = decltype(bias_tag)::value;
            const float* prow = a.qkv_partials + (int64_t)s * a.ld_qkv;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int d = ks * 32 + g * 8;
                const int64_t cq = (int64_t)head * HD + d, ck = (int64_t)(a.nq + kh) * HD + d;
                const bf16x8 qv = adec_sum_frag<NS, BIAS>(prow + cq, a.n_splits, a.split_stride, a.qkv_bias + cq);
                qraw[ks] = rvalid ? qv : zero_frag();    // idle rows read head kh*G again: no branch around the loads
                kraw[ks] = adec_sum_frag<NS, BIAS>(prow + ck, a.n_splits, a.split_stride, a.qkv_bias + ck);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vnew[dt] = adec_raw<NS, BIAS>(a, s, (int64_t)(a.nq + a.nkv + kh) * HD + dt * 16 + j);
        };
        if (a.qkv_bias) load_rows(std::true_type{});
        else load_rows(std::false_type{});
    } else {
        const bf16_t* row = a.qkv + (int64_t)s * a.ld_qkv;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            qraw[ks] = rvalid ? ldg_frag(row + (int64_t)head * HD + d) : zero_frag();
            kraw[ks] = ldg_frag(row + (int64_t)(a.nq + kh) * HD + d);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vnew[dt] = bf2f(row[(int64_t)(a.nq + a.nkv + kh) * HD + dt * 16 + j]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d = ks * 32 + g * 8;
        qw[ks] = ldg_frag(a.q_norm_w + d);
        kw[ks] = ldg_frag(a.k_norm_w + d);
        cs[ks] = ldg_frag(a.cos_tab + (int64_t)pos * HD + d);
        sn[ks] = ldg_frag(a.sin_tab + (int64_t)pos * HD + d);
    }

    // key range of this split, in 32-key blocks (same partition as attn_kernel)
    int kb_begin = 0, kb_end = Lk;
    if (a.nsplit > 1) {
        const int chunk = ((Lk + a.nsplit - 1) / a.nsplit + 31) & ~31;
        kb_begin = split * chunk;
        kb_end = min(Lk, kb_begin + chunk);
    }
    const bf16_t* kbase = a.k_slab + s * a.k_seg_stride + kh * a.k_head_stride;
    const bf16_t* vbase = a.vt_slab + s * a.v_seg_stride + kh * a.v_head_stride;
    // two stages of K / V^T fragments: blocks kb_begin and kb_begin + 32 are requested now, block i + 2 when block i is done
    bf16x8 kst[2][2][KS], vst[2][DT];
    auto load_kv = [&](int kb, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[DT]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = kb + (j >> 2) * 8 + t * 4 + (j & 3);
            const bf16_t* kp = kbase + (int64_t)key * HD;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) kf[t][ks] = ldg_frag(kp + ks * 32 + g * 8);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vf[dt] = ldg_frag(vbase + (int64_t)(dt * 16 + j) * a.v_d_stride + kb + g * 8);
    };
    if (kb_begin < kb_end) load_kv(kb_begin, kst[0], vst[0]);
    if (kb_begin + 32 < kb_end) load_kv(kb_begin + 32, kst[1], vst[1]);

    bf16x8 qf[KS], knew[KS];
    adec_norm_rope(qraw, qw, cs, sn, a.eps, qf);
    adec_norm_rope(kraw, kw, cs, sn, a.eps, knew);
    const bool owner = kpos >= kb_begin && kpos < kb_end;   // exactly one split per (segment, kv head)
    if (owner) {
        bf16_t* kdst = a.k_slab + s * a.k_seg_stride + kh * a.k_head_stride + (int64_t)kpos * HD;
        if (j == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(kdst + ks * 32 + g * 8) = knew[ks];
        }
        if (g == 0) {
            bf16_t* vdst = a.vt_slab + s * a.v_seg_stride + kh * a.v_head_stride + kpos;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vdst[(int64_t)(dt * 16 + j) * a.v_d_stride] = f2bf(vnew[dt]);
        }
    }

    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    auto block = [&](int kb, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[DT]) {
        if (kpos >= kb && kpos < kb + 32) {   // the block of the new token: take its K row / V column from registers
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (kb + (j >> 2) * 8 + t * 4 + (j & 3) == kpos) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) kf[t][ks] = knew[ks];
                }
            const int e = kpos - (kb + g * 8);
            if (e >= 0 && e < 8) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i == e) vf[dt][i] = (short)f2bf(vnew[dt]);
            }
        }
        f32x4 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[t] = mfma16(kf[t][ks], qf[ks], st[t]);
        }
        float sc[8];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb + g * 8 + t * 4 + r;
                float v = st[t][r] * scale_log2e;
                v = key < kb_end ? v : -INFINITY;      // one query at the end of the sequence: causal == length mask
                sc[t * 4 + r] = v;
                mx = fmaxf(mx, v);
            }
        mx = xor16_max(mx);
        mx = xor32_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : umv_exp2(m_run - m_use);
        float ps = 0.f;
        bf16x8 pf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = umv_exp2(sc[i] - m_use);
            ps += p;
            pf[i] = (short)f2bf(p);
        }
        ps = xor16_sum(ps);
        ps = xor32_sum(ps);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        const bool partial = kb + 32 > Lk;
        const int nvalid = min(8, max(0, Lk - (kb + g * 8)));
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            bf16x8 v = vf[dt];
            if (partial) v = adec_mask_keys(v, nvalid);
            f32x4 acc = o[dt];
            acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
            o[dt] = mfma16(v, pf, acc);
        }
    };
    for (int kb = kb_begin; kb < kb_end; kb += 64) {
        block(kb, kst[0], vst[0]);
        if (kb + 64 < kb_end) load_kv(kb + 64, kst[0], vst[0]);
        if (kb + 32 < kb_end) {
            block(kb + 32, kst[1], vst[1]);
            if (kb + 96 < kb_end) load_kv(kb + 96, kst[1], vst[1]);
        }
    }
    if (!rvalid) return;
    const int64_t rowid = (int64_t)s * a.nq + head;
    if (a.nsplit == 1) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* op = a.out + rowid * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2 pk;
            pk.x = pack2bf(o[dt].x * inv, o[dt].y * inv);
            pk.y = pack2bf(o[dt].z * inv, o[dt].w * inv);
            *reinterpret_cast<u32x2*>(op + dt * 16 + g * 4) = pk;
        }
    } else {   // same workspace layout as attn_kernel: [row][split][HD + 4] fp32 (O, m, l)
        float* po = reinterpret_cast<float*>(a.workspace) + (rowid * a.nsplit + split) * (HD + 4);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16 + g * 4) = o[dt];
        if (g == 0) { po[HD] = m_run; po[HD + 1] = l_run; }
    }
}

template <int NS>
static void adec_launch(const umv_attn_decode_args& a, float scale_log2e, hipStream_t s) {
    hipLaunchKernelGGL((attn_decode_fused_kernel<NS>), dim3(1, a.nkv * a.nsplit, a.nseg), dim3(64), 0, s, a, scale_log2e);
}


extern "C" int umv_attn_decode_fused(const umv_attn_decode_args* ap, umv_stream_t stream) {
    UMV_CHECK(ap, UMV_ERR_ARG, "attn_decode_fused: null args");
    const umv_attn_decode_args& a = *ap;
    UMV_CHECK((a.qkv || a.qkv_partials) && a.out && a.kv_len && a.tok_pos && a.k_slab && a.vt_slab && a.q_norm_w && a.k_norm_w &&
                  a.cos_tab && a.sin_tab && a.cu_q, UMV_ERR_ARG, "attn_decode_fused: null pointer");
    UMV_CHECK(a.hd == 128, UMV_ERR_UNSUPPORTED, "attn_decode_fused: head_dim %d unsupported (128)", a.hd);
    UMV_CHECK(a.nkv > 0 && a.nq % a.nkv == 0 && a.nq / a.nkv <= 16, UMV_ERR_ARG, "attn_decode_fused: bad head counts nq=%d nkv=%d", a.nq, a.nkv);
    UMV_CHECK(a.nsplit >= 1 && a.nsplit <= 32 && (a.nsplit == 1 || a.workspace), UMV_ERR_ARG, "attn_decode_fused: nsplit=%d (1..32) needs workspace", a.nsplit);
    UMV_CHECK((a.v_d_stride % 8) == 0 && (a.ld_qkv % 8) == 0, UMV_ERR_ARG, "attn_decode_fused: strides must be multiples of 8");
    UMV_CHECK(!a.qkv_partials || (a.n_splits >= 1 && a.n_splits <= 64 && (a.split_stride % 4) == 0), UMV_ERR_ARG,
              "attn_decode_fused: partial input needs 1 <= n_splits <= 64 and split_stride %% 4 == 0");
    if (a.nseg == 0) return UMV_OK;
    hipStream_t s = (hipStream_t)stream;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)a.hd);
    switch (a.qkv_partials ? (a.n_splits <= 4 ? a.n_splits : -1) : 0) {
        case 0: adec_launch<0>(a, scale_log2e, s); break;
        case 1: adec_launch<1>(a, scale_log2e, s); break;
        case 2: adec_launch<2>(a, scale_log2e, s); break;
        case 3: adec_launch<3>(a, scale_log2e, s); break;
        case 4: adec_launch<4>(a, scale_log2e, s); break;
        default: adec_launch<-1>(a, scale_log2e, s); break;
    }
    UMV_LAUNCH_CHECK();
    if (a.nsplit > 1) return umv_attn_combine_launch((const float*)a.workspace, a.out, a.cu_q, a.nseg, a.nq, a.hd, a.nsplit,
                                                     (int64_t)a.nseg * a.nq, s);
    return UMV_OK;
}
