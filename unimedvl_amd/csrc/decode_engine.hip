// Decode layer engine for libunimedvl_hip (gfx950): a CHAIN of weight-streaming GEMMs of one decode step
// (Bagel.generate_text, bagel.py:1262-1314 -> Qwen2MoTDecoderLayer.forward_inference, qwen2_navit.py:843-902:
// o_proj + residual -> post-attention RMSNorm -> gate/up + SwiGLU -> down_proj -> residual -> next input RMSNorm -> QKV)
// as ONE persistent launch, so that the weight stream of GEMM n+1 is already in flight while GEMM n's results travel
// between the workgroups.  Weights do not depend on activations: only the x operand of an op waits for its producers.
//
// Geometry: one workgroup of 8 waves per CU (grid = 256).  Every wave owns a private ring of DE_R 1-KiB pieces in LDS,
// filled by LDS-DMA (global_load_lds_dwordx4, non-temporal: a weight byte is used once) DE_R pieces ahead of the piece it
// is consuming - ACROSS tile and op boundaries - and drained with counted s_waitcnt vmcnt(DE_R - 1); a piece is one
// 16(n) x 32(k) tile of the packed weight image (already in MFMA A-fragment order, gemm.hip), i.e. one ds_read_b128 per
// lane and one v_mfma_f32_16x16x32_bf16.  A workgroup owns whole 16-column tiles (or (gate, up) tile pairs) over a K
// range (all of K, or one of `kgroups` equal parts: then the result is an fp32 partial sum); its 8 waves split that K
// range into contiguous slices exactly like gemm_skinny_kernel (kt_per = ceil(nkt / 8)) and keep the matching slice of
// x - optionally RMS-normalised on the way in - in registers.  Per tile the 8 slices meet in LDS and one wave (rotating)
// finishes the tile in wave order 0..7: same summation order, same rounding points, same bits as umv_gemm_bf16 at
// M <= 16 without / with k_splits = kgroups.
//
// Workgroups of one launch exchange data through global memory with the placement-independent protocol of the CDNA4
// guide (Guideline 16, R1): the producer stores write-through (agent-scope relaxed atomic stores = global_store ... sc1),
// drains its stores (s_waitcnt vmcnt(0)), then bumps an arrival counter with one relaxed agent-scope atomic; the consumer
// polls the counter words relaxed from ONE lane (bounded: a timeout writes an error code and lets the launch finish with
// garbage instead of hanging), issues ONE agent-scope acquire (buffer_inv sc1) and only then reads with plain loads.
// Counters are zeroed by the host before every launch (umv_decode_engine does it with a memset node on the stream).
#include "common.h"
#include "../../include/unimedvl_hip.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

#define DE_WAVES 8
#define DE_THREADS (DE_WAVES * 64)
#define DE_R 14                      // ring pieces (KiB) per wave
#define DE_NB 7                      // pieces per batch of the consuming loop (DE_XK % DE_NB == 0)
#define DE_XK 14                     // k-tiles of x a wave keeps in registers (its K slice)
#define DE_RING_BYTES (DE_WAVES * DE_R * 1024)
#define DE_RED_BYTES (2 * DE_WAVES * 2 * 64 * 16)     // [buf][wave][part][lane] f32x4
#define DE_MISC_BYTES 1024
#define DE_LDS_BYTES (DE_RING_BYTES + DE_RED_BYTES + DE_MISC_BYTES)
#define DE_SPIN_LIMIT 400000u

typedef __attribute__((address_space(3))) void* de_lds_ptr_t;
// the op table is read through the CONSTANT address space: it never changes during a launch, and only constant-space
// loads are guaranteed to stay scalar (s_load) and hoistable once the kernel also stores to global memory - as ordinary
// global loads they became vector loads with s_waitcnt vmcnt(0), i.e. a full drain of the LDS-DMA ring, inside the hot loop
typedef const __attribute__((address_space(4))) umv_de_op* de_ops_t;

__device__ __forceinline__ uint32_t de_ld_relaxed(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void de_st64(void* p, uint32_t lo, uint32_t hi, bool publish) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    if (publish) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<unsigned long long*>(p) = v;
}

// what a workgroup / wave owns of one GEMM op
struct DeShare {
    int t0, t1;          // tiles [t0, t1) of the packed image
    int nk;              // pieces (k-tiles) per tile for this wave
    int kfirst;          // first k-tile of this wave's slice
    int kg;              // K group of this workgroup
};

template <class OP>
__device__ __forceinline__ DeShare de_share(const OP& op, int cu, int G, int wave) {
    DeShare s;
    const int KG = op.kgroups > 1 ? op.kgroups : 1;
    const int NG = G / KG;
    const int ng = cu / KG;
    s.kg = cu - ng * KG;
    int idx = ng + op.rot;
    idx = idx >= NG ? idx - NG : idx;
    const int mul = op.pair ? 2 : 1;
    const int units = op.ntiles / mul;
    const int base = units / NG, rem = units - base * NG;
    const int u0 = idx < rem ? idx * (base + 1) : rem * (base + 1) + (idx - rem) * base;
    const int nu = base + (idx < rem ? 1 : 0);
    s.t0 = u0 * mul;
    s.t1 = (u0 + nu) * mul;
    const int nkt = op.KT / KG;
    const int kt_per = (nkt + DE_WAVES - 1) / DE_WAVES;
    const int kb = min(nkt, wave * kt_per), ke = min(nkt, kb + kt_per);
    s.nk = ke - kb;
    s.kfirst = s.kg * nkt + kb;
    return s;
}

// the weight stream of this wave: runs DE_R pieces ahead of the consumer, across tiles and ops
struct DeProd {
    int op, t, t1, kk, nk;
    int64_t tstride;            // bytes between consecutive tiles of the image
    const char* lane_base;      // image + (kfirst * 64 + lane) * 16
};

__device__ __forceinline__ void de_prod_seek(DeProd& P, de_ops_t ops, int nops, int cu, int G, int wave, int lane) {
    while (P.op < nops) {
        const auto& o = ops[P.op];
        if (o.kind == UMV_DE_GEMM) {
            const DeShare s = de_share(o, cu, G, wave);
            if (s.t0 < s.t1 && s.nk > 0) {
                P.t = s.t0; P.t1 = s.t1; P.kk = 0; P.nk = s.nk;
                P.tstride = (int64_t)o.KT * 1024;
                P.lane_base = reinterpret_cast<const char*>(o.w) + ((int64_t)s.kfirst * 64 + lane) * 16;
                return;
            }
        }
        ++P.op;
    }
}

template <int DBG>
__global__ __launch_bounds__(DE_THREADS) void decode_engine_kernel(const umv_de_op* __restrict__ ops_g, int nops, int M, int G,
                                                                   uint32_t* __restrict__ err, const uint16_t* __restrict__ dummy) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const de_ops_t ops = (de_ops_t)(uintptr_t)ops_g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int cu = blockIdx.x;
    char* ring = smem + wave * (DE_R * 1024);
    f32x4* red = reinterpret_cast<f32x4*>(smem + DE_RING_BYTES);
    float* misc = reinterpret_cast<float*>(smem + DE_RING_BYTES + DE_RED_BYTES);   // [DE_WAVES][16] norm partials

    // ---- producer: prime the ring
    DeProd P;
    P.op = 0; P.t = P.t1 = P.kk = P.nk = 0; P.tstride = 0; P.lane_base = nullptr;
    de_prod_seek(P, ops, nops, cu, G, wave, lane);
    const char* dummy_lane = reinterpret_cast<const char*>(dummy) + lane * 16;
    auto issue = [&](int slot) {
        const char* src;
        if (P.op < nops) {
            src = P.lane_base + ((int64_t)P.t * P.tstride + (int64_t)P.kk * 1024);
            if (++P.kk == P.nk) {
                P.kk = 0;
                if (++P.t == P.t1) {
                    ++P.op;
                    de_prod_seek(P, ops, nops, cu, G, wave, lane);
                }
            }
        } else {
            src = dummy_lane;       // keep DE_R pieces in flight to the end: every counted wait sees the same queue depth
        }
        if constexpr (DBG & 4) __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(ring + slot * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(ring + slot * 1024), 16, 0, 2 /* nt */);
    };
#pragma unroll 1
    for (int s = 0; s < DE_R; ++s) issue(s);
    int slot = 0;
    int unit_no = 0;      // tiles finished by this workgroup: red double buffer + rotating finisher

    auto wait_counter = [&](const uint32_t* cnt, int word, uint32_t target) {
        // one lane polls (relaxed, L2), bounded; then ONE agent-scope acquire; the barrier releases the other waves
        if (wave == 0) {
            if (lane == 0) {
                uint32_t spins = 0;
                for (;;) {
                    uint32_t v;
                    if (word < 0) {
                        v = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) v += de_ld_relaxed(cnt + i * 16);
                    } else {
                        v = de_ld_relaxed(cnt + word * 16);
                    }
                    if (v >= target) break;
                    if (++spins > DE_SPIN_LIMIT || de_ld_relaxed(err) != 0u) {
                        __hip_atomic_store(err, 0xDE000000u | (uint32_t)(cu & 0xFFFF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    for (int oi = 0; oi < nops; ++oi) {
        const auto& op = ops[oi];
        const bool publish = op.publish != 0;

        if (op.kind == UMV_DE_REDUCE) {
            // partial sums of a K-grouped GEMM -> residual stream: seq = bf16(bf16(sum_kg partial) + seq), tile by tile.
            // Tile `idx` (16 columns) belongs to workgroup idx; it waits for the kgroups producers of its n-group.
            const int ntile = op.ntiles;
            const int idx = cu;
            if (op.wait_cnt) {
                const int word = idx < ntile ? idx / op.sig_div : 0;      // n-group of the tile
                wait_counter(idx < ntile ? op.wait_cnt : op.wait_cnt, idx < ntile ? word : 0, idx < ntile ? op.wait_target : 0u);
            }
            if (wave == 0 && idx < ntile && r < M) {
                const int n0 = idx * 16 + g * 4;
                const float* pp = reinterpret_cast<const float*>(op.x) + (int64_t)r * op.ldx + n0;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int KG = op.kgroups;
                for (int s = 0; s < KG; ++s) acc += *reinterpret_cast<const f32x4*>(pp + (int64_t)s * op.split_stride);
                uint16_t* rr = op.resid + (int64_t)r * op.ldr + n0;
                const u32x2 pk = *reinterpret_cast<const u32x2*>(rr);
                const float v0 = rbf(rbf(acc.x) + __uint_as_float(pk.x << 16)), v1 = rbf(rbf(acc.y) + __uint_as_float(pk.x & 0xFFFF0000u));
                const float v2 = rbf(rbf(acc.z) + __uint_as_float(pk.y << 16)), v3 = rbf(rbf(acc.w) + __uint_as_float(pk.y & 0xFFFF0000u));
                de_st64(rr, pack2bf(v0, v1), pack2bf(v2, v3), publish);
            }
            if (op.sig_cnt && wave == 0 && idx < ntile) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(op.sig_cnt + (cu & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }

        // ---------------------------------------------------------------- GEMM op
        const DeShare S = de_share(op, cu, G, wave);
        const bool has_work = S.t0 < S.t1;            // workgroup-uniform
        if (!has_work) continue;                      // (no barriers are skipped: every barrier below is inside has_work)
        if (op.wait_cnt) wait_counter(op.wait_cnt, op.wait_mode ? S.kg : -1, op.wait_target);

        // ---- x slice of this wave -> registers (B-operand fragments), optionally RMS-normalised (Qwen2RMSNorm, two roundings)
        bf16x8 x[DE_XK];
        {
            const bool rowok = r < M;
            const uint16_t* xr = op.x + (int64_t)(rowok ? r : 0) * op.ldx + (int64_t)S.kfirst * 32 + g * 8;
#pragma unroll
            for (int kk = 0; kk < DE_XK; ++kk) x[kk] = (kk < S.nk && rowok) ? ldg_frag(xr + kk * 32) : zero_frag();
            if (op.norm_w) {
                bf16x8 nw[DE_XK];
                const uint16_t* wr = op.norm_w + (int64_t)S.kfirst * 32 + g * 8;
#pragma unroll
                for (int kk = 0; kk < DE_XK; ++kk) nw[kk] = kk < S.nk ? ldg_frag(wr + kk * 32) : zero_frag();
                float ss = 0.f;
#pragma unroll
                for (int kk = 0; kk < DE_XK; ++kk)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = bf2f((bf16_t)x[kk][j]);
                        ss += f * f;
                    }
                ss = xor16_sum(ss);
                ss = xor32_sum(ss);
                if (g == 0) misc[wave * 16 + r] = ss;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < DE_WAVES; ++w) tot += misc[w * 16 + r];
                const float rstd = rsqrt_ieee(tot / (float)(op.KT * 32) + op.norm_eps);
#pragma unroll
                for (int kk = 0; kk < DE_XK; ++kk)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        x[kk][j] = (short)f2bf(bf2f((bf16_t)nw[kk][j]) * rbf(bf2f((bf16_t)x[kk][j]) * rstd));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();          // misc may be rewritten by the next normalising op
            }
        }

        const int mul = op.pair ? 2 : 1;
        for (int t = S.t0; t < S.t1; t += mul) {
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int part = 0; part < mul; ++part) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                // batches of DE_NB pieces: ONE counted wait, DE_NB fragment reads, ONE LDS wait, DE_NB refills, DE_NB MFMAs - piece by
                // piece (wait, read, wait, refill, MFMA) every piece paid an LDS round trip before its slot was refilled and the
                // consuming loop held the stream at 5.8 TB/s where the DMA issue alone runs 6.3
#pragma unroll
                for (int kb = 0; kb < DE_XK; kb += DE_NB) {
                    if (kb < S.nk) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DE_R - DE_NB) : "memory");    // the DE_NB oldest pieces have landed
                        bf16x8 wf[DE_NB];
                        if constexpr (!(DBG & 1)) {
#pragma unroll
                            for (int j = 0; j < DE_NB; ++j) {
                                const int sl = slot + j >= DE_R ? slot + j - DE_R : slot + j;
                                wf[j] = (kb + j < S.nk) ? *reinterpret_cast<const bf16x8*>(ring + sl * 1024 + lane * 16) : zero_frag();
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // ... and have been read: refill their slots
                        }
#pragma unroll
                        for (int j = 0; j < DE_NB; ++j)
                            if (kb + j < S.nk) {
                                issue(slot);
                                slot = slot + 1 == DE_R ? 0 : slot + 1;
                            }
#pragma unroll
                        for (int j = 0; j < DE_NB; ++j) {
                            if constexpr (!(DBG & 1)) { if (kb + j < S.nk) a = mfma16(wf[j], x[kb + j], a); }
                            else asm volatile("" : "+v"(a));
                        }
                    }
                }
                if (part == 0) acc[0] = a; else acc[1] = a;
            }
            if constexpr (DBG & 2) { asm volatile("" :: "v"(acc[0]), "v"(acc[1])); ++unit_no; continue; }
            // ---- the 8 K slices meet in LDS; one wave (rotating) finishes the tile in wave order
            const int buf = unit_no & 1;
            f32x4* rb = red + (buf * DE_WAVES) * 2 * 64;
            rb[(wave * 2 + 0) * 64 + lane] = acc[0];
            if (mul == 2) rb[(wave * 2 + 1) * 64 + lane] = acc[1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wave == (unit_no & (DE_WAVES - 1))) {
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < DE_WAVES; ++w) {
                    s0 += rb[(w * 2 + 0) * 64 + lane];
                    if (mul == 2) s1 += rb[(w * 2 + 1) * 64 + lane];
                }
                const int m = r;
                if (m < M) {
                    if (op.pair) {      // SwiGLU: act[m][c0..c0+3] = bf16(bf16(silu(bf16 g)) * bf16 u)   (modeling_qwen2.py:235)
                        const int c0 = (t >> 1) * 16 + g * 4;
                        const float gg[4] = {s0.x, s0.y, s0.z, s0.w}, uu[4] = {s1.x, s1.y, s1.z, s1.w};
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = rbf(rbf(silu_f(rbf(gg[j]))) * rbf(uu[j]));
                        de_st64(reinterpret_cast<uint16_t*>(op.out) + (int64_t)m * op.ldo + c0, pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), publish);
                    } else if (op.epi == UMV_DE_EPI_PARTIAL) {
                        float* o = reinterpret_cast<float*>(op.out) + (int64_t)S.kg * op.split_stride + (int64_t)m * op.ldo + t * 16 + g * 4;
                        de_st64(o, __float_as_uint(s0.x), __float_as_uint(s0.y), publish);
                        de_st64(o + 2, __float_as_uint(s0.z), __float_as_uint(s0.w), publish);
                    } else {
                        const int n0 = t * 16 + g * 4;
                        float v[4] = {s0.x, s0.y, s0.z, s0.w};
                        if (op.bias) {
                            const u32x2 pk = *reinterpret_cast<const u32x2*>(op.bias + n0);
                            v[0] += __uint_as_float(pk.x << 16); v[1] += __uint_as_float(pk.x & 0xFFFF0000u);
                            v[2] += __uint_as_float(pk.y << 16); v[3] += __uint_as_float(pk.y & 0xFFFF0000u);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = rbf(v[j]);
                        uint16_t* o = reinterpret_cast<uint16_t*>(op.out) + (int64_t)m * op.ldo + n0;
                        if (op.epi == UMV_DE_EPI_RESIDUAL) {
                            const u32x2 pk = *reinterpret_cast<const u32x2*>(op.resid + (int64_t)m * op.ldr + n0);
                            v[0] = rbf(v[0] + __uint_as_float(pk.x << 16)); v[1] = rbf(v[1] + __uint_as_float(pk.x & 0xFFFF0000u));
                            v[2] = rbf(v[2] + __uint_as_float(pk.y << 16)); v[3] = rbf(v[3] + __uint_as_float(pk.y & 0xFFFF0000u));
                        }
                        de_st64(o, pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), publish);
                    }
                }
                if (op.sig_cnt) {
                    const bool last = t + mul >= S.t1;
                    if (op.sig_mode != UMV_DE_SIG_GROUP_END || last) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every store of this wave is out (write-through)
                        if (lane == 0) {
                            int word;
                            if (op.sig_mode == UMV_DE_SIG_XCD) word = cu & 7;
                            else if (op.sig_mode == UMV_DE_SIG_UNIT_DIV) word = (t / mul) / op.sig_div;
                            else word = (cu / (op.kgroups > 1 ? op.kgroups : 1));   // GROUP_END: one arrival per workgroup, word = n-group
                            __hip_atomic_fetch_add(op.sig_cnt + word * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            ++unit_no;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup's LDS allocation
}

extern "C" size_t umv_decode_engine_counter_words(void) { return 16 * 64; }

extern "C" int umv_decode_engine(const umv_de_op* ops_dev, int nops, int M, uint32_t* counters, size_t counter_words, uint32_t* err,
                                 const uint16_t* dummy_kib, int grid, umv_stream_t stream) {
    UMV_CHECK(ops_dev && nops > 0 && err && dummy_kib, UMV_ERR_ARG, "decode_engine: null pointer");
    UMV_CHECK(M >= 1 && M <= 16, UMV_ERR_UNSUPPORTED, "decode_engine: M=%d (1..16 rows)", M);
    UMV_CHECK(grid >= 8 && grid <= 1024 && grid % 8 == 0, UMV_ERR_ARG, "decode_engine: grid=%d", grid);
    hipStream_t s = (hipStream_t)stream;
    const char* dbg_env = getenv("UMV_DE_DBG");     // tuning only: 1 = no LDS read / MFMA, 2 = no tile-end reduce, 4 = default cache policy
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    auto kern = decode_engine_kernel<0>;
    switch (dbg) {
        case 1: kern = decode_engine_kernel<1>; break;
        case 2: kern = decode_engine_kernel<2>; break;
        case 3: kern = decode_engine_kernel<3>; break;
        case 4: kern = decode_engine_kernel<4>; break;
        default: break;
    }
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DE_LDS_BYTES);
        UMV_CHECK(e == hipSuccess, UMV_ERR_LAUNCH, "decode_engine: cannot reserve %d bytes of LDS: %s", DE_LDS_BYTES, hipGetErrorString(e));
    }
    if (counters && counter_words) {
        hipError_t e = hipMemsetAsync(counters, 0, counter_words * sizeof(uint32_t), s);
        UMV_CHECK(e == hipSuccess, UMV_ERR_LAUNCH, "decode_engine: counter memset failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DE_THREADS), DE_LDS_BYTES, s, ops_dev, nops, M, grid, err, dummy_kib);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
