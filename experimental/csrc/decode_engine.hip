// Decode layer engine for libunimedvl_hip (gfx950): a CHAIN of weight-streaming GEMMs of one decode step
// (Bagel.generate_text, bagel.py:1262-1314 -> Qwen2MoTDecoderLayer.forward_inference, qwen2_navit.py:843-902:
// o_proj + residual -> post-attention RMSNorm -> gate/up + SwiGLU -> down_proj -> residual -> next input RMSNorm -> QKV)
// as ONE persistent launch, so that the weight stream of GEMM n+1 is already in flight while GEMM n's results travel
// between the workgroups.  Weights do not depend on activations: only the x operand of an op waits for its producers.
//
// Geometry: one workgroup per CU (grid = 256) of 8 STREAMING waves + 1 SERVICE wave.
//   * A streaming wave owns a private ring of DE_R 1-KiB pieces in LDS, filled by LDS-DMA (global_load_lds_dwordx4,
//     non-temporal: a weight byte is used once) DE_R pieces ahead of the piece it is consuming - ACROSS tile and op
//     boundaries - and drained with counted s_waitcnt vmcnt; a piece is one 16(n) x 32(k) tile of the packed weight image
//     (already in MFMA A-fragment order, gemm.hip), i.e. one ds_read_b128 per lane and one v_mfma_f32_16x16x32_bf16.
//     LDS-DMA is the ONLY vector-memory traffic of a streaming wave: loads return in order, so a single ordinary load
//     behind a full ring costs a ring drain plus a round trip of idle HBM (measured: 46.1 vs 43.4 us per gate/up GEMM).
//   * The service wave does everything else: it polls the arrival counters, copies the x rows into LDS (LDS-DMA again,
//     but on its own, empty queue), derives the RMSNorm scale from the producers' partial sums of squares, and finishes
//     every tile - the 8 K slices meet in LDS, are summed in wave order, get the reference's epilogue (bias / residual /
//     SwiGLU with a bf16 rounding wherever the reference materialises a bf16 tensor) and are stored and published.
// A workgroup owns whole 16-column tiles (or (gate, up) tile pairs) over a K range (all of K, or one of `kgroups` equal
// parts: then the result is an fp32 partial sum); its 8 streaming waves split that K range into contiguous slices exactly
// like gemm_skinny_kernel (kt_per = ceil(nkt / 8)) and keep the matching slice of x in registers: same summation order, same
// rounding points, same bits as umv_gemm_bf16 at M <= 8 without / with k_splits = kgroups.
//
// Workgroups of one launch exchange data through global memory with the placement-independent protocol of the CDNA4
// guide (Guideline 16, R1): the producer stores write-through (agent-scope relaxed atomic stores = global_store ... sc1),
// drains its stores (s_waitcnt vmcnt(0)), then bumps an arrival counter with one relaxed agent-scope atomic; the consumer
// polls the counter words relaxed from ONE lane (bounded: a timeout writes an error code and lets the launch finish with
// garbage instead of hanging), issues ONE agent-scope acquire (buffer_inv sc1) and only then reads.
// Counters are zeroed by the host before every launch (umv_decode_engine does it with a memset node on the stream).
#include "common.h"
#include "unimedvl_hip_experimental.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

#define DE_SW 8                      // streaming waves
#define DE_NSV 4                     // service waves
#define DE_THREADS ((DE_SW + DE_NSV) * 64)
#define DE_XK 14                     // k-tiles per wave slice and tile = pieces per register frame
#define DE_MROWS 8                   // rows (tokens) per step
#define DE_RED_BYTES (2 * DE_SW * 2 * 32 * 16)              // [buf][wave][part][g*8 + r] f32x4
#define DE_XS_BYTES (DE_SW * DE_XK * 4 * DE_MROWS * 16)     // [wave][kk][g][r] 16-byte fragment pieces of x
#define DE_NW_BYTES (DE_SW * DE_XK * 4 * 16)                // [k-tile][g] norm weights
#define DE_MISC_BYTES 256
#define DE_LDS_BYTES (DE_RED_BYTES + DE_XS_BYTES + DE_NW_BYTES + DE_MISC_BYTES)
#define DE_SPIN_LIMIT 400000u
#define DE_MAXKG 8                   // K groups of a partial-sum op
#define DE_SS_PER_LANE 28             // ss_in tiles per lane of the service wave (8 * 28 = 224 tiles = 3584 columns)

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
__device__ __forceinline__ void de_st32f(float* p, float v, bool publish) {
    if (publish) __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// what a workgroup owns of one GEMM op, and how its K range is cut into the 8 wave slices
struct DeShare {
    int t0, t1;          // tiles [t0, t1) of the packed image
    int kg;              // K group of this workgroup
    int kbase;           // first k-tile of the workgroup's K range
    int nkt;             // k-tiles in the workgroup's K range
    int kt_per;          // k-tiles per wave slice (the last slices may be shorter or empty)
};

template <class OP>
__device__ __forceinline__ DeShare de_share(const OP& op, int cu, int G) {
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
    s.nkt = op.KT / KG;
    s.kbase = s.kg * s.nkt;
    s.kt_per = (s.nkt + DE_SW - 1) / DE_SW;
    return s;
}
__device__ __forceinline__ int de_slice_nk(const DeShare& s, int wave) {
    const int kb = min(s.nkt, wave * s.kt_per), ke = min(s.nkt, kb + s.kt_per);
    return ke - kb;
}

// the tile runs of a workgroup: GEMM ops with work, tile by tile.  Two cursors walk them: the consumer, and the loader two runs ahead.
struct DeIter {
    int op, t, t1, nk;
    int64_t tstride;            // bytes between consecutive tiles of the image
    const char* lane_base;      // image + (first k-tile of this wave's slice * 64 + lane) * 16
};

__device__ __forceinline__ void de_iter_seek(DeIter& it, de_ops_t ops, int nops, int cu, int G, int wave, int lane) {
    while (it.op < nops) {
        const auto& o = ops[it.op];
        if (o.kind == UMV_DE_GEMM) {
            const DeShare s = de_share(o, cu, G);
            if (s.t0 < s.t1) {
                it.t = s.t0; it.t1 = s.t1; it.nk = de_slice_nk(s, wave);
                it.tstride = (int64_t)o.KT * 1024;
                it.lane_base = reinterpret_cast<const char*>(o.w) + ((int64_t)(s.kbase + wave * s.kt_per) * 64 + lane) * 16;
                return;
            }
        }
        ++it.op;
    }
}
__device__ __forceinline__ void de_iter_next(DeIter& it, de_ops_t ops, int nops, int cu, int G, int wave, int lane) {
    if (++it.t == it.t1) {
        ++it.op;
        de_iter_seek(it, ops, nops, cu, G, wave, lane);
    }
}

// DBG (tuning only): 1 = no MFMA, 8 = no x copy, 16 = no row squaring in the fallback path
template <int DBG>
__global__ __launch_bounds__(DE_THREADS) void decode_engine_kernel(const umv_de_op* __restrict__ ops_g, int nops, int M, int G,
                                                                   uint32_t* __restrict__ err, const uint16_t* __restrict__ zeros,
                                                                   unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const de_ops_t ops = (de_ops_t)(uintptr_t)ops_g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x;
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    char* xs = smem + DE_RED_BYTES;
    char* nws = xs + DE_XS_BYTES;
    float* misc = reinterpret_cast<float*>(nws + DE_NW_BYTES);   // [0..7] rstd per row
    {   // the op table is read field by field with scalar loads: touch every 64-byte line of it once, all loads in flight together,
        // so that none of the later reads is a cold scalar-cache miss under a saturated memory system (5.5 us at op 0 otherwise)
        const __attribute__((address_space(4))) int* tbl = (const __attribute__((address_space(4))) int*)(uintptr_t)ops_g;
        const int nline = (nops * (int)sizeof(umv_de_op) + 63) / 64;
        int acc = 0;
#pragma unroll
        for (int i = 0; i < 24; ++i) acc += tbl[(i < nline ? i : 0) * 16];
        asm volatile("" ::"s"(acc));
    }

    if (wave < DE_SW) {
        // =========================================================================== streaming wave
        // The weight stream lives in REGISTERS: two frames of 14 pieces (16 bytes per lane each = one 16 x 32 tile of the image as
        // an MFMA A fragment), frame A = the run being consumed, frame B = the next one; a piece's registers are reloaded - for
        // the run after next - right behind the MFMA that consumed them, so 28 KiB per wave (224 KiB per CU, ~9 us of stream)
        // stay in flight across tile, op and dependency boundaries.  Plain non-temporal loads: the compiler counts vmcnt; these
        // are the only vector-memory operations of a streaming wave.  Unused positions (a slice shorter than 14 k-tiles, the end of
        // the program) load the zero page, so the queue depth never changes.
        const int r = lane & 15, g = lane >> 4;
        const char* zero_lane = reinterpret_cast<const char*>(zeros) + lane * 16;
        DeIter C, P;
        C.op = 0; C.t = C.t1 = C.nk = 0; C.tstride = 0; C.lane_base = nullptr;
        de_iter_seek(C, ops, nops, cu, G, wave, lane);
        P = C;
        // The first op of the launch, when its x was complete before the launch and needs no norm (o_proj): every streaming wave
        // fetches its OWN slice of x - requested ahead of the ring fills, so it arrives first - instead of waiting for the
        // service waves (4 us at launch start, with nothing to hide behind).
        bool self_x = false;
        bf16x8 xv0[DE_XK];
        if (C.op < nops) {
            const auto& op0 = ops[C.op];
            self_x = op0.wait_cnt == nullptr && op0.norm_w == nullptr;
            if (self_x) {
                const DeShare S0 = de_share(op0, cu, G);
                const int nk0 = de_slice_nk(S0, wave);
                const bool ok0 = r < DE_MROWS && r < M;
                const uint16_t* xr = op0.x + (int64_t)(ok0 ? r : 0) * op0.ldx + (int64_t)(S0.kbase + wave * S0.kt_per) * 32 + g * 8;
#pragma unroll
                for (int kk = 0; kk < DE_XK; ++kk) xv0[kk] = (kk < nk0 && ok0) ? ldg_frag(xr + kk * 32) : zero_frag();
            }
        }
        bf16x8 FA[DE_XK], FB[DE_XK];
        auto fill = [&](bf16x8(&F)[DE_XK]) {       // the run under the loader cursor -> F, cursor to the next run
            const bool pv = P.op < nops;
            const char* base = pv ? P.lane_base + (int64_t)P.t * P.tstride : zero_lane;
            const int pnk = pv ? P.nk : 0;
#pragma unroll
            for (int kk = 0; kk < DE_XK; ++kk)
                F[kk] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kk < pnk ? base + kk * 1024 : zero_lane));
            if (pv) de_iter_next(P, ops, nops, cu, G, wave, lane);
        };
        fill(FA);
        const bool rowok = r < DE_MROWS;
        const char* xlane = xs + ((wave * DE_XK * 4 + g) * DE_MROWS + (rowok ? r : 0)) * 16;
        if (self_x && rowok) {          // (between the fills: x + one frame in registers, not x + two)
#pragma unroll
            for (int kk = 0; kk < DE_XK; ++kk) *reinterpret_cast<bf16x8*>(const_cast<char*>(xlane) + kk * (4 * DE_MROWS * 16)) = xv0[kk];
        }
        fill(FB);
        int unit_no = 0;      // tiles finished by this workgroup: red double buffer
        int cur_op = -1;
        int pair = 0, t0 = 0;

        auto run = [&](bf16x8(&F)[DE_XK]) {
            if (C.op != cur_op) {                 // op entry: wait until the service waves have staged (and normalised) x
                cur_op = C.op;
                const auto& op = ops[cur_op];
                pair = op.pair;
                t0 = C.t;
                if (op.wait_cnt || (op.norm_w && op.ss_in)) __builtin_amdgcn_s_barrier();   // B0: the lead service wave saw the producers
                if (op.norm_w && !op.ss_in) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // (fallback: raw x staged; scale known)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                                           // B1: x is ready in LDS
            }
            const bool pv = P.op < nops;
            const char* base = pv ? P.lane_base + (int64_t)P.t * P.tstride : zero_lane;
            const int pnk = pv ? P.nk : 0;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < DE_XK; ++kk) {
                const bf16x8 xf = rowok ? *reinterpret_cast<const bf16x8*>(xlane + kk * (4 * DE_MROWS * 16)) : zero_frag();
                if constexpr (!(DBG & 1)) { if (kk < C.nk) a = mfma16(F[kk], xf, a); }
                else asm volatile("" : "+v"(a) : "v"(F[kk]), "v"(xf));
                F[kk] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(kk < pnk ? base + kk * 1024 : zero_lane));
            }
            if (pv) de_iter_next(P, ops, nops, cu, G, wave, lane);
            // ---- the 8 K slices meet in LDS (rows < 8 only); the lead service wave finishes the tile in wave order
            const int part = pair ? ((C.t - t0) & 1) : 0;
            if (rowok) red[(((unit_no & 1) * DE_SW + wave) * 2 + part) * 32 + g * 8 + r] = a;
            if (!pair || part) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                ++unit_no;
            }
            de_iter_next(C, ops, nops, cu, G, wave, lane);
        };
        while (C.op < nops) {
            run(FA);
            if (C.op >= nops) break;
            run(FB);
        }
#pragma unroll
        for (int kk = 0; kk < DE_XK; ++kk) asm volatile("" ::"v"(FA[kk]), "v"(FB[kk]));   // (the trailing zero-page loads stay counted)
        return;
    }

    // =============================================================================== service waves
    // sv 0 (the lead) polls, derives the norm scale and finishes the tiles; all DE_NSV of them copy x (and the norm
    // weights) into LDS - one wave issues an LDS-DMA instruction every ~116 cycles, 56 KiB of x took 3.1 us from one wave.
    // Data published by other workgroups of this launch is read with sc1 loads (L1 bypass; the producers stored
    // write-through), so no acquire fence (1.7 us) sits on the hand-off path.
    const int sv = wave - DE_SW;
    const int r8 = lane & 7;
    int unit_no = 0;
    bool first_gemm = true;
    int nev = 0;
    auto stamp = [&]() {       // tuning only: the lead's timeline (s_memtime) into trace[cu][64]
        if (trace && sv == 0 && nev < 64) {
            const unsigned long long tck = __builtin_readcyclecounter();
            if (lane == 0) trace[cu * 64 + nev] = tck;
            ++nev;
        }
    };
    stamp();
    auto poll = [&](const uint32_t* cnt, int word, uint32_t target) {
        if (lane == 0) {      // one lane polls (relaxed, L2), bounded
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
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };

    for (int oi = 0; oi < nops; ++oi) {
        const auto& op = ops[oi];
        const bool publish = op.publish != 0;

        if (op.kind == UMV_DE_REDUCE) {
            // partial sums of a K-grouped GEMM -> residual stream: seq = bf16(bf16(sum_kg partial) + seq), tile by tile.
            // Tile `cu` (16 columns) belongs to workgroup cu; it waits for the kgroups producers of its n-group.
            if (cu >= op.ntiles || sv != 0) { stamp(); stamp(); stamp(); continue; }
            stamp();
            if (op.wait_cnt) poll(op.wait_cnt, cu / op.sig_div, op.wait_target);
            stamp();
            const int r = lane & 15, g = lane >> 4;
            float p = 0.f;
            if (r < M) {
                const int n0 = cu * 16 + g * 4;
                const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const float*>(op.x) + (int64_t)r * op.ldx + n0);
                const int KG = op.kgroups;
                uint16_t* rr = op.resid + (int64_t)r * op.ldr + n0;
                // every load in flight before the first add (a dependent chain of L2 round trips otherwise); sc1: L1 bypass
                const unsigned long long pkr = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long pv[DE_MAXKG][2];
#pragma unroll
                for (int s = 0; s < DE_MAXKG; ++s) {
                    const unsigned long long* q = pp + (int64_t)s * (op.split_stride / 2);
                    pv[s][0] = s < KG ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    pv[s][1] = s < KG ? __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < DE_MAXKG; ++s) {
                    acc.x += __uint_as_float((uint32_t)pv[s][0]); acc.y += __uint_as_float((uint32_t)(pv[s][0] >> 32));
                    acc.z += __uint_as_float((uint32_t)pv[s][1]); acc.w += __uint_as_float((uint32_t)(pv[s][1] >> 32));
                }
                const uint32_t rlo = (uint32_t)pkr, rhi = (uint32_t)(pkr >> 32);
                const float v0 = rbf(rbf(acc.x) + __uint_as_float(rlo << 16)), v1 = rbf(rbf(acc.y) + __uint_as_float(rlo & 0xFFFF0000u));
                const float v2 = rbf(rbf(acc.z) + __uint_as_float(rhi << 16)), v3 = rbf(rbf(acc.w) + __uint_as_float(rhi & 0xFFFF0000u));
                de_st64(rr, pack2bf(v0, v1), pack2bf(v2, v3), publish);
                p = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
            if (op.ss_out) {
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                if (g == 0 && r < DE_MROWS) de_st32f(op.ss_out + cu * DE_MROWS + r, p, publish);
            }
            if (op.sig_cnt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(op.sig_cnt + (cu & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stamp();
            continue;
        }

        // ---------------------------------------------------------------- GEMM op
        const DeShare S = de_share(op, cu, G);
        if (S.t0 >= S.t1) { stamp(); stamp(); stamp(); stamp(); continue; }
        const bool dep = op.wait_cnt != nullptr;
        stamp();      // x (and ss_in) are produced inside this launch
        const bool fast_norm = op.norm_w && op.ss_in;      // the producers published their rows' sums of squares
        const bool self_x = first_gemm && !dep && !op.norm_w; // the streaming waves copy x themselves (first op of the launch)
        first_gemm = false;
        const int c8 = lane >> 3;
        constexpr int SLOTS = DE_SW * DE_XK * 4 * DE_MROWS / (64 * DE_NSV);     // 16-byte x slots per lane and service wave (14)
        if (fast_norm) {
            // ---- Qwen2RMSNorm on the way in, without a round trip through LDS: every service wave takes a quarter of the x slots
            //      (slot u = ((slice*14 + kk)*4 + g)*8 + row), fetches the norm weights of its slots BEFORE it waits for the
            //      producers (they depend on nobody), then - one batch, one memory round trip - the producers' sums of squares and
            //      its x slots with sc1 loads, derives the scale of its lanes' rows, normalises in registers and writes LDS once.
            {   // norm weights -> LDS [k-tile of the workgroup's range][g], every service wave a part; they are all there after B0
                const int nslots = S.nkt * 4;
                for (int q0 = sv * 64; q0 < nslots; q0 += 64 * DE_NSV) {
                    const int q = q0 + lane;
                    const char* src = q < nslots ? reinterpret_cast<const char*>(op.norm_w + ((int64_t)S.kbase * 32 + q * 8)) : reinterpret_cast<const char*>(zeros);
                    __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(nws + q0 * 16), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            auto slot_src = [&](int it) -> const uint16_t* {
                const int u = (sv * SLOTS + it) * 64 + lane;
                const int chunk = u >> 3;
                const int w = chunk / (DE_XK * 4), rem = chunk - w * (DE_XK * 4);
                const int kk = rem >> 2, gq = rem & 3;
                const int kb = min(S.nkt, w * S.kt_per), ke = min(S.nkt, kb + S.kt_per);
                const bool ok = (kb + kk < ke) && r8 < M;
                return ok ? op.x + (int64_t)r8 * op.ldx + (int64_t)(S.kbase + kb + kk) * 32 + gq * 8 : zeros;
            };
            if (sv == 0 && dep) poll(op.wait_cnt, op.wait_mode ? S.kg : -1, op.wait_target);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // B0: the inputs of this op are visible (and the norm weights in LDS)
            stamp();
            float sq[DE_SS_PER_LANE];
#pragma unroll
            for (int i = 0; i < DE_SS_PER_LANE; ++i)
                sq[i] = (c8 + i * 8 < op.ss_n) ? __uint_as_float(de_ld_relaxed(reinterpret_cast<const uint32_t*>(op.ss_in + (c8 + i * 8) * DE_MROWS + r8))) : 0.f;
            // x slots of this wave: LDS-DMA (sc1: the rows were published by other workgroups) into its own, lane-linear part of xs
#pragma unroll
            for (int it = 0; it < SLOTS; ++it)
                __builtin_amdgcn_global_load_lds((const void*)slot_src(it), (de_lds_ptr_t)(xs + (sv * SLOTS + it) * 1024), 16, 0, 16 /* sc1 */);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < DE_SS_PER_LANE; ++i) ss += sq[i];
            ss += __shfl_xor(ss, 8, 64);
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rstd = rsqrt_ieee(ss / (float)(op.KT * 32) + op.norm_eps);       // of row r8 = lane & 7 = the row of this lane's slots
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll 2
            for (int it = 0; it < SLOTS; ++it) {
                const int u = (sv * SLOTS + it) * 64 + lane;
                const int chunk = u >> 3;
                const int w = chunk / (DE_XK * 4), rem = chunk - w * (DE_XK * 4);
                const bf16x8 nw = *reinterpret_cast<const bf16x8*>(nws + ((w * S.kt_per + (rem >> 2)) * 4 + (rem & 3)) * 16);
                bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + u * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)nw[j]) * rbf(bf2f((bf16_t)v[j]) * rstd));
                *reinterpret_cast<bf16x8*>(xs + u * 16) = v;
            }
        } else {
            // (a) the norm weights do not depend on anybody: stage them first.  [k-tile of the workgroup's range][g] 16 bytes
            if (op.norm_w) {
                const int nslots = S.nkt * 4;
                for (int q0 = sv * 64; q0 < nslots; q0 += 64 * DE_NSV) {
                    const int q = q0 + lane;
                    const char* src = q < nslots ? reinterpret_cast<const char*>(op.norm_w + ((int64_t)S.kbase * 32 + q * 8)) : reinterpret_cast<const char*>(zeros);
                    __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(nws + q0 * 16), 16, 0, 0);
                }
            }
            // (b) the producers of x
            if (dep) {
                if (sv == 0) poll(op.wait_cnt, op.wait_mode ? S.kg : -1, op.wait_target);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // B0: the inputs of this op are visible
            }
            stamp();
            // (c) x rows -> LDS, already as the B fragments of every wave slice: slot ((w*14 + kk)*4 + g)*8 + r
            if (!self_x && !(DBG & 8)) {
                const int pair_l = lane >> 5, gq = (lane >> 3) & 3;
                for (int i = sv; i < DE_SW * DE_XK / 2; i += DE_NSV) {
                    // skip instructions whose two (w, kk) pairs are both beyond their slices (wave-uniform)
                    const int w0 = (i * 2) / DE_XK, kk0 = (i * 2) - w0 * DE_XK;
                    const int w1 = (i * 2 + 1) / DE_XK, kk1 = (i * 2 + 1) - w1 * DE_XK;
                    if (kk0 >= de_slice_nk(S, w0) && kk1 >= de_slice_nk(S, w1)) continue;
                    const int w = pair_l ? w1 : w0, kk = pair_l ? kk1 : kk0;
                    const int kb = min(S.nkt, w * S.kt_per), ke = min(S.nkt, kb + S.kt_per);
                    const bool ok = (kb + kk < ke) && r8 < M;
                    const char* src = ok ? reinterpret_cast<const char*>(op.x + (int64_t)r8 * op.ldx + (int64_t)(S.kbase + kb + kk) * 32 + gq * 8)
                                         : reinterpret_cast<const char*>(zeros);
                    if (dep) __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(xs + i * 1024), 16, 0, 16 /* sc1 */);
                    else __builtin_amdgcn_global_load_lds((const void*)src, (de_lds_ptr_t)(xs + i * 1024), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (op.norm_w) {      // fallback: x was complete before the launch and nobody published its sums of squares
                __builtin_amdgcn_s_barrier();                   // raw x and the norm weights are in LDS (every service wave copied a part)
                float ss = 0.f;
                if (sv == 0) {
                    if (!(DBG & 16)) {
#pragma unroll 8
                        for (int q = c8; q < DE_SW * DE_XK * 4; q += 8) {
                            const bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + (q * DE_MROWS + r8) * 16);
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const float f = bf2f((bf16_t)v[j]);
                                ss += f * f;
                            }
                        }
                    }
                    ss += __shfl_xor(ss, 8, 64);
                    ss += __shfl_xor(ss, 16, 64);
                    ss += __shfl_xor(ss, 32, 64);
                    if (lane < DE_MROWS) misc[lane] = rsqrt_ieee(ss / (float)(op.KT * 32) + op.norm_eps);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // the scale of every row is known
                // Qwen2RMSNorm in place, two bf16 roundings (modeling_qwen2.py:89-94): x <- bf16(w * bf16(x * rstd))
                const float rstd = misc[r8];
#pragma unroll 2
                for (int it = 0; it < SLOTS; ++it) {
                    const int u = (sv * SLOTS + it) * 64 + lane;
                    const int chunk = u >> 3;
                    const int w = chunk / (DE_XK * 4), rem = chunk - w * (DE_XK * 4);
                    const int kk = rem >> 2, gq = rem & 3;
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(xs + u * 16);
                    const bf16x8 nw = *reinterpret_cast<const bf16x8*>(nws + ((w * S.kt_per + kk) * 4 + gq) * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(bf2f((bf16_t)nw[j]) * rbf(bf2f((bf16_t)v[j]) * rstd));
                    *reinterpret_cast<bf16x8*>(xs + u * 16) = v;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // B1: x is ready in LDS
        stamp();

        // (e) finish the tiles as the streaming waves deliver them (the lead; the others only keep the barrier count)
        const int mul = op.pair ? 2 : 1;
        const int g = (lane >> 3) & 3, m = r8;               // lanes 0..31: (g, row) of the compacted red layout
        for (int t = S.t0; t < S.t1; t += mul) {
            // bias / residual of this tile: requested before the tile is complete, consumed after
            u32x2 pk_b = {0u, 0u}, pk_r = {0u, 0u};
            if (sv == 0 && !op.pair && op.epi != UMV_DE_EPI_PARTIAL && lane < 32 && m < M) {
                const int n0 = t * 16 + g * 4;
                if (op.bias) pk_b = *reinterpret_cast<const u32x2*>(op.bias + n0);
                if (op.epi == UMV_DE_EPI_RESIDUAL) pk_r = *reinterpret_cast<const u32x2*>(op.resid + (int64_t)m * op.ldr + n0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const f32x4* rb = red + ((unit_no & 1) * DE_SW * 2) * 32 + (lane & 31);
            ++unit_no;
            if (sv != 0) continue;
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            if (lane < 32) {
#pragma unroll
                for (int w = 0; w < DE_SW; ++w) {
                    s0 += rb[(w * 2 + 0) * 32];
                    if (mul == 2) s1 += rb[(w * 2 + 1) * 32];
                }
            }
            float p = 0.f;
            if (lane < 32 && m < M) {
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
                        const u32x2 pk = pk_b;
                        v[0] += __uint_as_float(pk.x << 16); v[1] += __uint_as_float(pk.x & 0xFFFF0000u);
                        v[2] += __uint_as_float(pk.y << 16); v[3] += __uint_as_float(pk.y & 0xFFFF0000u);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = rbf(v[j]);
                    uint16_t* o = reinterpret_cast<uint16_t*>(op.out) + (int64_t)m * op.ldo + n0;
                    if (op.epi == UMV_DE_EPI_RESIDUAL) {
                        const u32x2 pk = pk_r;
                        v[0] = rbf(v[0] + __uint_as_float(pk.x << 16)); v[1] = rbf(v[1] + __uint_as_float(pk.x & 0xFFFF0000u));
                        v[2] = rbf(v[2] + __uint_as_float(pk.y << 16)); v[3] = rbf(v[3] + __uint_as_float(pk.y & 0xFFFF0000u));
                    }
                    de_st64(o, pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), publish);
                    p = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
            }
            if (op.ss_out) {      // per-row sum of squares of this tile's final values: lanes (g, m) -> m
                p += __shfl_xor(p, 8, 64);
                p += __shfl_xor(p, 16, 64);
                if (lane < DE_MROWS) de_st32f(op.ss_out + t * DE_MROWS + lane, p, publish);
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
        stamp();
    }
}

extern "C" size_t umv_decode_engine_counter_words(void) { return 16 * 64; }

extern "C" int umv_decode_engine(const umv_de_op* ops_dev, int nops, int M, uint32_t* counters, size_t counter_words, uint32_t* err,
                                 const uint16_t* zero_kib, int grid, umv_stream_t stream) {
    return umv_decode_engine_traced(ops_dev, nops, M, counters, counter_words, err, zero_kib, grid, nullptr, stream);
}

extern "C" int umv_decode_engine_traced(const umv_de_op* ops_dev, int nops, int M, uint32_t* counters, size_t counter_words, uint32_t* err,
                                        const uint16_t* zero_kib, int grid, unsigned long long* trace, umv_stream_t stream) {
    UMV_CHECK(ops_dev && nops > 0 && err && zero_kib, UMV_ERR_ARG, "decode_engine: null pointer");
    UMV_CHECK(M >= 1 && M <= DE_MROWS, UMV_ERR_UNSUPPORTED, "decode_engine: M=%d (1..%d rows)", M, DE_MROWS);
    UMV_CHECK(grid >= 8 && grid <= 1024 && grid % 8 == 0, UMV_ERR_ARG, "decode_engine: grid=%d", grid);
    hipStream_t s = (hipStream_t)stream;
    const char* dbg_env = getenv("UMV_DE_DBG");     // tuning only
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    auto kern = decode_engine_kernel<0>;
    switch (dbg) {
        case 1: kern = decode_engine_kernel<1>; break;
        case 4: kern = decode_engine_kernel<4>; break;
        case 9: kern = decode_engine_kernel<9>; break;
        case 17: kern = decode_engine_kernel<17>; break;
        case 25: kern = decode_engine_kernel<25>; break;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DE_THREADS), DE_LDS_BYTES, s, ops_dev, nops, M, grid, err, zero_kib, trace);
    UMV_LAUNCH_CHECK();
    return UMV_OK;
}
