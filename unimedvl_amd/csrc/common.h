// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libunimedvl_hip.
// Wave = 64 lanes everywhere.  bf16 is carried as raw uint16_t bit patterns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 MFMA C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define UMV_WAVE 64

// Workgroup barrier that the COMPILER also honours as a memory barrier.  __builtin_amdgcn_s_barrier() orders side effects but not
// plain loads: an LDS read of data that the barrier publishes (a tile other waves staged by LDS-DMA behind their own counted
// s_waitcnt) may be hoisted above it - seen in attention_prefill32.hip, where the first K / V^T fragment reads of an iteration
// landed in front of the barrier and read stale fragments now and then.  The asm form with a "memory" clobber cannot move.
#define UMV_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 on the gfx950 converter (v_cvt_pk_bf16_f32: one instruction for two values instead
// of ~6 integer ops each).  Same bits as torch's .to(bfloat16) for every non-NaN input - checked over all 2^32 patterns in
// tests/test_kernels_gpu.py::test_f2bf_exhaustive - and a quiet NaN for NaN.
typedef __attribute__((ext_vector_type(2))) __bf16 umv_bf16x2_hw;
typedef __attribute__((ext_vector_type(2))) float umv_f32x2;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }   // round through bf16

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    umv_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, umv_bf16x2_hw));
}

// 2^x on the hardware unit alone (v_exp_f32, ~1 ulp; results below 2^-126 flush to 0, exp2(-inf) = 0): the softmax
// weights of the attention kernels.  exp2f() wraps the same instruction in ~5 more VALU ops of denormal handling.
__device__ __forceinline__ float umv_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 1/sqrt(x) with correctly rounded sqrt and divide (what torch.rsqrt does on CPU)
__device__ __forceinline__ float rsqrt_ieee(float x) { return __fdiv_rn(1.0f, __fsqrt_rn(x)); }

// x[l] (op) x[l ^ 16] and x[l] (op) x[l ^ 32] on the gfx950 row-swap VALU instructions (v_permlane16_swap_b32 /
// v_permlane32_swap_b32) instead of __shfl_xor's ds_bpermute round trip through the LDS crossbar (~100 cycles of dependent
// latency each - the softmax of the attention kernels has four of them per 32-key block and q-tile).  With both operands
// = x every lane ends up holding {x[l], x[l ^ 16]} (resp. ^ 32) in its two results (tools/permlane_probe.hip), in
// (even row, odd row) order; + and max are commutative, so the result is bit-identical to the shuffle form.
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// x[l ^ 8], x[l ^ 4], x[l ^ 2], x[l ^ 1] inside a 16-lane row as DPP moves (the compiler folds them into the consuming
// VALU op) instead of ds_bpermute: row_ror:8; row_shl:4 into the lanes with bit 2 clear + row_shr:4 into the others
// (bank masks 0b0101 / 0b1010); quad_perm [2,3,0,1] and [1,0,3,2].
template <int O>
__device__ __forceinline__ float row_xor(float v) {
    const int x = __float_as_int(v);
    int r;
    if constexpr (O == 8) r = __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);
    else if constexpr (O == 4) {
        r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
        r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);
    } else if constexpr (O == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
    else r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
    return __int_as_float(r);
}

__device__ __forceinline__ float wave_sum(float v) {   // same pairing and order as `for o = 32..1: v += shfl_xor(v, o)`
    v = xor32_sum(v);
    v = xor16_sum(v);
    v += row_xor<8>(v);
    v += row_xor<4>(v);
    v += row_xor<2>(v);
    v += row_xor<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor32_max(v);
    v = xor16_max(v);
    v = fmaxf(v, row_xor<8>(v));
    v = fmaxf(v, row_xor<4>(v));
    v = fmaxf(v, row_xor<2>(v));
    v = fmaxf(v, row_xor<1>(v));
    return v;
}

// ----------------------------------------------------------------------------- one online-softmax block of the attention kernels
// A lane holds 8 scores of one query row (keys kb + g*8 .. +7 of a 32-key block, masked ones = -inf), RAW (unscaled); the row's 32 scores
// sit in the 4 lanes l, l^16, l^32, l^48.  c = softmax scale * log2(e) rides in the exponent's fma: p = 2^(s c - m), m = (max s) c.  attn_kernel (attention.hip: decode / short prefill) and attn_prefill_kernel (attention_prefill.hip) both call THIS function,
// so a row's arithmetic - and its bits - do not depend on which kernel a batch shape is sent to.  (Round 5 form: two values per VALU
// instruction where the ISA has the packed op, a tree for the row sum, maxima as asm - fmaxf() on MFMA results costs a canonicalising
// v_max_f32 x, x per operand.)
typedef float umv_f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float umv_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float umv_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// HAZARD: the first consumers of the scores are inline-asm maxima, and the compiler's hazard recogniser covers MFMA-write -> VALU-read only
// for instructions it knows as VALU, not for inline asm (a v_max3 issued 2 cycles behind the last QK^T MFMA read a stale accumulator: a
// slightly low row maximum - results still accurate, the softmax is shift invariant, but no longer the bits of the other kernel).  A caller
// that passes MFMA results untouched (no compiler-visible VALU instruction in between, e.g. a mask select) must tie them to the wait
// states by hand first: attn_mfma_guard.
__device__ __forceinline__ void attn_mfma_guard(f32x4& a, f32x4& b) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void attn_mfma_guard(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("s_nop 7\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void attn_softmax_block(const float (&v)[8], float c, float& m_run, float& l_run, float& alpha, bf16x8& pf) {
    float mx = umv_max3(v[0], v[1], v[2]);
    mx = umv_max3(mx, v[3], v[4]);
    mx = umv_max3(mx, v[5], v[6]);
    mx = umv_max2(mx, v[7]);
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = umv_max2(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = umv_max2(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    // the running maximum lives in log2 units (m = max s * c): the rescale factor 2^(m_old - m_new) is then EXACTLY 1 whenever the maximum
    // did not move (an fma on the raw maxima would leave the rounding residue of m c in the exponent: alpha = 1 + 3e-7, and the "nothing
    // to rescale" test of the prefill kernel would never fire), and the split-KV partials keep their units
    const float m_new = umv_max2(m_run, mx * c);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float nmc = -m_use;
    alpha = (m_run == -INFINITY) ? 0.f : umv_exp2(m_run - m_use);
    const umv_f32x2_v c2 = {c, c}, n2 = {nmc, nmc};
    umv_f32x2_v pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const umv_f32x2_v e = __builtin_elementwise_fma((umv_f32x2_v){v[2 * i], v[2 * i + 1]}, c2, n2);      // (-inf) c + n = -inf, exp2(-inf) = 0
        pv[i] = (umv_f32x2_v){umv_exp2(e[0]), umv_exp2(e[1])};
    }
    const umv_f32x2_v s2 = (pv[0] + pv[1]) + (pv[2] + pv[3]);
    float ps = s2[0] + s2[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = pack2bf(pv[i][0], pv[i][1]);
        pf[2 * i] = (short)(w & 0xFFFFu);
        pf[2 * i + 1] = (short)(w >> 16);
    }
    ps = xor16_sum(ps);
    ps = xor32_sum(ps);
    l_run = l_run * alpha + ps;
    m_run = m_new;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// 16-byte global load of 8 bf16 as an MFMA fragment
__device__ __forceinline__ bf16x8 ldg_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero_frag() { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }

// hipFuncSetAttribute is per DEVICE: an engine on cuda:1 of the same process needs the dynamic-LDS limit raised there too, so the
// "already done" flag of a kernel is an array indexed by the current device (true = first call on this device)
#define UMV_MAX_DEVICES 64
static inline bool umv_first_on_device(bool* flags) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= UMV_MAX_DEVICES) return true;
    if (flags[d]) return false;
    flags[d] = true;
    return true;
}

// error plumbing (host side)
void umv_set_error(const char* fmt, ...);
#define UMV_CHECK(cond, code, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            umv_set_error(__VA_ARGS__);       \
            return (code);                    \
        }                                     \
    } while (0)
#define UMV_LAUNCH_CHECK()                                            \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) {                                      \
            umv_set_error("launch failed: %s", hipGetErrorString(e__)); \
            return UMV_ERR_LAUNCH;                                    \
        }                                                             \
    } while (0)

enum { UMV_OK = 0, UMV_ERR_ARG = -1, UMV_ERR_LAUNCH = -2, UMV_ERR_UNSUPPORTED = -3 };
