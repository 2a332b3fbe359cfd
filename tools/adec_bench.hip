// How fast can ONE workgroup per (segment, kv head) stream a decode-attention context?  (profiles/HISTORY.md section 5b: the fused
// workgroup kernel took 33 us at B=8 / 1.1k keys because its MFMA-fragment-direct loads touch 16 cache lines per
// quarter-wave.)  Two load patterns over the same bytes, same MFMAs, same LDS merge:
//   mode 0  row-major K [cap][128] and V^T [128][cap] - the slab layout of today
//   mode 1  fragment-major 32-key blocks: every load instruction reads 1 KiB contiguous
// Build: hipcc --offload-arch=gfx950 -O3 tools/adec_bench.hip -o tools/bin/adec_bench ; run: tools/bin/adec_bench [B] [ctx]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int NW = 8, HD = 128, KS = 4, DT = 8, PST = HD + 4;

template <int MODE>
__global__ __launch_bounds__(NW * 64) void adec_kernel(const short* __restrict__ k, const short* __restrict__ vt, float* __restrict__ out,
                                                        int Lk, int cap, int nkv) {
    __shared__ __attribute__((aligned(16))) float part[NW][8][PST];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int s = blockIdx.y, kh = blockIdx.x;
    const short* kbase = k + ((size_t)s * nkv + kh) * cap * HD;
    const short* vbase = vt + ((size_t)s * nkv + kh) * cap * HD;
    bf16x8 kst[2][2][KS], vst[2][DT];
    auto load_kv = [&](int kb, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[DT]) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const short* p;
                if (MODE == 0) p = kbase + (size_t)(kb + (j >> 2) * 8 + t * 4 + (j & 3)) * HD + ks * 32 + g * 8;
                else p = kbase + (size_t)kb * HD + (t * KS + ks) * 512 + lane * 8;
                kf[t][ks] = *reinterpret_cast<const bf16x8*>(p);
            }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const short* p;
            if (MODE == 0) p = vbase + (size_t)(dt * 16 + j) * cap + kb + g * 8;
            else p = vbase + (size_t)kb * HD + dt * 512 + lane * 8;
            vf[dt] = *reinterpret_cast<const bf16x8*>(p);
        }
    };
    constexpr int STEP = NW * 32;
    const int kb0 = wave * 32;
    if (kb0 < Lk) load_kv(kb0, kst[0], vst[0]);
    if (kb0 + STEP < Lk) load_kv(kb0 + STEP, kst[1], vst[1]);
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ks][e] = (short)(0x3c00 + ((lane * 7 + ks * 3 + e) & 63));
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    auto block = [&](int kb, bf16x8 (&kf)[2][KS], bf16x8 (&vf)[DT]) {
        f32x4 st[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            st[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][ks], qf[ks], st[t], 0, 0, 0);
        }
        float sc[8], mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = st[t][r] * 0.01f;
                v = (kb + g * 8 + t * 4 + r) < Lk ? v : -INFINITY;
                sc[t * 4 + r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        bf16x8 pf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float p = __builtin_amdgcn_exp2f(sc[i] - m_new);
            ps += p;
            pf[i] = (short)(__float_as_uint(p) >> 16);
        }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x4 acc = o[dt];
            acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf, acc, 0, 0, 0);
        }
    };
    for (int kb = kb0; kb < Lk; kb += 2 * STEP) {
        block(kb, kst[0], vst[0]);
        if (kb + 2 * STEP < Lk) load_kv(kb + 2 * STEP, kst[0], vst[0]);
        if (kb + STEP < Lk) {
            block(kb + STEP, kst[1], vst[1]);
            if (kb + 3 * STEP < Lk) load_kv(kb + 3 * STEP, kst[1], vst[1]);
        }
    }
    if (j < 7) {
        float* po = &part[wave][j][0];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(po + dt * 16 + g * 4) = o[dt];
        if (g == 0) { po[HD] = m_run; po[HD + 1] = l_run; }
    }
    __syncthreads();
    for (int item = threadIdx.x; item < 7 * HD; item += NW * 64) {
        const int r = item >> 7, d = item & (HD - 1);
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) M = fmaxf(M, part[w][r][HD]);
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float m = part[w][r][HD];
            const float wgt = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
            L += wgt * part[w][r][HD + 1];
            acc += wgt * part[w][r][d];
        }
        out[(((size_t)s * nkv + kh) * 7 + r) * HD + d] = L > 0.f ? acc / L : 0.f;
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, ctx = argc > 2 ? atoi(argv[2]) : 1100, nkv = 4;
    const int cap = (ctx + 63) / 32 * 32;
    const size_t n = (size_t)B * nkv * cap * HD;
    short *k, *vt;
    float* out;
    CK(hipMalloc(&k, n * 2));
    CK(hipMalloc(&vt, n * 2));
    CK(hipMalloc(&out, (size_t)B * nkv * 7 * HD * 4));
    short* h = (short*)malloc(n * 2);
    for (size_t i = 0; i < n; ++i) h[i] = (short)(0x3c00 + (i * 2654435761u >> 26));
    CK(hipMemcpy(k, h, n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, h, n * 2, hipMemcpyHostToDevice));
    // a second buffer set to rotate through, so that nothing stays L2 resident between launches (28 layers in the real step)
    const int NBUF = 16;
    short *ks[NBUF], *vs[NBUF];
    for (int i = 0; i < NBUF; ++i) {
        CK(hipMalloc(&ks[i], n * 2));
        CK(hipMalloc(&vs[i], n * 2));
        CK(hipMemcpy(ks[i], k, n * 2, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(vs[i], vt, n * 2, hipMemcpyDeviceToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        const int reps = 200;
        for (int it = -20; it < reps; ++it) {
            if (it == 0) CK(hipEventRecord(e0));
            const int b = (it + 20) % NBUF;
            if (mode == 0) hipLaunchKernelGGL(adec_kernel<0>, dim3(nkv, B), dim3(NW * 64), 0, 0, ks[b], vs[b], out, ctx, cap, nkv);
            else hipLaunchKernelGGL(adec_kernel<1>, dim3(nkv, B), dim3(NW * 64), 0, 0, ks[b], vs[b], out, ctx, cap, nkv);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("B=%d ctx=%d mode %d (%s): %.2f us per launch (%.1f MB of K+V -> %.0f GB/s)\n", B, ctx, mode,
               mode ? "fragment-major blocks" : "row-major slabs", ms * 1e3 / reps, 2.0 * B * nkv * ctx * HD * 2 / 1e6,
               2.0 * B * nkv * ctx * HD * 2 / (ms * 1e3 / reps) / 1e3);
    }
    return 0;
}
