// How many wait states does a VALU read of a v_mfma_f32_16x16x32_bf16 result REALLY need on gfx950, alone on its SIMD and with other
// waves keeping the SIMD's matrix pipe busy?  (Round 6: the paired-call prefill attention kernel reads the second q-tile's QK^T scores
// ~17 instruction slots behind their last MFMA and gives wrong values in lanes 48-63 - the rows an MFMA writes in its last pass - now and
// then, only with >= 2 waves per SIMD.  hipcc (ROCm 7.2) places s_nop 7 = 8 wait states between this MFMA and a dependent VALU read.)
// The probe pre-loads the destination registers with a sentinel, issues PRE independent MFMAs and then the probed one (all-ones operands:
// every element of D is 32), waits exactly N wait states (s_nop), reads D with v_mov and counts sentinel / partial values per 16-lane group.
// Build: tools/build_tools.sh mfma_raw_probe     Run: tools/bin/mfma_raw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int N, int PRE, int SALU>
__global__ __launch_bounds__(256) void probe(unsigned* bad, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)0x3F80; b[i] = (short)0x3F80; }      // bf16 1.0
    const float sent = 12345.0f;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r1, r2, r3;
        // the destination holds the sentinel; PRE independent MFMAs go first (v[24:27]..), then the probed MFMA, N wait states, the reads.
        // SALU = 1: the wait states are s_mov_b32 instructions (what the attention kernel has between the MFMA and the read) instead of s_nop
        asm volatile(
            "v_mov_b32 v20, %[s]\n\tv_mov_b32 v21, %[s]\n\tv_mov_b32 v22, %[s]\n\tv_mov_b32 v23, %[s]\n\t"
            "s_nop 7\n\ts_nop 7\n\t"
            ".if %[pre] > 0\n\tv_mfma_f32_16x16x32_bf16 v[24:27], %[a], %[b], 0\n\t.endif\n\t"
            ".if %[pre] > 1\n\tv_mfma_f32_16x16x32_bf16 v[28:31], %[a], %[b], 0\n\t.endif\n\t"
            ".if %[pre] > 2\n\tv_mfma_f32_16x16x32_bf16 v[32:35], %[a], %[b], 0\n\t.endif\n\t"
            "v_mfma_f32_16x16x32_bf16 v[20:23], %[a], %[b], 0\n\t"
            ".if %[salu] == 0\n\t"
            "  .if %[n] > 16\n\ts_nop 15\n\ts_nop %[n] - 17\n\t.elseif %[n] > 0\n\ts_nop %[n] - 1\n\t.endif\n\t"
            ".else\n\t"
            "  .rept %[n]\n\ts_mov_b32 s40, 0\n\t.endr\n\t"
            ".endif\n\t"
            "v_mov_b32 %[r0], v20\n\tv_mov_b32 %[r1], v21\n\tv_mov_b32 %[r2], v22\n\tv_mov_b32 %[r3], v23\n\t"
            "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
            : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
            : [s] "v"(sent), [a] "v"(a), [b] "v"(b), [n] "n"(N), [pre] "n"(PRE), [salu] "n"(SALU)
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "s40", "memory");
        nbad += (r0 != 32.0f) + (r1 != 32.0f) + (r2 != 32.0f) + (r3 != 32.0f);
    }
    if (nbad) atomicAdd(bad + (threadIdx.x & 63) / 16, nbad);         // per 16-lane group (D rows 4g .. 4g+3)
}

template <int N, int PRE, int SALU>
static void run(unsigned* d_bad, int occ, int iters) {
    hipMemset(d_bad, 0, 4 * sizeof(unsigned));
    hipLaunchKernelGGL((probe<N, PRE, SALU>), dim3(256 * occ), dim3(256), 0, 0, d_bad, iters);
    unsigned h[4];
    hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost);
    const double total = 256.0 * occ * 256 * iters * 4;
    printf("  N=%2d wait states (%s), %d MFMA(s) in front, %d wave(s)/SIMD: wrong reads per lane group [%u %u %u %u] of %.3g  %s\n", N,
           SALU ? "s_mov" : "s_nop", PRE, occ, h[0], h[1], h[2], h[3], total / 4, (h[0] | h[1] | h[2] | h[3]) ? "<-- STALE" : "");
}

template <int PRE, int SALU>
static void sweep(unsigned* d_bad, int occ, int iters) {
    run<0, PRE, SALU>(d_bad, occ, iters); run<2, PRE, SALU>(d_bad, occ, iters); run<4, PRE, SALU>(d_bad, occ, iters); run<6, PRE, SALU>(d_bad, occ, iters);
    run<7, PRE, SALU>(d_bad, occ, iters); run<8, PRE, SALU>(d_bad, occ, iters); run<9, PRE, SALU>(d_bad, occ, iters); run<10, PRE, SALU>(d_bad, occ, iters);
    run<11, PRE, SALU>(d_bad, occ, iters); run<12, PRE, SALU>(d_bad, occ, iters); run<14, PRE, SALU>(d_bad, occ, iters); run<16, PRE, SALU>(d_bad, occ, iters);
    run<20, PRE, SALU>(d_bad, occ, iters); run<24, PRE, SALU>(d_bad, occ, iters); run<32, PRE, SALU>(d_bad, occ, iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned* d_bad;
    hipMalloc(&d_bad, 4 * sizeof(unsigned));
    for (int occ : {1, 2, 4, 8}) {
        printf("== %d wave(s) per SIMD, probed MFMA alone\n", occ);
        sweep<0, 0>(d_bad, occ, iters);
        printf("== %d wave(s) per SIMD, one independent MFMA right in front (the second q-tile's MFMA behind the first's)\n", occ);
        sweep<1, 0>(d_bad, occ, iters);
        printf("== %d wave(s) per SIMD, three independent MFMAs in front\n", occ);
        sweep<3, 0>(d_bad, occ, iters);
        printf("== %d wave(s) per SIMD, one MFMA in front, SALU instructions as the wait states\n", occ);
        sweep<1, 1>(d_bad, occ, iters);
    }
    return 0;
}
