// Sanitizer driver for the C-ABI shim of libunimedvl_hip (SURVEY.md section 5 "race detection / sanitizers"; VERDICT r05 "missing" #4).
// Linked against a HOST-sanitized build of the library (tools/sanitize_abi.sh: -fsanitize=address,undefined resp. -fsanitize=thread, the
// device code is not instrumented).  It needs no GPU: it exercises everything the shim does BEFORE a launch -
//   * every host-only query over a sweep of shapes (sizes, tile / kernel policies: the lazily initialised, read-once policies),
//   * every entry point with NULL structs, NULL pointers and out-of-range dimensions (the UMV_CHECK paths, the thread-local error text),
//   * M = 0 / T = 0 early returns,
// first on one thread, then on 8 threads at once (ThreadSanitizer: the error buffer is thread-local, the policies are immutable after a
// thread-safe first use).  Exit code 0 = no entry point crashed, returned success for invalid input, or left the error text empty.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../include/unimedvl_hip.h"

static std::atomic<int> failures{0};
#define EXPECT_ERR(call)                                                                                  \
    do {                                                                                                  \
        const int rc__ = (call);                                                                          \
        const char* e__ = umv_last_error();                                                               \
        if (rc__ >= 0 || !e__ || !e__[0]) { ++failures; std::fprintf(stderr, "FAIL %s -> %d '%s'\n", #call, rc__, e__ ? e__ : "(null)"); } \
    } while (0)
#define EXPECT_OK(call)                                                                                   \
    do {                                                                                                  \
        const int rc__ = (call);                                                                          \
        if (rc__ != 0) { ++failures; std::fprintf(stderr, "FAIL %s -> %d '%s'\n", #call, rc__, umv_last_error()); } \
    } while (0)

static void queries() {
    size_t acc = 0;
    for (int n : {1, 16, 17, 1152, 3584, 4608, 37888, 152064})
        for (int k : {8, 32, 608, 1152, 3584, 18944}) {
            acc += umv_packed_weight_elems(n, k) + umv_packed_weight_fp8_bytes(n, k) + umv_packed_weight_fp8_mfma_bytes(n, k);
            for (int th : {1, 9, 14, 16}) acc += umv_repacked_weight_elems(n, k, th);
            for (int m : {1, 8, 64, 65, 272, 1026, 2064, 8208, 16500}) acc += (size_t)umv_gemm_tile_config(m, n, k);
        }
    for (int nseg : {1, 8, 32})
        for (int maxq : {1, 34, 258, 1026, 4096}) {
            acc += (size_t)umv_attn_prefill_tq(nseg, 28, 4, 128, maxq) + (size_t)umv_attn_prefill_tq(nseg, 16, 16, 72, maxq) + (size_t)umv_attn_prefill_tq(nseg, 2, 1, 64, maxq);
            for (int ns : {1, 4, 24}) acc += umv_attn_workspace_bytes(nseg, 28, 128, maxq, ns);
        }
    acc += umv_groupnorm_workspace_bytes(4, 256 * 256) + (size_t)umv_version();
    if (acc == 0) ++failures;
}

static void bad_arguments() {
    uint16_t dummy16[64] = {0};
    float dummyf[64] = {0};
    int32_t dummyi[64] = {0};
    int64_t dummyl[64] = {0};
    uint8_t dummy8[64] = {0};
    // GEMMs
    EXPECT_ERR(umv_gemm_bf16(nullptr, nullptr));
    umv_gemm_args g;
    std::memset(&g, 0, sizeof g);
    EXPECT_ERR(umv_gemm_bf16(&g, nullptr));                     // null x / wp / out
    g.x = dummy16; g.wp = dummy16; g.out = dummy16; g.M = 8; g.N = 16; g.K = 12;       // K % 8 != 0
    EXPECT_ERR(umv_gemm_bf16(&g, nullptr));
    g.K = 32; g.k_splits = 3; g.epilogue = UMV_EPI_SWIGLU;     // split-K with SwiGLU
    EXPECT_ERR(umv_gemm_bf16(&g, nullptr));
    EXPECT_ERR(umv_gemm_fp8w(nullptr, nullptr));
    std::memset(&g, 0, sizeof g);
    EXPECT_ERR(umv_gemm_fp8w(&g, nullptr));
    EXPECT_ERR(umv_gemm_fp8a8w(nullptr, nullptr));
    umv_gemm8_args g8;
    std::memset(&g8, 0, sizeof g8);
    EXPECT_ERR(umv_gemm_fp8a8w(&g8, nullptr));
    // packing / quantising
    EXPECT_ERR(umv_pack_weight_bf16(nullptr, nullptr, 16, 32, nullptr));
    EXPECT_ERR(umv_repack_weight_rows_bf16(nullptr, nullptr, 16, 32, 14, nullptr));
    EXPECT_ERR(umv_pack_weight_swiglu_bf16(nullptr, nullptr, nullptr, 16, 32, nullptr));
    EXPECT_ERR(umv_quantize_pack_weight_fp8(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 16, 512, nullptr));
    EXPECT_ERR(umv_repack_weight_fp8_mfma(nullptr, nullptr, 16, 128, nullptr));
    EXPECT_ERR(umv_quantize_act_fp8(nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, 8, 128, nullptr));
    // attention / qkv_post
    EXPECT_ERR(umv_attn_varlen(nullptr, nullptr));
    umv_attn_args a;
    std::memset(&a, 0, sizeof a);
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.q = dummy16; a.out = dummy16; a.cu_q = dummyi; a.kv_len = dummyi; a.k_slab = dummy16; a.vt_slab = dummy16;
    a.nseg = 1; a.nq = 28; a.nkv = 5; a.hd = 128; a.max_q = 1; a.max_kv = 32; a.nsplit = 1;        // nq % nkv != 0
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.nkv = 4; a.nsplit = 4;                                     // split without a workspace
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.nsplit = 1; a.v_d_stride = 36;                             // capacity not a multiple of 8
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.v_d_stride = 64; a.k_key_stride = 3584; a.causal = 1;      // packed K is the non-causal form
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.k_key_stride = 0; a.causal = 0; a.page_table = dummyi; a.page_table_stride = 0;               // page table without a stride
    EXPECT_ERR(umv_attn_varlen(&a, nullptr));
    a.page_table = nullptr; a.nseg = 0;
    EXPECT_OK(umv_attn_varlen(&a, nullptr));                     // nothing to do
    EXPECT_ERR(umv_qkv_post(nullptr, nullptr));
    umv_qkv_post_args p;
    std::memset(&p, 0, sizeof p);
    EXPECT_ERR(umv_qkv_post(&p, nullptr));
    p.qkv = dummy16; p.q_out = dummy16; p.k_slab = dummy16; p.vt_slab = dummy16; p.tok_seg = dummyi; p.tok_slot = dummyi;
    p.q_norm_w = dummy16; p.T = 4; p.nq = 28; p.nkv = 4; p.hd = 128;                                   // norm without rope tables
    EXPECT_ERR(umv_qkv_post(&p, nullptr));
    // row kernels
    EXPECT_ERR(umv_rmsnorm_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, 8, 3584, 1e-6f, nullptr));
    EXPECT_ERR(umv_residual_rmsnorm_bf16(nullptr, 4, 0, 0, nullptr, nullptr, nullptr, 8, 3584, 1e-6f, nullptr));
    EXPECT_ERR(umv_layernorm_bf16(nullptr, nullptr, nullptr, nullptr, 8, 1152, 1e-6f, nullptr));
    EXPECT_ERR(umv_layernorm_bf16(dummy16, dummy16, dummy16, dummy16, 8, 1153, 1e-6f, nullptr));    // H % 8 != 0
    EXPECT_ERR(umv_embed_gather_bf16(nullptr, nullptr, nullptr, nullptr, 8, 3584, nullptr));
    EXPECT_ERR(umv_add_rows_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 8, 3584, nullptr));
    EXPECT_ERR(umv_argmax_bf16(nullptr, 0, nullptr, 8, 152064, nullptr));
    EXPECT_ERR(umv_sample_bf16(nullptr, 0, nullptr, 8, 152064, 1.0f, 1, nullptr, nullptr));
    EXPECT_ERR(umv_sample_bf16(dummy16, 64, dummyl, 1, 64, 0.0f, 1, nullptr, nullptr));           // temperature 0
    EXPECT_OK(umv_sample_bf16(dummy16, 64, dummyl, 0, 64, 1.0f, 1, nullptr, nullptr));             // no rows
    EXPECT_ERR(umv_cast_pad_f32_bf16(nullptr, 0, nullptr, 0, 8, 588, 608, nullptr));
    EXPECT_ERR(umv_cast_pad_f32_bf16(dummyf, 588, dummy16, 608, 8, 588, 580, nullptr));            // Kp < K
    EXPECT_ERR(umv_patchify_f32_bf16(nullptr, 3, 448, 448, 14, nullptr, 608, 608, nullptr));
    EXPECT_ERR(umv_patchify_f32_bf16(dummyf, 3, 450, 448, 14, dummy16, 608, 608, nullptr));        // not whole patches
    EXPECT_ERR(umv_decode_advance(nullptr, nullptr, nullptr, 8, nullptr));
    EXPECT_ERR(umv_decode_step_end(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 8, 16, nullptr));
    EXPECT_ERR(umv_decode_step_end_argmax(nullptr, nullptr, nullptr, nullptr, 9504, nullptr, nullptr, nullptr, nullptr, 8, 16, nullptr));
    EXPECT_ERR(umv_timestep_embed(nullptr, nullptr, nullptr, 4, 128, nullptr));
    EXPECT_ERR(umv_cfg_renorm_euler(nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 1, 4.0f, 1.5f, 0.f, 0, 0.1f, 64, nullptr));
    // VAE
    EXPECT_ERR(umv_conv2d_nhwc_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 128, 32, 32, 128, 3, 0, nullptr));
    EXPECT_ERR(umv_conv2d_nhwc_bf16(dummy16, dummy16, nullptr, nullptr, dummy16, 1, 128, 32, 32, 128, 5, 0, nullptr));   // 5 x 5
    EXPECT_ERR(umv_groupnorm_nhwc_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1024, 128, 1e-6f, 1, nullptr));
    EXPECT_ERR(umv_nchw_f32_to_nhwc_bf16(nullptr, nullptr, 1, 3, 32, 32, 8, nullptr));
    EXPECT_ERR(umv_unpatchify_latent(nullptr, nullptr, 16, 16, 2, 16, 0.36f, 0.11f, nullptr));
    EXPECT_ERR(umv_pixels_to_u8(nullptr, nullptr, 1024, 8, nullptr));
    EXPECT_ERR(umv_softmax_rows_f32(nullptr, 0, nullptr, 0, nullptr, 16, 1024, 1.0f, nullptr));
    EXPECT_ERR(umv_softmax_rows_f32(dummyf, 1025, dummy16, 1025, dummyf, 1, 1025, 1.0f, nullptr));   // odd n
    EXPECT_ERR(umv_rowscale_f32_bf16(nullptr, 0, nullptr, nullptr, 0, 16, 512, nullptr));
    EXPECT_ERR(umv_latent_sample_patchify(nullptr, nullptr, nullptr, 0, 56, 56, 16, 28, 28, 2, 0.36f, 0.11f, nullptr));
    (void)dummy8;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 8, rounds = argc > 2 ? std::atoi(argv[2]) : 200;
    queries();
    bad_arguments();
    std::printf("single thread: %d failure(s)\n", failures.load());
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([rounds] {
            for (int r = 0; r < rounds; ++r) { queries(); bad_arguments(); }
        });
    for (auto& th : pool) th.join();
    std::printf("%d threads x %d rounds: %d failure(s) in all\n", threads, rounds, failures.load());
    return failures.load() ? 1 : 0;
}
