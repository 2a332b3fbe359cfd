/*
 * libunimedvl_hip_experimental.so - kernels that were built, checked against the product kernels and MEASURED, but are
 * not on the product path (profiles/HISTORY.md section 5b says why each of them lost); kept, with their tests, as the starting
 * point for whoever continues.  Nothing in unimedvl_amd's default path loads this library.  Same conventions as
 * unimedvl_hip.h (device pointers, asynchronous on `stream`, 0 / negative return, message from umv_exp_last_error()).
 */
#ifndef UNIMEDVL_HIP_EXPERIMENTAL_H
#define UNIMEDVL_HIP_EXPERIMENTAL_H
#include "unimedvl_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* umv_exp_last_error(void);

/* ------------------------------------------------------------------ decode GEMM (M <= 16; Bagel.generate_text,
 * bagel.py:1262-1314 -> the F.linear calls of one-token-per-sample steps: qwen2_navit.py:541-543,617-620,
 * modeling_qwen2.py:234-235, bagel.py:1295).  One persistent workgroup per CU streams ONE contiguous slab of a
 * "decode image": workgroup w owns output channels [w*C, (w+1)*C), C = ceil(rows/G), as tpw tiles of th <= 16 rows,
 *   bf16: D[w][tile][k/32][(k%32)/8][r < th][k%8]     e4m3: D8[w][tile][k/64][(k%32)/8][r < th][(k%64)/32][k%8]
 * (SwiGLU: every tile is a (gate, up) pair; rows = I).  x - optionally Qwen2RMSNorm(x)*norm_w, fused - is fetched once
 * per workgroup.  Same K split and summation order as umv_gemm_bf16 / umv_gemm_fp8w at M <= 16: bit-identical results. */
typedef struct {
    int G;   /* workgroups = slabs (the CU count of the device: 256 on MI355X) */
    int C;   /* channels per slab */
    int th;  /* rows per tile */
    int tpw; /* tiles per slab (the last one may be ragged) */
} umv_decode_layout;
int umv_decode_layout_for(int rows, int G, umv_decode_layout* out);
size_t umv_decode_image_bytes(int K, int swiglu, int fp8, const umv_decode_layout* L);
/* packed16: the standard image (umv_pack_weight_bf16 / _swiglu_bf16, or the e4m3 image with scale16 = its scales);
 * scale_out: f32 [G*tpw*(swiglu?2:1)*16] (e4m3 only) */
int umv_repack_weight_decode(const void* packed16, const float* scale16, void* out, float* scale_out, int rows, int K,
                             int swiglu, int fp8, const umv_decode_layout* L, umv_stream_t stream);
/* a->wp = decode image (a->w_scale = its scales when fp8); a->tile_rows is ignored; a->norm_w needs K <= 4096 */
int umv_gemm_decode(const umv_gemm_args* a, const umv_decode_layout* L, int fp8, umv_stream_t stream);

/* ------------------------------------------------------------------ decode layer engine (M <= 16): a CHAIN of the
 * weight-streaming linears of one decode step as ONE persistent launch - the token loop of Bagel.generate_text
 * (bagel.py:1262-1314) through Qwen2MoTDecoderLayer.forward_inference (qwen2_navit.py:843-902): o_proj + residual
 * (:617-620,873-874), post_attention_layernorm (:876), mlp gate/up/SwiGLU/down (modeling_qwen2.py:234-235), residual
 * (:897-898), the next layer's input_layernorm (:862) and q/k/v_proj (:541-543).  One workgroup per CU keeps the weight
 * stream of op n+1 in flight (LDS-DMA ring, 14 KiB per wave) while op n's results travel between the workgroups through
 * global memory (write-through stores + arrival counters + one agent-scope acquire per consumer); results equal
 * umv_gemm_bf16 at M <= 16 (same K slices per wave, same summation order, same bf16 rounding points) followed by
 * umv_residual_rmsnorm_bf16 / umv_rmsnorm_bf16 up to the order of the fp32 row sums of squares.  M <= 8 rows.
 * Ops run in array order; `ops` lives in DEVICE memory (it is read with scalar loads by every workgroup). */
enum { UMV_DE_GEMM = 0, UMV_DE_REDUCE = 1 };
enum { UMV_DE_EPI_BF16 = 0, UMV_DE_EPI_RESIDUAL = 1, UMV_DE_EPI_PARTIAL = 2 };
enum { UMV_DE_SIG_XCD = 0, UMV_DE_SIG_UNIT_DIV = 1, UMV_DE_SIG_GROUP_END = 2 };
typedef struct {
    const uint16_t* w;       /* GEMM: packed weight image (umv_pack_weight_bf16 / umv_pack_weight_swiglu_bf16) */
    const uint16_t* x;       /* GEMM: input rows bf16 [M, ldx]; REDUCE: fp32 partial sums [kgroups][M][ldx] (as float*) */
    int64_t ldx;
    const uint16_t* norm_w;  /* GEMM, optional: x = Qwen2RMSNorm(x rows) * norm_w on the way in (K = KT*32, all of K) */
    float norm_eps;
    int32_t kind;            /* UMV_DE_GEMM / UMV_DE_REDUCE */
    const uint16_t* bias;    /* GEMM, optional [N] */
    uint16_t* resid;         /* EPI_RESIDUAL: residual rows (read); REDUCE: residual stream, updated in place */
    int64_t ldr;
    void* out;               /* bf16 [M, ldo]; EPI_PARTIAL: fp32 [kgroups][M][ldo] with split_stride floats between groups */
    int64_t ldo;
    int64_t split_stride;
    const uint32_t* wait_cnt;/* optional: counter words (stride 16 words) that must reach wait_target before x is read */
    uint32_t* sig_cnt;       /* optional: counter words bumped when a finished unit's stores are visible */
    float* ss_out;           /* optional [ntiles][8]: per-row sums of squares of the FINAL bf16 values of each finished 16-column
                                tile (EPI_RESIDUAL / REDUCE): the statistics of the RMSNorm the next op applies on the way in */
    const float* ss_in;      /* norm_w set: the producers' ss_out, summed over ss_n tiles in a fixed order (NULL: the rows are
                                squared here - only possible when x was complete before the launch) */
    uint32_t wait_target;
    int32_t wait_mode;       /* 0: sum of the 8 XCD shard words; 1: the word of this workgroup's K group */
    int32_t sig_mode;        /* UMV_DE_SIG_*: word = XCD shard / unit index / sig_div / n-group (once, after the last unit) */
    int32_t sig_div;
    int32_t KT;              /* K / 32 */
    int32_t ntiles;          /* 16-column tiles of the image (a SwiGLU pair counts 2); REDUCE: 16-column tiles of the row */
    int32_t pair;            /* 1: (gate, up) tile pairs, SwiGLU epilogue, out = [M, ntiles/2*16] */
    int32_t kgroups;         /* 1, or G % kgroups == 0: workgroup cu takes K group cu % kgroups of n-group cu / kgroups */
    int32_t rot;             /* rotation of the n-group -> unit-range map (balances ops whose unit count is not a multiple) */
    int32_t epi;             /* UMV_DE_EPI_* */
    int32_t publish;         /* 1: the outputs are read by other workgroups of THIS launch (write-through stores) */
    int32_t ss_n;
} umv_de_op;
size_t umv_decode_engine_counter_words(void);
/* counters: `counter_words` words zeroed on `stream` ahead of the launch (may be NULL when no op waits or signals);
 * err: one word, non-zero after a bounded wait timed out (0xDE00xxxx); dummy_kib: 1 KiB of ZEROS in device memory;
 * grid: workgroups = CUs (256 on MI355X), a multiple of 8 - every workgroup must be resident. */
int umv_decode_engine(const umv_de_op* ops_dev, int nops, int M, uint32_t* counters, size_t counter_words, uint32_t* err,
                      const uint16_t* dummy_kib, int grid, umv_stream_t stream);
/* tuning only: trace (optional) = [grid][64] 64-bit s_memtime stamps of every workgroup's lead service wave: launch entry, then
 * per GEMM op {entry, producers seen, x staged, last tile published}, per REDUCE op {entry, producers seen, published} */
int umv_decode_engine_traced(const umv_de_op* ops_dev, int nops, int M, uint32_t* counters, size_t counter_words, uint32_t* err,
                             const uint16_t* dummy_kib, int grid, unsigned long long* trace, umv_stream_t stream);


/* One decode step's attention with umv_qkv_post folded in (one query token per segment, hd = 128, `und` chain): the wave
 * of a (segment, kv head, key split) normalises + rotates its G query heads and the new key straight from the raw fused QKV
 * row (qwen2_navit.py:544-583), appends K / V^T at slot kv_len-1 (:585-600) and attends over kv_len keys (:605-614); splits
 * are merged as in umv_attn_varlen.  Same MFMAs on the same operands as qkv_post + attn_varlen; only the row sum of squares
 * of the q/k norms is accumulated in another order. */
typedef struct {
    const uint16_t* qkv;   /* [nseg, (nq + 2 nkv) * hd] raw QKV GEMM output (bias applied), row stride ld_qkv */
    int64_t ld_qkv;
    uint16_t* out;         /* [nseg, nq, hd] */
    const int32_t* cu_q;   /* [nseg + 1] = 0..nseg (one token per segment; used by the split merge) */
    const int32_t* kv_len; /* [nseg] keys per segment INCLUDING this step's token */
    const int32_t* tok_pos;/* [nseg] rope position of this step's token */
    const uint16_t* q_norm_w; const uint16_t* k_norm_w; /* [hd] */
    const uint16_t* cos_tab; const uint16_t* sin_tab;   /* [max_pos, hd] bf16 */
    uint16_t* k_slab; uint16_t* vt_slab;
    int64_t k_seg_stride, k_head_stride, v_seg_stride, v_head_stride, v_d_stride;
    int nseg, nq, nkv, hd;
    float eps;
    int nsplit;
    void* workspace;       /* umv_attn_workspace_bytes(nseg, nq, hd, 1, nsplit) when nsplit > 1 */
    /* optional: the QKV row as the fp32 partial sums of a split-K umv_gemm_bf16 / umv_gemm_fp8w (then `qkv` may be NULL):
     * x = bf16(sum_s P[s][row][col] + bias[col]), exactly what umv_qkv_post does with the same fields */
    const float* qkv_partials; /* [n_splits][nseg, (nq + 2 nkv) * hd] fp32, row stride ld_qkv, split stride split_stride */
    int n_splits;
    int64_t split_stride;
    const uint16_t* qkv_bias;  /* [(nq + 2 nkv) * hd] or NULL (only read with qkv_partials) */
} umv_attn_decode_args;
int umv_attn_decode_fused(const umv_attn_decode_args* a, umv_stream_t stream);

/* Prefill attention on the 32x32x16 matrix instruction (csrc/attention_prefill32.hip): umv_attn_varlen's arguments, nsplit = 1 and
 * hd = 128 only (flash_attn_varlen_func at qwen2_navit.py:605-614).  One wave owns 32 q columns, K Q^T / softmax / P V of three
 * consecutive key blocks are software-pipelined, K / V^T blocks reach LDS as contiguous 1 KiB LDS-DMA pieces with a source-side
 * swizzle.  Correct (tests/test_attn_prefill32_gpu.py) and as fast as the shipped attn_prefill_kernel, not faster
 * (profiles/HISTORY.md section 5b has the counters and the ablation): kept here, not on the product path. */
int umv_attn_prefill32(const umv_attn_args* a, umv_stream_t stream);

/* Stream `bytes` at `ptr` through the cache hierarchy (no compute) so that they are resident in the
 * 256 MiB Infinity Cache for a later kernel; meant for a parallel stream / graph branch during the
 * latency-bound kernels of a decode step.  `sink` (4 bytes, may be NULL) only keeps the loads alive. */
int umv_prefetch(const void* ptr, size_t bytes, int blocks, void* sink, umv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
