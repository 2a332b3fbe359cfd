/*
 * libunimedvl_hip.so - C ABI of the MI355X (gfx950) kernel library that replaces the
 * third-party native ops on UniMedVL's forward path (SURVEY.md section 2.2 / 8b "B2").
 *
 * The reference has no FFI of its own: its native work is reached through
 * flash_attn / torch.nn.functional calls from Python.  Each entry point below names
 * the reference call site(s) it stands in for (paths relative to
 * /root/reference/codes/).  All pointers are DEVICE pointers owned by the caller;
 * nothing here allocates persistent memory; every call is asynchronous on `stream`
 * (a hipStream_t passed as void*).  Return 0 on success, negative on error
 * (message from umv_last_error()); nothing throws across the ABI.
 *
 * bf16 tensors are uint16_t bit patterns.  "rows" are tokens (packed NaViT
 * sequences have no batch dimension, as in the reference).
 */
#ifndef UNIMEDVL_HIP_H
#define UNIMEDVL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* umv_stream_t;

int umv_version(void);
const char* umv_last_error(void);

/* ------------------------------------------------------------------ weights
 * nn.Linear weights [N,K] (row-major, modeling_qwen2.py:283-286, 229-231) are
 * re-tiled ONCE at load time into MFMA A-fragment order:
 *   P[n/16][k/32][lane = ((k%32)/8)*16 + n%16][k%8], zero padded to N%16==0, K%32==0
 * so that a wavefront streams 1 KiB contiguous per 16x32 tile. */
size_t umv_packed_weight_elems(int N, int K);
int umv_pack_weight_bf16(const uint16_t* w, uint16_t* packed, int N, int K, umv_stream_t stream);
/* Decode-only second copy with `th`-row tiles, Q[n/th][k/32][(k%32)/8][n%th][k%8], built from the standard
 * image: with th = N/256 (14 for N=3584, 9 for N=4608) the weight-streaming GEMM has exactly one (or two)
 * tiles per CU of the 256-CU chip instead of 224-288 16-row tiles. */
size_t umv_repacked_weight_elems(int N, int K, int th);
int umv_repack_weight_rows_bf16(const uint16_t* packed16, uint16_t* out, int N, int K, int th, umv_stream_t stream);
/* gate_proj / up_proj [I,K] each -> one packed [2I,K] with 16-row tiles interleaved
 * (gate tile t, up tile t, ...) so SwiGLU fuses into the GEMM epilogue. */
int umv_pack_weight_swiglu_bf16(const uint16_t* gate, const uint16_t* up, uint16_t* packed, int I, int K,
                                umv_stream_t stream);

/* ------------------------------------------------------------------ GEMM
 * out[m,n] = epilogue(sum_k x[m,k] * W[n,k]); fp32 accumulate on MFMA; replaces
 * F.linear at qwen2_navit.py:541-543,555-562,617-620, modeling_qwen2.py:234-235,
 * bagel.py:1295,1104,1133, siglip_navit.py:190,216-218,243,256-258,
 * modeling_utils.py:108,120-122.  Epilogue order (each step rounds to bf16 exactly
 * where the reference's bf16 tensors are materialised):
 *   v = acc (+bias) -> bf16 ; GELU_TANH / SILU -> bf16 ; SWIGLU: bf16(bf16(silu(g))*u) ;
 *   RESIDUAL: bf16(v + residual[m,n]) */
enum {
    UMV_EPI_BIAS = 1,
    UMV_EPI_GELU_TANH = 2,
    UMV_EPI_SILU = 4,
    UMV_EPI_RESIDUAL = 8,
    UMV_EPI_SWIGLU = 16, /* packed weight from umv_pack_weight_swiglu_bf16; out is [M, N/2] */
    UMV_EPI_OUT_F32 = 32 /* store raw fp32 accumulators (+bias), no rounding */
};
typedef struct {
    const uint16_t* x;        /* [M,K] bf16, row stride ldx (elements), K%8==0 */
    int64_t ldx;
    const uint16_t* wp;       /* packed weight */
    const uint16_t* bias;     /* [N] or NULL */
    const uint16_t* residual; /* [M,N] row stride ldr, or NULL */
    int64_t ldr;
    void* out;                /* bf16 (or fp32) [M,N'] row stride ldo */
    int64_t ldo;
    const int32_t* row_idx;   /* optional [M]: x, residual and out rows are row_idx[m] (MoT routing,
                                 qwen2_navit.py:552-562,619-620,891-898) */
    int M, N, K;
    int epilogue;
    const uint16_t* norm_w;   /* optional [K]: fuse Qwen2RMSNorm(x) * norm_w into the prologue (decode path,
                                 qwen2_navit.py:861,888,1164 feeding :541-543, modeling_qwen2.py:234, bagel.py:1295);
                                 needs M <= 16 and K <= 4096 */
    float norm_eps;
    int tile_rows;            /* rows per n-tile of `wp`: 0 or 16 = umv_pack_weight_bf16 image; 1..15 = an image made by
                                 umv_repack_weight_rows_bf16 (decode only, M <= 64) */
    const float* w_scale;     /* umv_gemm_fp8w only: per-channel scales written by umv_quantize_pack_weight_fp8 */
    int k_splits;             /* > 1 (umv_gemm_bf16, M <= 64, 16-row image, no SwiGLU / norm_w): split-K decode mode. K is cut
                                 into k_splits ranges; `out` is then an fp32 buffer [k_splits][rows][ldo] (split s at
                                 out + s*split_stride) of RAW partial sums - no bias, activation or residual: the consumer
                                 (umv_qkv_post partials input, umv_residual_rmsnorm_bf16) adds the splits in order 0..S-1 and
                                 finishes the row with the reference's roundings */
    int64_t split_stride;     /* elements between consecutive splits of `out` */
    uint64_t* argmax_partial; /* optional [M][ceil(N/16)] (M <= 64, bf16 out, plain 16-row image): greedy argmax as an epilogue of
                                 the lm_head GEMM (bagel.py:1295-1301).  Every 16-column tile writes one key per row:
                                 (order-preserving image of the stored bf16 logit) << 32 | (0xFFFFFFFF - column); the maximum
                                 key over a row is torch.argmax(logits) (lowest index on ties, NaN highest).  Finished by
                                 umv_decode_step_end_argmax.  The logits are still written to `out`. */
    int64_t x_rows;           /* with row_idx: number of rows of the buffer `x` points into (rows row_idx may name), 0 = unknown.
                                 The 4-wave tiles of the M > 64 path address x through 32-bit offsets and take a row-indexed call
                                 only when x_rows * ldx * 2 < 2 GiB is known; unknown keeps the 8-wave tiles (same results). */
    float sample_temperature; /* with argmax_partial, > 0: the keys are those of SAMPLING instead of greedy decoding (bagel.py:1297-1299:
                                 probs = softmax(logits / temperature), token = multinomial(probs, 1)) by the Gumbel-max rule: the key of
                                 column n orders bf16(logit / T) - ln(-ln(u)), u uniform in (0, 1] from a counter-based generator
                                 (splitmix64 of sample_seed, *sample_step, the row and n - the stream of umv_sample_bf16), so the maximum key
                                 over a row is one draw from the softmax.  0: greedy keys. */
    uint64_t sample_seed;
    const int64_t* sample_step; /* device word: the decode step (changes the draw every step under a replayed HIP graph); NULL = 0 */
} umv_gemm_args;
int umv_gemm_bf16(const umv_gemm_args* a, umv_stream_t stream);
/* Host-only query: the tiled-kernel configuration umv_gemm_bf16 picks for an M x N x K problem (0 for M <= 64, the
 * weight-streaming kernels; 266 = 256x256x32 interleaved, 268 = 256(n)x128(m), 384 = 384(n)x128(m), 270 = 128x128,
 * 64 = 128(n)x64(m)x64).
 * Lets the parity tests assert that a shape reaches the kernel variant it is meant to pin. */
int umv_gemm_tile_config(int M, int N, int K);

/* ------------------------------------------------------------------ fp8 weights (BASELINE.json configs[4]; no
 * reference counterpart: the reference only has bf16 weights, qwen2_navit.py:541-562 / modeling_qwen2.py:229-235)
 * Weight-only OCP e4m3 with one POWER-OF-TWO scale per output channel, s[n] = the smallest 2^e with
 * 448 * 2^e >= max_k |W[n,k]|, q = rne_e4m3(W / s).  W' = q * s is exact in bf16, so
 *   umv_gemm_fp8w(x, q, s)  ==  umv_gemm_bf16(x, pack(W'))   bit for bit   (K % 512 == 0)
 * and prefill / diffusion (M > 64) run umv_gemm_bf16 on the bf16 image of W' while decode streams half the bytes.
 * Image: P8[n/16][k/64][lane = ((k%32)/8)*16 + n%16][(k%64)/32][k%8] bytes, zero padded; `scale` is f32
 * [ceil(N/16)*16] in image row order.  With w_up != NULL, `w`/`w_up` are gate_proj / up_proj [rows=I, K] and the
 * image interleaves their 16-row tiles (UMV_EPI_SWIGLU).  deq / deq_up (optional) receive W' row-major. */
size_t umv_packed_weight_fp8_bytes(int N, int K);
int umv_quantize_pack_weight_fp8(const uint16_t* w, const uint16_t* w_up, uint8_t* packed8, float* scale, uint16_t* deq,
                                 uint16_t* deq_up, int rows, int K, umv_stream_t stream);
/* M <= 64 only; a->wp = the e4m3 image, a->w_scale = its scales; norm_w / tile_rows unsupported */
int umv_gemm_fp8w(const umv_gemm_args* a, umv_stream_t stream);

/* W8A8 on the fp8 matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4) for the MFMA-bound GEMMs of the fp8 mode
 * (M > 64: prefill, flow passes).  Activations are quantised per ROW the way weights are per channel:
 *   sx[m] = smallest 2^e with 448 * 2^e >= max_k |x[m,k]|,  xq = rne_e4m3(x / sx)   (umv_quantize_act_fp8; rows may be
 *   gathered through row_idx; xq is [M, ldq] row-major, zero padded to ldq, a multiple of 128)
 *   out[m,n] = epilogue(sx[m] * sw[n] * sum_k xq[m,k] * wq[n,k])                    (umv_gemm_fp8a8w, fp32 accumulate)
 * Every product and both scales are exact, so the result equals a linear on the dequantised operands up to the order
 * of the fp32 sums - that is how oracle/fp8.py restates it.  The weight image for this instruction is a re-tiling of the
 * e4m3 image: P8M[n/16][k/128][(k%32)/16][lane = ((k%128)/32)*16 + n%16][k%16]. */
size_t umv_packed_weight_fp8_mfma_bytes(int N, int K);
int umv_repack_weight_fp8_mfma(const uint8_t* packed8, uint8_t* out, int N, int K, umv_stream_t stream);
/* deq (optional, row stride ldd): the dequantised rows as bf16 at their ORIGINAL positions (row_idx[m] when given) - the input
 * of the M <= 64 kernels, which take bf16 activations; xq may then be NULL */
int umv_quantize_act_fp8(const uint16_t* x, int64_t ldx, const int32_t* row_idx, uint8_t* xq, int64_t ldq, float* x_scale,
                         uint16_t* deq, int64_t ldd, int M, int K, umv_stream_t stream);
typedef struct {
    const uint8_t* xq;        /* [M, ldq] e4m3, rows 0..M-1 (already gathered) */
    int64_t ldq;
    const float* x_scale;     /* [M] */
    const uint8_t* wp;        /* umv_repack_weight_fp8_mfma image */
    const float* w_scale;     /* scales of umv_quantize_pack_weight_fp8 (image row order) */
    const uint16_t* bias;
    const uint16_t* residual;
    int64_t ldr;
    void* out;                /* bf16 [*, ldo] */
    int64_t ldo;
    const int32_t* row_idx;   /* optional [M]: residual / out rows (MoT routing); xq / x_scale are indexed by m */
    int M, N, K;
    int epilogue;             /* UMV_EPI_* except OUT_F32 */
} umv_gemm8_args;
int umv_gemm_fp8a8w(const umv_gemm8_args* a, umv_stream_t stream);

/* ------------------------------------------------------------------ norms / elementwise */
/* Qwen2RMSNorm (modeling_qwen2.py:89-94): out = w * bf16(x * rsqrt(mean(x^2)+eps)).
 * expert (optional, [T] int32): rows with expert[t]!=0 use w_gen (MoT *_moe_gen norms,
 * qwen2_navit.py:863-865,893-894,1166-1168).  row_idx optional gather/scatter. */
int umv_rmsnorm_bf16(const uint16_t* x, const uint16_t* w, const uint16_t* w_gen, const int32_t* expert,
                     uint16_t* out, int T, int H, float eps, umv_stream_t stream);
/* Consumer of a split-K decode GEMM (o_proj / down_proj, qwen2_navit.py:617-620, modeling_qwen2.py:235) fused with the
 * residual add (qwen2_navit.py:873-874,897-898) and the following Qwen2RMSNorm:
 *   seq[t,:] = bf16(bf16(sum_s partials[s*split_stride + t*ldp + :]) + seq[t,:])  (in place);  out = w * bf16(seq * rstd) */
int umv_residual_rmsnorm_bf16(const float* partials, int n_splits, int64_t split_stride, int64_t ldp, uint16_t* seq,
                              const uint16_t* w, uint16_t* out, int T, int H, float eps, umv_stream_t stream);
/* nn.LayerNorm with affine, bf16 in/out, fp32 statistics (siglip_navit.py:283,296,370) */
int umv_layernorm_bf16(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* out, int T, int H,
                       float eps, umv_stream_t stream);
/* nn.Embedding gather (bagel.py:438,577,753,1092,1264): out[t,:] = table[ids[t],:]; optional
 * out_rows scatter (packed_sequence[packed_text_indexes] = ..., bagel.py:579) */
int umv_embed_gather_bf16(const uint16_t* table, const int64_t* ids, const int32_t* out_rows, uint16_t* out,
                          int T, int H, umv_stream_t stream);
/* out[rows[t],:] = bf16(a[t,:] + table[idx[t],:]) (+ optional second addend broadcast row `bcast`,
 * added first): connector + vit_pos_embed (bagel.py:590-595), vae2llm + t_emb + pos (bagel.py:1104-1109) */
int umv_add_rows_bf16(const uint16_t* a, const uint16_t* bcast, const uint16_t* table, const int64_t* idx,
                      const int32_t* out_rows, uint16_t* out, int T, int H, umv_stream_t stream);
/* greedy sampling (bagel.py:1301): argmax over bf16 logits, lowest index wins ties */
int umv_argmax_bf16(const uint16_t* logits, int64_t ld, int64_t* out_ids, int M, int V, umv_stream_t stream);
/* do_sample=True (bagel.py:1297-1299): token = multinomial(softmax(logits / temperature), 1), drawn as
 * argmax(p_i / q_i), q_i ~ Exp(1), from a counter-based generator keyed by (seed, *step, row, i); `step`
 * is an optional device counter so the call replays from a hipGraph.  Reproducible per seed; not torch's stream. */
int umv_sample_bf16(const uint16_t* logits, int64_t ld, int64_t* out_ids, int M, int V, float temperature, uint64_t seed,
                    const int64_t* step, umv_stream_t stream);
/* pixels [N, 3*p*p] fp32 -> bf16 [N, Kp] zero padded (cast that autocast applies before
 * the patch-embed linear, siglip_navit.py:190) */
int umv_cast_pad_f32_bf16(const float* x, int64_t ldx, uint16_t* out, int64_t ldo, int T, int K, int Kp,
                          umv_stream_t stream);
/* patchify (data/data_utils.py:43-50) + that cast on the device: transformed image [C, H, W] fp32 -> tokens
 * [(H/p)*(W/p), Kp] bf16, column (pp*p + qq)*C + c of token (ph, pw) = image[c][ph*p + pp][pw*p + qq], zero padded to Kp.
 * Replaces the host-side permute of prepare_vit_images (bagel.py:540-548) on the engine's own path. */
int umv_patchify_f32_bf16(const float* img, int C, int H, int W, int p, uint16_t* out, int64_t ldo, int Kp, umv_stream_t stream);

/* ------------------------------------------------------------------ attention
 * KV slab layout (replaces NaiveCache's re-merged [sum K, kvh, hd] tensors,
 * qwen2_navit.py:207-221,585-600): per layer
 *   K  [seg][kv_head][cap][hd]   V^T [seg][kv_head][hd][cap]    (cap % 32 == 0)
 * V is kept transposed so that P.V consumes 16-byte key-contiguous MFMA fragments. */

/* per-head q/k RMSNorm + RoPE + cast + in-place KV append (qwen2_navit.py:544-545,568-583,
 * 585-600,622-624; modeling_qwen2.py:196-220).  qkv [T, (nq+2*nkv)*hd] from the fused QKV GEMM.
 * tok_seg/tok_slot/tok_pos: [T] int32 slab segment, slot in slab, rope position.
 * expert[t]!=0 -> the *_moe_gen norm weights; fp32_chain!=0 -> the whole call follows the
 * "gen" rounding chain (norm + rope in fp32, one final cast), as mode=="gen" does for text
 * and latent tokens alike.
 * cos/sin tables [max_pos, hd] bf16 as Qwen2RotaryEmbedding returns them (modeling_qwen2.py:184). */
typedef struct {
    const uint16_t* qkv;
    uint16_t* q_out; /* [T, nq, hd] bf16 */
    uint16_t* k_slab;
    uint16_t* vt_slab;
    int64_t k_seg_stride, k_head_stride, v_seg_stride, v_head_stride, v_d_stride;
    const int32_t *tok_seg, *tok_slot, *tok_pos, *expert;
    const uint16_t *q_norm_w, *k_norm_w, *q_norm_w_gen, *k_norm_w_gen; /* NULL norm -> no norm/rope (ViT) */
    const uint16_t *cos_tab, *sin_tab;
    int T, nq, nkv, hd;
    float eps;
    int fp32_chain;
    /* optional: take the fused QKV row from a split-K GEMM (umv_gemm_args.k_splits) instead of `qkv`:
     * x = bf16(sum_s qkv_partials[s*split_stride + t*(nq+2nkv)*hd + col] + qkv_bias[col]), splits in order 0..S-1 */
    const float* qkv_partials;
    int n_splits;
    int64_t split_stride;
    const uint16_t* qkv_bias;
    /* Paged KV (optional): page_table as in umv_attn_args - token t goes to page page_table[tok_seg[t] * page_table_stride + tok_slot[t] / 256],
     * in-page slot tok_slot[t] % 256; k_slab / vt_slab are the pools, *_seg_stride the page strides, v_d_stride = 256 */
    const int32_t* page_table;
    int page_table_stride;
} umv_qkv_post_args;
int umv_qkv_post(const umv_qkv_post_args* a, umv_stream_t stream);

/* flash_attn_varlen_func(q,k,v,cu_seqlens_q,cu_seqlens_k,max_q,max_k,causal) as used at
 * qwen2_navit.py:605-614 and siglip_navit.py:232-241: softmax scale 1/sqrt(hd), fp32
 * softmax, causal = bottom-right aligned.  Keys come from the slabs; kv_len[s] counts
 * the keys visible to segment s (including this call's new tokens).  nsplit>1 splits
 * the key range (decode) and needs workspace of umv_attn_workspace_bytes(). */
typedef struct {
    const uint16_t* q; /* [T, nq, hd] */
    uint16_t* out;     /* [T, nq, hd] */
    const int32_t* cu_q;
    const int32_t* kv_len;
    const uint16_t* k_slab;
    const uint16_t* vt_slab;
    int64_t k_seg_stride, k_head_stride, v_seg_stride, v_head_stride, v_d_stride;
    int nseg, nq, nkv, hd, causal;
    int max_q;  /* upper bound of query rows per segment (grid sizing) */
    int max_kv; /* upper bound of kv_len (grid sizing for nsplit) */
    int nsplit;
    void* workspace;
    /* Optional in-place operands (0 = the defaults above).  q_row_stride: elements between consecutive query rows (default
     * nq*hd; the q columns of a fused [T, (nq+2nkv)*hd] QKV buffer are read where the GEMM left them).  k_key_stride > 0: K is
     * NOT a slab but a packed [T, ...] buffer like q - key p of segment s is row cu_q[s] + p, k_slab points at its first K
     * column, k_head_stride is the head pitch inside a row, k_seg_stride is ignored (self-attention without a cache: the
     * SigLIP tower, siglip_navit.py:232-241).  V^T always comes from vt_slab. */
    int64_t q_row_stride, k_key_stride;
    /* Optional, for tests and A/B runs (0 / NULL in production).  variant: 0 = the library's own shape -> kernel policy; with
     * UMV_ATTN_VARIANT_FORCE set, the other UMV_ATTN_VARIANT_* bits pick the nsplit = 1 kernel for THIS call (a per-call value: the
     * library keeps no mutable state).  stats: two device counters the lazy-softmax kernels add to - [0] the (wave, q-tile, 32-key block)
     * events in which the softmax reference moved on a NON-EMPTY accumulator (O and l rescaled by alpha), [1] those in which a row's
     * reference was set for the first time. */
    int variant;
    uint32_t* stats;
    /* Paged KV (optional): page_table [nseg][page_table_stride] int32 - entry p of segment s is the pool page holding its keys
     * p*256 .. p*256+255.  k_slab / vt_slab are then page POOLS, K [page][kv_head][256][hd], V^T [page][kv_head][hd][256]: k_seg_stride /
     * v_seg_stride = elements per page, v_d_stride = 256; a pool is at most 2 GiB (32-bit offsets of the LDS-shared prefill kernels, whose
     * PAGED instantiations give the slab call's bits; hd 72 / exact-maximum variants of a paged call run on the per-wave kernel).  wave_split = 2 / 4 (decode, max_q rows in one q-tile, hd 128): the waves of a workgroup split its key
     * range and merge in LDS, so the same parallelism needs nsplit / wave_split partials for the combine. */
    const int32_t* page_table;
    int page_table_stride;
    int wave_split;
} umv_attn_args;
#define UMV_KV_PAGE 256
#define UMV_KV_PAGE_LOG2 8
enum {
    UMV_ATTN_VARIANT_FORCE = 1,        /* the bits below replace the process policy (and its UMV_ATTN_* environment knobs) */
    UMV_ATTN_VARIANT_STREAM = 2,       /* the per-wave streaming kernel (attn_kernel) even where the LDS-shared kernels would run */
    UMV_ATTN_VARIANT_TQ1 = 4,          /* LDS-shared kernel, one q-tile per wave */
    UMV_ATTN_VARIANT_TQ2 = 8,          /* ... two q-tiles per wave (neither bit: by grid size) */
    UMV_ATTN_VARIANT_EXACT = 16,       /* exact running maximum (the bits of attn_kernel) instead of the lazy softmax reference */
    UMV_ATTN_VARIANT_WHOLE_TOKENS = 32,/* q-tiles of whole tokens instead of densely packed (token, head) pairs */
    UMV_ATTN_VARIANT_PAIR = 64         /* debug builds (UMV_ATTN_PAIR_DEBUG) only: both q-tiles of a wave in one softmax call - a form that is
                                          NOT deterministic (csrc/attention_prefill.hip); ignored by the product build */
};
size_t umv_attn_workspace_bytes(int nseg, int nq, int hd, int max_q, int nsplit);
int umv_attn_varlen(const umv_attn_args* a, umv_stream_t stream);
/* Host-only query: which kernel an nsplit = 1 call goes to: 0 = per-wave streaming kernel, 1 / 2 = the LDS-shared
 * prefill kernel with 1 / 2 query tiles per wave (2 needs >= 512 workgroups). */
int umv_attn_prefill_tq(int nseg, int nq, int nkv, int hd, int max_q);


/* decode bookkeeping kept on device so a whole step replays from a hipGraph:
 * slot[b]+=1, pos[b]+=1, kv_len[b]+=1 (the .tolist() bookkeeping of bagel.py:1266-1275,1303-1310) */
int umv_decode_advance(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, int B, umv_stream_t stream);
/* The same plus the token log of the loop, one launch at the end of a step: with s = step_idx[0],
 * pred_ids[s][b] = ids[b] (the curr_tokens appended at bagel.py:1311-1312), in_ids[s+1][b] = ids[b] (what the next step is fed;
 * in_ids[0] holds the start tokens, bagel.py:1263), counters += 1, step_idx[0] = s + 1.  in_ids / pred_ids are [max_len][B]. */
int umv_decode_step_end(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, const int64_t* ids, int64_t* in_ids,
                        int64_t* pred_ids, int64_t* step_idx, int B, int max_len, umv_stream_t stream);
/* umv_decode_step_end with the greedy pick folded in: ids[b] = column of the maximum key of argmax_partial[b][0..n_tiles)
 * (written by umv_gemm_bf16 / umv_gemm_fp8w with argmax_partial set), then the bookkeeping above.  `ids` is an OUTPUT here
 * (the token the next step embeds).  One workgroup per sample, and one step counter PER SAMPLE: step_idx has B entries (all
 * equal; the caller zeroes them), workgroup b reads and advances step_idx[b] - no word is shared between workgroups. */
int umv_decode_step_end_argmax(int32_t* tok_slot, int32_t* tok_pos, int32_t* kv_len, const uint64_t* argmax_partial, int n_tiles,
                               int64_t* ids, int64_t* in_ids, int64_t* pred_ids, int64_t* step_idx, int B,
                               int max_len, umv_stream_t stream);


/* TimestepEmbedder.timestep_embedding (modeling_utils.py:87-109): out[r] = bf16(cat(cos(t[r] * freqs), sin(t[r] * freqs))),
 * out [n, 2*half]; freqs [half] fp32 = exp(-ln(10000) * i / half) from the caller.  The two linears + SiLU of the embedder
 * are umv_gemm_bf16 calls (UMV_EPI_SILU). */
int umv_timestep_embed(const float* t, const float* freqs, uint16_t* out, int n, int half, umv_stream_t stream);

/* ------------------------------------------------------------------ image head
 * CFG combine + renorm + Euler update (bagel.py:1173-1207, :983), bf16 rounding after each
 * op as the reference's bf16 tensors imply; x_t [N,D] fp32 updated in place.  v_* are
 * [T, D] bf16 (row stride ldv); rows[n] picks the latent rows; seg_off [nseg+1] delimits the
 * samples in n (norms are per sample).  renorm_type 0 global, 1 channel, 2 text_channel. */
int umv_cfg_renorm_euler(float* x_t, const uint16_t* v_t, const uint16_t* v_text, const uint16_t* v_img, int64_t ldv,
                         const int32_t* rows, const int32_t* seg_off, int nseg, float cfg_text_scale,
                         float cfg_img_scale, float renorm_min, int renorm_type, float dt, int D, umv_stream_t stream);

/* ------------------------------------------------------------------ VAE (FLUX autoencoder, autoencoder.py)
 * Activations are NHWC bf16 ([B,H,W,C], C % 8 == 0).  Convolution weights [Cout,Cin,k,k] are
 * permuted to [Cout,(ky,kx,ci)] and packed with umv_pack_weight_bf16 (N=Cout, K=k*k*Cin).
 * umv_conv2d_nhwc_bf16 = F.conv2d (+bias, + optional residual x + h of ResnetBlock/AttnBlock,
 * autoencoder.py:95,65) as an MFMA implicit GEMM:
 *   mode 0: stride 1, pad (k-1)/2            (autoencoder.py:76,78,80,138,167,214,238)
 *   mode 1: nearest 2x upsample then 3x3     (Upsample, autoencoder.py:116-118)
 *   mode 2: F.pad(0,1,0,1) then 3x3 stride 2 (Downsample, autoencoder.py:104-107) */
int umv_conv2d_nhwc_bf16(const uint16_t* x, const uint16_t* wp, const uint16_t* bias, const uint16_t* residual,
                         uint16_t* out, int B, int Cin, int Hin, int Win, int Cout, int ksize, int mode,
                         umv_stream_t stream);
/* nn.GroupNorm(32, C, eps, affine) on bf16 with optional swish x*sigmoid(x) (autoencoder.py:34-35,
 * 43,75,77,166,237); deterministic two-level reduction through `workspace`. */
size_t umv_groupnorm_workspace_bytes(int B, int HW);
int umv_groupnorm_nhwc_bf16(const uint16_t* x, const uint16_t* gamma, const uint16_t* beta, uint16_t* out, void* workspace,
                            int B, int HW, int C, float eps, int swish, umv_stream_t stream);
/* [B,C,H,W] fp32 -> NHWC bf16 with channels zero padded to Cp (input of encoder.conv_in) */
int umv_nchw_f32_to_nhwc_bf16(const float* x, uint16_t* out, int B, int C, int H, int W, int Cp, umv_stream_t stream);
/* latent tokens [h*w, p*p*c] fp32 -> NHWC bf16 [h*p, w*p, c] with z/scale + shift
 * (inferencer.py:239-241 unpatchify + autoencoder.py:306) */
int umv_unpatchify_latent(const float* tokens, uint16_t* out, int h, int w, int p, int c, float scale, float shift,
                          umv_stream_t stream);
/* decoder output NHWC bf16 (first 3 of Cs channels) -> uint8 HWC: (x*0.5+0.5).clamp(0,1)*255,
 * truncating cast (inferencer.py:253-254) */
int umv_pixels_to_u8(const uint16_t* x, uint8_t* out, int64_t npix, int Cs, umv_stream_t stream);
/* AttnBlock.attention (autoencoder.py:50-62: one head of C channels over H*W positions) as two umv_gemm_bf16 calls with fp32
 * outputs, S = Q K^T and O = P V, and these two row kernels between / after them (unimedvl_amd/vae.py::attnblock):
 *   umv_softmax_rows_f32:  P[r][c] = bf16(exp((S[r][c] - max_c S[r][c]) * scale)), l[r] = sum_c of the unrounded weights
 *                          (n even, <= 16384 columns; S [rows, ld_s] fp32, P [rows, ld_p] bf16)
 *   umv_rowscale_f32_bf16: out[r][c] = bf16(O[r][c] / l[r])
 * - the arithmetic of flash attention (fp32 scores and sums, bf16 weights, one division) with the row's true maximum. */
int umv_softmax_rows_f32(const float* S, int64_t ld_s, uint16_t* P, int64_t ld_p, float* l, int rows, int n, float scale,
                         umv_stream_t stream);
int umv_rowscale_f32_bf16(const float* O, int64_t ld_in, const float* l, uint16_t* out, int64_t ld_out, int rows, int C,
                          umv_stream_t stream);
/* encoder tail for sample b: moments NHWC [B,Hm,Wm,2z] + noise NCHW bf16 [B,z,Hm,Wm] ->
 * scale*(mean + exp(0.5*logvar)*noise - shift) (autoencoder.py:266-272,300-303), 2x2-patchified
 * tokens [h*w, p*p*z] of the top-left window (bagel.py:771-775) */
int umv_latent_sample_patchify(const uint16_t* moments, const uint16_t* noise, uint16_t* tokens, int b, int Hm, int Wm,
                               int z, int h, int w, int p, float scale, float shift, umv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
