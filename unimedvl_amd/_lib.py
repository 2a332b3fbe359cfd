"""ctypes binding of libunimedvl_hip.so (include/unimedvl_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol
is absent this module raises, and every op raises on a non-zero return code.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UMV_LIB_PATH") or os.path.join(HERE, "lib", "libunimedvl_hip.so")   # (override: A/B builds, tuning only)

EPI_BIAS, EPI_GELU_TANH, EPI_SILU, EPI_RESIDUAL, EPI_SWIGLU, EPI_OUT_F32 = 1, 2, 4, 8, 16, 32


class GemmArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64), ("wp", C.c_void_p), ("bias", C.c_void_p),
        ("residual", C.c_void_p), ("ldr", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
        ("row_idx", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("epilogue", C.c_int),
        ("norm_w", C.c_void_p), ("norm_eps", C.c_float), ("tile_rows", C.c_int), ("w_scale", C.c_void_p),
        ("k_splits", C.c_int), ("split_stride", C.c_int64), ("argmax_partial", C.c_void_p), ("x_rows", C.c_int64),
        ("sample_temperature", C.c_float), ("sample_seed", C.c_uint64), ("sample_step", C.c_void_p),
    ]


class QkvPostArgs(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("q_out", C.c_void_p), ("k_slab", C.c_void_p), ("vt_slab", C.c_void_p),
        ("k_seg_stride", C.c_int64), ("k_head_stride", C.c_int64), ("v_seg_stride", C.c_int64),
        ("v_head_stride", C.c_int64), ("v_d_stride", C.c_int64),
        ("tok_seg", C.c_void_p), ("tok_slot", C.c_void_p), ("tok_pos", C.c_void_p), ("expert", C.c_void_p),
        ("q_norm_w", C.c_void_p), ("k_norm_w", C.c_void_p), ("q_norm_w_gen", C.c_void_p), ("k_norm_w_gen", C.c_void_p),
        ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p),
        ("T", C.c_int), ("nq", C.c_int), ("nkv", C.c_int), ("hd", C.c_int), ("eps", C.c_float),
        ("fp32_chain", C.c_int),
        ("qkv_partials", C.c_void_p), ("n_splits", C.c_int), ("split_stride", C.c_int64), ("qkv_bias", C.c_void_p),
        ("page_table", C.c_void_p), ("page_table_stride", C.c_int),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("out", C.c_void_p), ("cu_q", C.c_void_p), ("kv_len", C.c_void_p),
        ("k_slab", C.c_void_p), ("vt_slab", C.c_void_p),
        ("k_seg_stride", C.c_int64), ("k_head_stride", C.c_int64), ("v_seg_stride", C.c_int64),
        ("v_head_stride", C.c_int64), ("v_d_stride", C.c_int64),
        ("nseg", C.c_int), ("nq", C.c_int), ("nkv", C.c_int), ("hd", C.c_int), ("causal", C.c_int),
        ("max_q", C.c_int), ("max_kv", C.c_int), ("nsplit", C.c_int), ("workspace", C.c_void_p),
        ("q_row_stride", C.c_int64), ("k_key_stride", C.c_int64),
        ("variant", C.c_int), ("stats", C.c_void_p),
        ("page_table", C.c_void_p), ("page_table_stride", C.c_int), ("wave_split", C.c_int),
    ]


# umv_attn_args.variant bits (tests / A-B only): FORCE makes the others replace the library's shape -> kernel policy for one call
ATTN_FORCE, ATTN_STREAM, ATTN_TQ1, ATTN_TQ2, ATTN_EXACT, ATTN_WHOLE_TOKENS, ATTN_PAIR = 1, 2, 4, 8, 16, 32, 64


class Gemm8Args(C.Structure):
    _fields_ = [
        ("xq", C.c_void_p), ("ldq", C.c_int64), ("x_scale", C.c_void_p), ("wp", C.c_void_p), ("w_scale", C.c_void_p),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldr", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
        ("row_idx", C.c_void_p), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("epilogue", C.c_int),
    ]


_SIGS = {
    "umv_packed_weight_fp8_mfma_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "umv_repack_weight_fp8_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_quantize_act_fp8": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_int, C.c_int, C.c_void_p]),
    "umv_gemm_fp8a8w": (C.c_int, [C.POINTER(Gemm8Args), C.c_void_p]),
    "umv_gemm_tile_config": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "umv_attn_prefill_tq": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "umv_version": (C.c_int, []),
    "umv_last_error": (C.c_char_p, []),
    "umv_packed_weight_elems": (C.c_size_t, [C.c_int, C.c_int]),
    "umv_pack_weight_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_repacked_weight_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "umv_repack_weight_rows_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "umv_pack_weight_swiglu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "umv_packed_weight_fp8_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "umv_quantize_pack_weight_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_int, C.c_void_p]),
    "umv_gemm_fp8w": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "umv_residual_rmsnorm_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                            C.c_int, C.c_float, C.c_void_p]),
    "umv_rmsnorm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_float, C.c_void_p]),
    "umv_layernorm_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                     C.c_void_p]),
    "umv_embed_gather_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_add_rows_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p]),
    "umv_argmax_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_sample_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_void_p,
                                  C.c_void_p]),
    "umv_cast_pad_f32_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p]),
    "umv_patchify_f32_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "umv_qkv_post": (C.c_int, [C.POINTER(QkvPostArgs), C.c_void_p]),
    "umv_attn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "umv_attn_varlen": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "umv_decode_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "umv_decode_step_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_void_p]),
    "umv_decode_step_end_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_timestep_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "umv_cfg_renorm_euler": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "umv_conv2d_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "umv_groupnorm_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "umv_groupnorm_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "umv_nchw_f32_to_nhwc_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p]),
    "umv_unpatchify_latent": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.c_float, C.c_void_p]),
    "umv_pixels_to_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "umv_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "umv_rowscale_f32_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "umv_latent_sample_patchify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]),
}

_lib = None


class UmvError(RuntimeError):
    pass


def load():
    """Load the library once; raises if it (or any declared symbol) is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); importing torch first makes
    # the dynamic loader bind our library to THAT runtime instance, so device pointers and
    # streams are shared.  Loading ours first would pull a second runtime from /opt/rocm.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise UmvError(
            f"{LIB_PATH} not found: build it with `python -m unimedvl_amd.build` "
            "(there is no CPU or PyTorch fallback for the product path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols():
    return list(_SIGS.keys())


def check(rc, what):
    if rc != 0:
        msg = load().umv_last_error().decode()
        raise UmvError(f"{what} failed (rc={rc}): {msg}")
