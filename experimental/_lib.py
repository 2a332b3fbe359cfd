"""ctypes binding of libunimedvl_hip_experimental.so (experimental/include/unimedvl_hip_experimental.h)."""
import ctypes as C
import os

from unimedvl_amd import _lib as _prod
from unimedvl_amd._lib import AttnArgs, GemmArgs, UmvError

HERE = os.path.dirname(os.path.abspath(__file__))
EXP_LIB_PATH = os.environ.get("UMV_EXP_LIB_PATH") or os.path.join(HERE, "lib", "libunimedvl_hip_experimental.so")


class DecodeLayout(C.Structure):
    _fields_ = [("G", C.c_int), ("C", C.c_int), ("th", C.c_int), ("tpw", C.c_int)]


class AttnDecodeArgs(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("ld_qkv", C.c_int64), ("out", C.c_void_p), ("cu_q", C.c_void_p), ("kv_len", C.c_void_p),
        ("tok_pos", C.c_void_p), ("q_norm_w", C.c_void_p), ("k_norm_w", C.c_void_p), ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p),
        ("k_slab", C.c_void_p), ("vt_slab", C.c_void_p),
        ("k_seg_stride", C.c_int64), ("k_head_stride", C.c_int64), ("v_seg_stride", C.c_int64),
        ("v_head_stride", C.c_int64), ("v_d_stride", C.c_int64),
        ("nseg", C.c_int), ("nq", C.c_int), ("nkv", C.c_int), ("hd", C.c_int), ("eps", C.c_float), ("nsplit", C.c_int),
        ("workspace", C.c_void_p),
        ("qkv_partials", C.c_void_p), ("n_splits", C.c_int), ("split_stride", C.c_int64), ("qkv_bias", C.c_void_p),
    ]


class DeOp(C.Structure):
    """umv_de_op (experimental/include/unimedvl_hip_experimental.h): one op of a decode-engine chain"""
    _fields_ = [
        ("w", C.c_void_p), ("x", C.c_void_p), ("ldx", C.c_int64), ("norm_w", C.c_void_p), ("norm_eps", C.c_float),
        ("kind", C.c_int32), ("bias", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int64), ("out", C.c_void_p),
        ("ldo", C.c_int64), ("split_stride", C.c_int64), ("wait_cnt", C.c_void_p), ("sig_cnt", C.c_void_p),
        ("ss_out", C.c_void_p), ("ss_in", C.c_void_p), ("wait_target", C.c_uint32), ("wait_mode", C.c_int32), ("sig_mode", C.c_int32), ("sig_div", C.c_int32),
        ("KT", C.c_int32), ("ntiles", C.c_int32), ("pair", C.c_int32), ("kgroups", C.c_int32), ("rot", C.c_int32),
        ("epi", C.c_int32), ("publish", C.c_int32), ("ss_n", C.c_int32),
    ]


_EXP_SIGS = {
    "umv_exp_last_error": (C.c_char_p, []),
    "umv_decode_engine_counter_words": (C.c_size_t, []),
    "umv_decode_engine": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "umv_decode_engine_traced": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p]),
    "umv_attn_decode_fused": (C.c_int, [C.POINTER(AttnDecodeArgs), C.c_void_p]),
    "umv_attn_prefill32": (C.c_int, [C.POINTER(AttnArgs), C.c_void_p]),
    "umv_decode_layout_for": (C.c_int, [C.c_int, C.c_int, C.POINTER(DecodeLayout)]),
    "umv_decode_image_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.POINTER(DecodeLayout)]),
    "umv_repack_weight_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(DecodeLayout), C.c_void_p]),
    "umv_gemm_decode": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(DecodeLayout), C.c_int, C.c_void_p]),
    "umv_prefetch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
}

_exp_lib = None


def available():
    return os.path.exists(EXP_LIB_PATH)


def load():
    """The experimental library; raises if it has not been built (python -m experimental.build)."""
    global _exp_lib
    if _exp_lib is not None:
        return _exp_lib
    _prod.load()      # torch's HIP runtime first, as for the product library
    if not os.path.exists(EXP_LIB_PATH):
        raise UmvError(f"{EXP_LIB_PATH} not found: build it with `python -m experimental.build`")
    lib = C.CDLL(EXP_LIB_PATH)
    for name, (res, args) in _EXP_SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _exp_lib = lib
    return lib


def declared_symbols():
    return list(_EXP_SIGS.keys())


def check(rc, what):
    if rc != 0:
        msg = load().umv_exp_last_error().decode()
        raise UmvError(f"{what} failed (rc={rc}): {msg}")
