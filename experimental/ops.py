"""Host wrappers of the experimental entry points (same conventions as unimedvl_amd/ops.py)."""
import ctypes as C

import torch

from unimedvl_amd import ops as _ops
from unimedvl_amd._lib import EPI_BIAS, EPI_RESIDUAL, EPI_SWIGLU, GemmArgs, UmvError
from unimedvl_amd.ops import BF16, _p, _req, _stream

from . import _lib


class DecodeLinear:
    """Decode image of a PackedLinear (include/unimedvl_hip.h "decode GEMM"): one contiguous slab of th-row tiles per
    CU for the persistent M <= 16 weight-streaming kernel.  Built once per weight; e4m3 when the source has an fp8 image."""

    __slots__ = ("wd", "scale", "bias", "N", "K", "swiglu", "fp8", "layout")

    def __init__(self, lin, n_cus=None):
        lib = _lib.load()
        dev = (lin.w8 if lin.w8 is not None else lin.wp).device
        if n_cus is None:
            n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        if lin.th != 16:
            raise UmvError("DecodeLinear needs the standard 16-row image")
        self.N, self.K, self.swiglu, self.bias = lin.N, lin.K, lin.swiglu, lin.bias
        self.fp8 = lin.w8 is not None
        rows = lin.N // 2 if lin.swiglu else lin.N
        self.layout = _lib.DecodeLayout()
        _lib.check(lib.umv_decode_layout_for(rows, n_cus, C.byref(self.layout)), "umv_decode_layout_for")
        nbytes = lib.umv_decode_image_bytes(lin.K, int(lin.swiglu), int(self.fp8), C.byref(self.layout))
        self.wd = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L = self.layout
        self.scale = torch.empty(L.G * L.tpw * (2 if lin.swiglu else 1) * 16, dtype=torch.float32, device=dev) if self.fp8 else None
        src = lin.w8 if self.fp8 else lin.wp
        _lib.check(lib.umv_repack_weight_decode(_p(src), _p(lin.scale) if self.fp8 else None, _p(self.wd), _p(self.scale), rows, lin.K,
                                           int(lin.swiglu), int(self.fp8), C.byref(self.layout), _stream()), "umv_repack_weight_decode")

    def nbytes(self):
        return self.wd.numel()


def gemm_decode(x, dlin, out=None, *, M=None, residual=None, row_idx=None, use_bias=True, norm_w=None, norm_eps=1e-6):
    """out = epilogue(x @ W^T) for M <= 16 rows from a DecodeLinear; norm_w fuses Qwen2RMSNorm(x) * norm_w (K <= 4096).
    Bit-identical to gemm() on the same weight."""
    lib = _lib.load()
    _req(x, BF16, "x")
    assert x.stride(-1) == 1
    M = x.shape[0] if M is None else M
    flags = 0
    if dlin.bias is not None and use_bias:
        flags |= EPI_BIAS
    if dlin.swiglu:
        flags |= EPI_SWIGLU
    if residual is not None:
        flags |= EPI_RESIDUAL
    n_out = dlin.N // 2 if dlin.swiglu else dlin.N
    if out is None:
        assert row_idx is None, "row-indexed GEMM writes into a caller-provided buffer"
        out = torch.empty((x.shape[0], n_out), dtype=BF16, device=x.device)
    a = GemmArgs(
        x=x.data_ptr(), ldx=x.stride(0), wp=dlin.wd.data_ptr(),
        bias=dlin.bias.data_ptr() if (flags & EPI_BIAS) else None,
        residual=residual.data_ptr() if residual is not None else None,
        ldr=residual.stride(0) if residual is not None else 0,
        out=out.data_ptr(), ldo=out.stride(0),
        row_idx=row_idx.data_ptr() if row_idx is not None else None,
        M=M, N=dlin.N, K=dlin.K, epilogue=flags,
        norm_w=norm_w.data_ptr() if norm_w is not None else None, norm_eps=norm_eps, tile_rows=0,
        w_scale=dlin.scale.data_ptr() if dlin.fp8 else None)
    _lib.check(lib.umv_gemm_decode(C.byref(a), C.byref(dlin.layout), int(dlin.fp8), _stream()), "umv_gemm_decode")
    return out


def attn_decode_fused(qkv, out, slab, cu_q, kv_len, tok_pos, nq, nkv, hd, eps, q_norm, k_norm, cos_tab, sin_tab, nsplit=1,
                      workspace=None, partials=None, bias=None):
    """One decode step: q/k norm + RoPE + KV append + attention over kv_len keys, from the raw fused QKV rows (`qkv` bf16)
    or from the fp32 partial sums [n_splits, B, (nq+2nkv)*hd] of a split-K QKV GEMM (`partials`, + `bias`)."""
    lib = _lib.load()
    extra = {}
    if partials is not None:
        _req(partials, torch.float32, "partials")
        if partials.dim() != 3 or partials.stride(2) != 1:
            raise UmvError("attn_decode_fused: partials must be [n_splits, B, (nq+2nkv)*hd] with unit column stride")
        extra = dict(qkv=None, ld_qkv=partials.stride(1), qkv_partials=partials.data_ptr(), n_splits=partials.shape[0],
                     split_stride=partials.stride(0), qkv_bias=None if bias is None else bias.data_ptr())
    else:
        _req(qkv, BF16, "qkv")
        extra = dict(qkv=qkv.data_ptr(), ld_qkv=qkv.stride(0))
    a = _lib.AttnDecodeArgs(
        out=out.data_ptr(), cu_q=cu_q.data_ptr(), kv_len=kv_len.data_ptr(),
        tok_pos=tok_pos.data_ptr(), q_norm_w=q_norm.data_ptr(), k_norm_w=k_norm.data_ptr(), cos_tab=cos_tab.data_ptr(),
        sin_tab=sin_tab.data_ptr(), k_slab=slab.k.data_ptr(), vt_slab=slab.vt.data_ptr(), nseg=kv_len.numel(), nq=nq, nkv=nkv, hd=hd,
        eps=eps, nsplit=nsplit, workspace=None if workspace is None else workspace.data_ptr(), **extra, **slab.strides())
    _lib.check(lib.umv_attn_decode_fused(C.byref(a), _stream()), "umv_attn_decode_fused")
    return out


def prefetch(t, nbytes=None, offset=0, blocks=128, stream=None):
    """Pull `nbytes` of tensor `t` (from byte `offset`) into L2 / Infinity Cache on `stream` (default: current)."""
    lib = _lib.load()
    total = t.numel() * t.element_size()
    nbytes = total - offset if nbytes is None else min(nbytes, total - offset)
    if nbytes <= 0:
        return
    st = _stream() if stream is None else C.c_void_p(stream.cuda_stream)
    _lib.check(lib.umv_prefetch(C.c_void_p(t.data_ptr() + offset), nbytes, blocks, None, st), "umv_prefetch")


def attn_prefill32(q, out, slab, cu_q, kv_len, nq, nkv, hd, causal, max_q, max_kv, k_packed=None):
    """umv_attn_prefill32: the nsplit == 1, hd 128 shapes of umv_attn_varlen on the 32x32x16 matrix instruction."""
    lib = _lib.load()
    return _ops.attention(q, out, slab, cu_q, kv_len, nq, nkv, hd, causal, max_q, max_kv, k_packed=k_packed,
                          _entry=(lib.umv_attn_prefill32, _lib.check, "umv_attn_prefill32"))
