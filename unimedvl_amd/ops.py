"""Thin torch-tensor wrappers over the C ABI (torch is used for device memory and
streams only).  Every function launches a hand-written gfx950 kernel from
libunimedvl_hip.so on torch's current stream; there is no fallback path."""
import ctypes as C
import functools

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_GELU_TANH, EPI_OUT_F32, EPI_RESIDUAL, EPI_SILU, EPI_SWIGLU, AttnArgs, GemmArgs,
                   QkvPostArgs, check)

BF16 = torch.bfloat16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (every op launches there)."""
    if _raw_stream is not None:       # ~0.3 us instead of ~3 us for building a torch.cuda.Stream object per launch
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(fn):
    """Method decorator: run with `self.device` as torch's current device.  Every kernel launches on the CURRENT device's
    current stream (_stream), so an engine built on cuda:N (the scripts' target_gpu_device, interactive_vqa_inferencer.py:60)
    must make N current around its public entry points; a no-op when it already is."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = torch.device(self.device)
        if dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapper


def device_scope(device):
    """Context manager form of on_device (constructors, where self.device does not exist yet)."""
    dev = torch.device(device)
    if dev.type != "cuda" or dev.index is None:
        import contextlib
        return contextlib.nullcontext()
    return torch.cuda.device(dev)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if not t.is_cuda:
        raise _lib.UmvError(f"{name}: tensor must live on the GPU (no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.UmvError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


class PackedLinear:
    """nn.Linear weight [N,K] (+bias) re-tiled for the MFMA GEMM kernels."""

    __slots__ = ("wp", "bias", "N", "K", "swiglu", "th", "w8", "scale", "w8m")

    def __init__(self, wp, bias, N, K, swiglu=False, th=16, w8=None, scale=None):
        self.wp, self.bias, self.N, self.K, self.swiglu, self.th = wp, bias, N, K, swiglu, th
        # fp8 weights (BASELINE.json configs[4]): w8 = e4m3 image streamed by the decode GEMM (M <= 64), scale = its
        # power-of-two channel scales; wp is then the bf16 image of the SAME dequantised weights for M > 64
        self.w8, self.scale = w8, scale
        # optional: the e4m3 image re-tiled for the fp8 matrix instruction; when present, GEMMs with M > 64 rows quantise
        # their activations per row and run W8A8 (enable_fp8_mfma)
        self.w8m = None

    def enable_fp8_mfma(self, keep_bf16=False):
        """Build the fp8-MFMA image from the e4m3 image; the bf16 image of the dequantised weights is dropped unless asked."""
        if self.w8m is not None:        # already there (e.g. read back from the packed fast-path file)
            if not keep_bf16:
                self.wp = None
            return self
        if self.w8 is None:
            raise _lib.UmvError("enable_fp8_mfma needs fp8 weights (from_weight_fp8 / from_gate_up_fp8)")
        lib = _lib.load()
        self.w8m = torch.empty(lib.umv_packed_weight_fp8_mfma_bytes(self.N, self.K), dtype=torch.uint8, device=self.w8.device)
        check(lib.umv_repack_weight_fp8_mfma(_p(self.w8), _p(self.w8m), self.N, self.K, _stream()), "umv_repack_weight_fp8_mfma")
        if not keep_bf16:
            self.wp = None
        return self

    @staticmethod
    def from_weight_fp8(w, bias=None):
        """Quantise an nn.Linear weight to e4m3 with power-of-two channel scales; see include/unimedvl_hip.h."""
        lib = _lib.load()
        w = _req(w.contiguous(), BF16, "weight")
        N, K = w.shape
        w8 = torch.empty(lib.umv_packed_weight_fp8_bytes(N, K), dtype=torch.uint8, device=w.device)
        scale = torch.empty(((N + 15) // 16) * 16, dtype=torch.float32, device=w.device)
        deq = torch.empty_like(w)
        check(lib.umv_quantize_pack_weight_fp8(_p(w), None, _p(w8), _p(scale), _p(deq), None, N, K, _stream()),
              "umv_quantize_pack_weight_fp8")
        lin = PackedLinear.from_weight(deq, bias)
        lin.w8, lin.scale = w8, scale
        return lin

    @staticmethod
    def from_gate_up_fp8(gate, up):
        lib = _lib.load()
        gate = _req(gate.contiguous(), BF16, "gate")
        up = _req(up.contiguous(), BF16, "up")
        I, K = gate.shape
        assert I % 16 == 0, "intermediate size must be a multiple of 16"
        w8 = torch.empty(lib.umv_packed_weight_fp8_bytes(2 * I, K), dtype=torch.uint8, device=gate.device)
        scale = torch.empty(2 * I, dtype=torch.float32, device=gate.device)
        dg, du = torch.empty_like(gate), torch.empty_like(up)
        check(lib.umv_quantize_pack_weight_fp8(_p(gate), _p(up), _p(w8), _p(scale), _p(dg), _p(du), I, K, _stream()),
              "umv_quantize_pack_weight_fp8")
        lin = PackedLinear.from_gate_up(dg, du)
        lin.w8, lin.scale = w8, scale
        return lin

    def for_decode(self, n_cus=256):
        """A second, decode-only image with th-row tiles such that the number of tiles is a multiple of the
        CU count (exact partition of the weight stream over the chip); returns self when 16 is already fine."""
        if self.swiglu or self.th != 16 or self.w8 is not None:
            return self
        best = None
        for th in range(15, 7, -1):
            if self.N % th == 0 and (self.N // th) % n_cus == 0:
                best = th
                break
        if best is None or ((self.N + 15) // 16) % n_cus == 0:
            return self
        lib = _lib.load()
        out = torch.empty(lib.umv_repacked_weight_elems(self.N, self.K, best), dtype=BF16, device=self.wp.device)
        check(lib.umv_repack_weight_rows_bf16(_p(self.wp), _p(out), self.N, self.K, best, _stream()), "umv_repack_weight_rows_bf16")
        return PackedLinear(out, self.bias, self.N, self.K, False, best)

    @staticmethod
    def from_weight(w, bias=None):
        lib = _lib.load()
        w = _req(w.contiguous(), BF16, "weight")
        N, K = w.shape
        wp = torch.empty(lib.umv_packed_weight_elems(N, K), dtype=BF16, device=w.device)
        check(lib.umv_pack_weight_bf16(_p(w), _p(wp), N, K, _stream()), "umv_pack_weight_bf16")
        return PackedLinear(wp, None if bias is None else bias.contiguous(), N, K)

    @staticmethod
    def from_gate_up(gate, up):
        lib = _lib.load()
        gate = _req(gate.contiguous(), BF16, "gate")
        up = _req(up.contiguous(), BF16, "up")
        I, K = gate.shape
        assert I % 16 == 0, "intermediate size must be a multiple of 16"
        wp = torch.empty(lib.umv_packed_weight_elems(2 * I, K), dtype=BF16, device=gate.device)
        check(lib.umv_pack_weight_swiglu_bf16(_p(gate), _p(up), _p(wp), I, K, _stream()), "umv_pack_weight_swiglu_bf16")
        return PackedLinear(wp, None, 2 * I, K, swiglu=True)

    def nbytes(self):
        return self.wp.numel() * 2


def gemm(x, lin, out=None, *, M=None, residual=None, act=None, row_idx=None, out_f32=False, use_bias=True,
         norm_w=None, norm_eps=1e-6, act8=False, argmax_partial=None, sample=None):
    """out = epilogue(x @ W^T).  x [M,K] bf16 (row stride may exceed K).  act in {None,'gelu_tanh','silu'}.
    norm_w: fuse Qwen2RMSNorm(x)*norm_w into the GEMM prologue (M <= 16, K <= 4096).
    act8 (W8A8 mode, needs lin.w8m): the activations are rounded per row through e4m3 - on the fp8 matrix instruction for
    M > 64 rows, as a bf16 copy of the rounded rows for the weight-streaming kernels below that.
    argmax_partial (int64 [M, ceil(N/16)], M <= 64): greedy-argmax keys per 16-column tile, finished by decode_step_end_argmax.
    sample (with argmax_partial): (temperature, seed, step tensor or None) - the keys then order bf16(logit / T) + Gumbel noise, so the
    row maximum is one draw from softmax(logits / T) (bagel.py:1297-1299) instead of the greedy token."""
    lib = _lib.load()
    _req(x, BF16, "x")
    assert x.stride(-1) == 1
    M = x.shape[0] if M is None else M
    flags = 0
    if lin.bias is not None and use_bias:
        flags |= EPI_BIAS
    if act == "gelu_tanh":
        flags |= EPI_GELU_TANH
    elif act == "silu":
        flags |= EPI_SILU
    elif act is not None:
        raise ValueError(act)
    if lin.swiglu:
        flags |= EPI_SWIGLU
    if residual is not None:
        flags |= EPI_RESIDUAL
    if out_f32:
        flags |= EPI_OUT_F32
    n_out = lin.N // 2 if lin.swiglu else lin.N
    rows_out = x.shape[0] if row_idx is None else None
    if out is None:
        assert row_idx is None, "row-indexed GEMM writes into a caller-provided buffer"
        out = torch.empty((rows_out, n_out), dtype=torch.float32 if out_f32 else BF16, device=x.device)
    if act8 and lin.w8m is None:
        raise _lib.UmvError("act8 needs a linear with the fp8-MFMA image (enable_fp8_mfma)")
    if act8 and M <= 64:
        x = fake_quantize_act(x, M, row_idx)
    amax = None
    if argmax_partial is not None:
        _req(argmax_partial, torch.int64, "argmax_partial")
        if not (argmax_partial.is_contiguous() and tuple(argmax_partial.shape) == (M, (lin.N + 15) // 16)):
            raise _lib.UmvError(f"argmax_partial must be a contiguous int64 [{M}, {(lin.N + 15) // 16}] tensor")
        amax = argmax_partial.data_ptr()
    # which kernel family takes the call (mirrors the branches below exactly, so a dropped image raises instead of crashing)
    use_a8 = act8 and M > 64 and norm_w is None and not out_f32       # fp8 matrix instruction, e4m3 activations
    use_w8 = lin.w8 is not None and M <= 64 and norm_w is None        # weight-streaming kernel on the e4m3 image
    if lin.wp is None and not use_a8 and not use_w8:
        raise _lib.UmvError(f"this linear only has fp8 images (the bf16 image was dropped by enable_fp8_mfma): M={M} rows with "
                            f"act8={act8}, out_f32={out_f32}, norm_w={'set' if norm_w is not None else 'None'} need the bf16 kernel - "
                            "build the weights with enable_fp8_mfma(keep_bf16=True)")
    if use_a8:
        # W8A8: per-row e4m3 activations (rows gathered through row_idx), fp8 matrix instruction, exact pow2 scales
        ldq = (lin.K + 127) // 128 * 128
        xq = torch.empty((M, ldq), dtype=torch.uint8, device=x.device)
        xs = torch.empty((M,), dtype=torch.float32, device=x.device)
        check(lib.umv_quantize_act_fp8(_p(x), x.stride(0), _p(row_idx), _p(xq), ldq, _p(xs), None, 0, M, lin.K, _stream()),
              "umv_quantize_act_fp8")
        a8 = _lib.Gemm8Args(
            xq=xq.data_ptr(), ldq=ldq, x_scale=xs.data_ptr(), wp=lin.w8m.data_ptr(), w_scale=lin.scale.data_ptr(),
            bias=lin.bias.data_ptr() if (flags & EPI_BIAS) else None,
            residual=residual.data_ptr() if residual is not None else None,
            ldr=residual.stride(0) if residual is not None else 0, out=out.data_ptr(), ldo=out.stride(0),
            row_idx=row_idx.data_ptr() if row_idx is not None else None, M=M, N=lin.N, K=lin.K, epilogue=flags)
        check(lib.umv_gemm_fp8a8w(C.byref(a8), _stream()), "umv_gemm_fp8a8w")
        return out
    if use_w8:
        a = GemmArgs(
            x=x.data_ptr(), ldx=x.stride(0), wp=lin.w8.data_ptr(),
            bias=lin.bias.data_ptr() if (flags & EPI_BIAS) else None,
            residual=residual.data_ptr() if residual is not None else None,
            ldr=residual.stride(0) if residual is not None else 0,
            out=out.data_ptr(), ldo=out.stride(0),
            row_idx=row_idx.data_ptr() if row_idx is not None else None,
            M=M, N=lin.N, K=lin.K, epilogue=flags, norm_w=None, norm_eps=norm_eps, tile_rows=0,
            w_scale=lin.scale.data_ptr(), argmax_partial=amax, **_sample_fields(sample, amax))
        check(lib.umv_gemm_fp8w(C.byref(a), _stream()), "umv_gemm_fp8w")
        return out
    a = GemmArgs(
        x=x.data_ptr(), ldx=x.stride(0), wp=lin.wp.data_ptr(),
        bias=lin.bias.data_ptr() if (flags & EPI_BIAS) else None,
        residual=residual.data_ptr() if residual is not None else None,
        ldr=residual.stride(0) if residual is not None else 0,
        out=out.data_ptr(), ldo=out.stride(0),
        row_idx=row_idx.data_ptr() if row_idx is not None else None,
        M=M, N=lin.N, K=lin.K, epilogue=flags,
        norm_w=norm_w.data_ptr() if norm_w is not None else None, norm_eps=norm_eps, tile_rows=lin.th, argmax_partial=amax,
        x_rows=x.shape[0], **_sample_fields(sample, amax))
    check(lib.umv_gemm_bf16(C.byref(a), _stream()), "umv_gemm_bf16")
    return out


def rmsnorm(x, w, eps, out=None, w_gen=None, expert=None):
    lib = _lib.load()
    _req(x, BF16, "x")
    T, H = x.shape
    out = torch.empty_like(x) if out is None else out
    check(lib.umv_rmsnorm_bf16(_p(x), _p(w), _p(w_gen), _p(expert), _p(out), T, H, eps, _stream()), "umv_rmsnorm_bf16")
    return out


def layernorm(x, w, b, eps, out=None):
    lib = _lib.load()
    _req(x, BF16, "x")
    T, H = x.shape
    out = torch.empty_like(x) if out is None else out
    check(lib.umv_layernorm_bf16(_p(x), _p(w), _p(b), _p(out), T, H, eps, _stream()), "umv_layernorm_bf16")
    return out


def embed_gather(table, ids, out=None, out_rows=None):
    lib = _lib.load()
    _req(table, BF16, "table")
    _req(ids, torch.int64, "ids")
    T, H = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty((T, H), dtype=BF16, device=table.device)
    check(lib.umv_embed_gather_bf16(_p(table), _p(ids), _p(out_rows), _p(out), T, H, _stream()), "umv_embed_gather_bf16")
    return out


def add_rows(a, out, bcast=None, table=None, idx=None, out_rows=None):
    lib = _lib.load()
    _req(a, BF16, "a")
    T, H = a.shape
    check(lib.umv_add_rows_bf16(_p(a), _p(bcast), _p(table), _p(idx), _p(out_rows), _p(out), T, H, _stream()),
          "umv_add_rows_bf16")
    return out


def argmax(logits, out=None):
    lib = _lib.load()
    _req(logits, BF16, "logits")
    M, V = logits.shape
    out = torch.empty((M,), dtype=torch.int64, device=logits.device) if out is None else out
    check(lib.umv_argmax_bf16(_p(logits), logits.stride(0), _p(out), M, V, _stream()), "umv_argmax_bf16")
    return out


def sample(logits, temperature, seed, step=None, out=None):
    """multinomial(softmax(logits / temperature), 1) per row (device-side counter-based RNG)."""
    lib = _lib.load()
    _req(logits, BF16, "logits")
    M, V = logits.shape
    out = torch.empty((M,), dtype=torch.int64, device=logits.device) if out is None else out
    check(lib.umv_sample_bf16(_p(logits), logits.stride(0), _p(out), M, V, float(temperature), int(seed) & (2 ** 64 - 1),
                              _p(step), _stream()), "umv_sample_bf16")
    return out


def cast_pad(x, Kp):
    lib = _lib.load()
    _req(x, torch.float32, "x")
    T, K = x.shape
    out = torch.empty((T, Kp), dtype=BF16, device=x.device)
    check(lib.umv_cast_pad_f32_bf16(_p(x), x.stride(0), _p(out), Kp, T, K, Kp, _stream()), "umv_cast_pad_f32_bf16")
    return out


def patchify(img, out, patch):
    """Transformed image [C, H, W] fp32 (device) -> its patch tokens, cast to bf16 and zero padded, into out [(H/p)*(W/p), Kp]."""
    lib = _lib.load()
    _req(img, torch.float32, "img")
    _req(out, BF16, "out")
    if img.dim() != 3 or not img.is_contiguous() or out.dim() != 2 or out.stride(1) != 1:
        raise _lib.UmvError("patchify: img must be a contiguous [C, H, W] tensor, out a [tokens, Kp] view with unit column stride")
    C_, H, W = img.shape
    if out.shape[0] != (H // patch) * (W // patch):
        raise _lib.UmvError(f"patchify: out has {out.shape[0]} rows, the image has {(H // patch) * (W // patch)} patches")
    check(lib.umv_patchify_f32_bf16(_p(img), C_, H, W, patch, _p(out), out.stride(0), out.shape[1], _stream()), "umv_patchify_f32_bf16")
    return out


class KVSlab:
    """One layer's K / V^T slabs: K [seg][nkv][cap][hd], V^T [seg][nkv][hd][cap]."""

    __slots__ = ("k", "vt", "nseg", "nkv", "cap", "hd")

    def __init__(self, nseg, nkv, cap, hd, device, keys=True):
        """keys=False: V^T only - K is read in place from a packed buffer (attention(..., k_packed=...))."""
        assert cap % 32 == 0
        self.k = torch.zeros((nseg, nkv, cap, hd), dtype=BF16, device=device) if keys else None
        self.vt = torch.zeros((nseg, nkv, hd, cap), dtype=BF16, device=device)
        self.nseg, self.nkv, self.cap, self.hd = nseg, nkv, cap, hd

    @staticmethod
    def from_tensors(k, vt):
        """Wrap existing (possibly sliced along the segment dim) slab tensors."""
        s = KVSlab.__new__(KVSlab)
        s.k, s.vt = k, vt
        s.nseg, s.nkv, s.cap, s.hd = k.shape
        assert k.stride(3) == 1 and k.stride(2) == s.hd and k.stride(1) == s.cap * s.hd and k.stride(0) == s.nkv * s.cap * s.hd
        return s

    def strides(self):
        return dict(k_seg_stride=self.nkv * self.cap * self.hd, k_head_stride=self.cap * self.hd,
                    v_seg_stride=self.nkv * self.hd * self.cap, v_head_stride=self.hd * self.cap, v_d_stride=self.cap)


KV_PAGE = 256      # keys per page of a paged KV pool (UMV_KV_PAGE)


class PagedSlab:
    """One layer's K / V^T page POOLS of a paged cache (kvcache.PagedCache): K [page][nkv][256][hd], V^T [page][nkv][hd][256], shared
    by all segments through `table` ([nseg, max_pages] int32 on the device: entry p of segment s = the pool page of its keys p*256 ..).
    Stands in for a KVSlab in qkv_post() / attention(): same strides() contract with "segment" read as "page"."""

    __slots__ = ("k", "vt", "table", "nkv", "hd", "cap")

    def __init__(self, npages, nkv, hd, device, table):
        self.k = torch.zeros((npages, nkv, KV_PAGE, hd), dtype=BF16, device=device)
        self.vt = torch.zeros((npages, nkv, hd, KV_PAGE), dtype=BF16, device=device)
        self.table, self.nkv, self.hd = table, nkv, hd
        self.cap = table.shape[1] * KV_PAGE          # longest context the table can describe

    def strides(self):
        return dict(k_seg_stride=self.nkv * KV_PAGE * self.hd, k_head_stride=KV_PAGE * self.hd,
                    v_seg_stride=self.nkv * self.hd * KV_PAGE, v_head_stride=self.hd * KV_PAGE, v_d_stride=KV_PAGE)


def _paging(slab):
    t = getattr(slab, "table", None)
    return {} if t is None else dict(page_table=t.data_ptr(), page_table_stride=t.stride(0))


def quantize_act(x, M=None, row_idx=None):
    """Per-row e4m3 quantisation of bf16 activations (umv_quantize_act_fp8): returns (xq uint8 [M, ldq], scale f32 [M])."""
    lib = _lib.load()
    _req(x, BF16, "x")
    M = x.shape[0] if M is None else M
    K = x.shape[1]
    ldq = (K + 127) // 128 * 128
    xq = torch.empty((M, ldq), dtype=torch.uint8, device=x.device)
    xs = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(lib.umv_quantize_act_fp8(_p(x), x.stride(0), _p(row_idx), _p(xq), ldq, _p(xs), None, 0, M, K, _stream()), "umv_quantize_act_fp8")
    return xq, xs


def fake_quantize_act(x, M=None, row_idx=None):
    """bf16 copy of x whose rows (row_idx[m] when given, the others are left undefined) are rounded through the per-row e4m3
    grid: the activations of the W8A8 mode for the M <= 64 kernels, which take bf16 inputs."""
    lib = _lib.load()
    _req(x, BF16, "x")
    M = x.shape[0] if M is None else M
    out = torch.empty_like(x)
    xs = torch.empty((M,), dtype=torch.float32, device=x.device)
    ldq = (x.shape[1] + 127) // 128 * 128
    check(lib.umv_quantize_act_fp8(_p(x), x.stride(0), _p(row_idx), None, ldq, _p(xs), _p(out), out.stride(0), M, x.shape[1], _stream()),
          "umv_quantize_act_fp8")
    return out


def _sample_fields(sample, amax):
    if sample is None:
        return {}
    if amax is None:
        raise _lib.UmvError("gemm: sample=(temperature, seed, step) is a mode of the argmax_partial epilogue")
    t, seed, step = sample
    if not float(t) > 0.0:
        raise _lib.UmvError(f"gemm: sampling temperature must be > 0 (got {t})")
    return dict(sample_temperature=float(t), sample_seed=int(seed) & (2 ** 64 - 1), sample_step=None if step is None else step.data_ptr())


def gemm_splitk(x, lin, partials, k_splits, *, M=None):
    """Split-K decode GEMM (M <= 64): raw fp32 partial sums [k_splits, rows, N] into `partials`; the consumer
    (qkv_post(partials=...) / residual_rmsnorm) adds the splits and finishes the row."""
    lib = _lib.load()
    _req(x, BF16, "x")
    _req(partials, torch.float32, "partials")
    M = x.shape[0] if M is None else M
    assert partials.dim() == 3 and partials.shape[0] == k_splits and partials.shape[2] == lin.N and partials.is_contiguous()
    assert lin.th == 16 and not lin.swiglu
    a = GemmArgs(x=x.data_ptr(), ldx=x.stride(0), wp=None if lin.wp is None else lin.wp.data_ptr(), out=partials.data_ptr(), ldo=lin.N,
                 M=M, N=lin.N, K=lin.K, epilogue=0, tile_rows=0, k_splits=k_splits, split_stride=partials.stride(0))
    if lin.w8 is not None:   # e4m3 image: same split, same consumers
        a.wp, a.w_scale = lin.w8.data_ptr(), lin.scale.data_ptr()
        check(lib.umv_gemm_fp8w(C.byref(a), _stream()), "umv_gemm_fp8w")
    else:
        check(lib.umv_gemm_bf16(C.byref(a), _stream()), "umv_gemm_bf16")
    return partials


def residual_rmsnorm(partials, seq, w, eps, out):
    """seq = bf16(bf16(sum_s partials[s]) + seq) in place; out = RMSNorm(seq) * w.  partials [S, T, H] fp32."""
    lib = _lib.load()
    _req(partials, torch.float32, "partials")
    _req(seq, BF16, "seq")
    S, T, H = partials.shape
    assert seq.shape == (T, H) and seq.is_contiguous() and out.is_contiguous() and partials.is_contiguous()
    check(lib.umv_residual_rmsnorm_bf16(_p(partials), S, partials.stride(0), partials.stride(1), _p(seq), _p(w), _p(out), T, H, eps,
                                        _stream()), "umv_residual_rmsnorm_bf16")
    return out


def qkv_post(qkv, q_out, slab, tok_seg, tok_slot, tok_pos, nq, nkv, hd, eps=1e-6, q_norm=None, k_norm=None,
             q_norm_gen=None, k_norm_gen=None, expert=None, cos_tab=None, sin_tab=None, T=None, fp32_chain=False,
             partials=None, bias=None):
    """partials (fp32 [S, T, (nq+2nkv)*hd]) + bias: take the QKV row from a split-K GEMM instead of `qkv`.
    q_out=None with a keys=False slab: only V is split off (into V^T); q and K are then read in place by attention()."""
    if (q_out is None) != (slab.k is None):
        raise _lib.UmvError("qkv_post: q_out=None (V-only split) goes with a KVSlab(keys=False), and only with it")
    lib = _lib.load()
    if partials is None:
        _req(qkv, BF16, "qkv")
        T = qkv.shape[0] if T is None else T
    else:
        _req(partials, torch.float32, "partials")
        T = partials.shape[1] if T is None else T
    a = QkvPostArgs(
        qkv=None if qkv is None else qkv.data_ptr(),
        qkv_partials=None if partials is None else partials.data_ptr(),
        n_splits=0 if partials is None else partials.shape[0],
        split_stride=0 if partials is None else partials.stride(0),
        qkv_bias=None if bias is None else bias.data_ptr(),
        q_out=None if q_out is None else q_out.data_ptr(), k_slab=None if slab.k is None else slab.k.data_ptr(), vt_slab=slab.vt.data_ptr(),
        tok_seg=tok_seg.data_ptr(), tok_slot=tok_slot.data_ptr(),
        tok_pos=None if tok_pos is None else tok_pos.data_ptr(),
        expert=None if expert is None else expert.data_ptr(),
        q_norm_w=None if q_norm is None else q_norm.data_ptr(), k_norm_w=None if k_norm is None else k_norm.data_ptr(),
        q_norm_w_gen=None if q_norm_gen is None else q_norm_gen.data_ptr(),
        k_norm_w_gen=None if k_norm_gen is None else k_norm_gen.data_ptr(),
        cos_tab=None if cos_tab is None else cos_tab.data_ptr(), sin_tab=None if sin_tab is None else sin_tab.data_ptr(),
        T=T, nq=nq, nkv=nkv, hd=hd, eps=eps, fp32_chain=int(fp32_chain), **slab.strides(), **_paging(slab))
    check(lib.umv_qkv_post(C.byref(a), _stream()), "umv_qkv_post")


def attn_workspace(nseg, nq, hd, max_q, nsplit, device):
    n = _lib.load().umv_attn_workspace_bytes(nseg, nq, hd, max_q, nsplit)
    return torch.empty(max(n, 16) // 4, dtype=torch.float32, device=device)


def attention(q, out, slab, cu_q, kv_len, nq, nkv, hd, causal, max_q, max_kv, nsplit=1, workspace=None, k_packed=None,
              _entry=None, variant=0, stats=None, wave_split=0):
    """q: [T, nq*hd] or [T, nq, hd] rows, possibly a column slice of a wider buffer (row stride = q.stride(0)).
    k_packed: [T, nkv*hd] column slice holding K row-aligned with q (cache-less self-attention): the K slab is not read.
    _entry: (function, checker, name) of another library taking the same umv_attn_args (experimental/ops.py; tests / tools only).
    variant / stats: umv_attn_args.variant (_lib.ATTN_* bits) and the two uint32 rare-path counters (tests / A-B only)."""
    lib = _lib.load()
    _req(q, BF16, "q")
    if q.stride(-1) != 1 or (q.dim() == 3 and q.stride(1) != hd):
        raise _lib.UmvError("attention: q rows must be contiguous [nq * hd] runs")
    strides = slab.strides()
    if k_packed is not None:
        _req(k_packed, BF16, "k_packed")
        if k_packed.dim() != 2 or k_packed.stride(1) != 1 or k_packed.shape[0] != q.shape[0] or k_packed.shape[1] != nkv * hd:
            raise _lib.UmvError("attention: k_packed must be a [T, nkv*hd] view with unit column stride")
        k_ptr, k_key_stride = k_packed.data_ptr(), k_packed.stride(0)
        strides["k_head_stride"] = hd
    else:
        if slab.k is None:
            raise _lib.UmvError("attention: this KVSlab holds no keys (keys=False) - pass k_packed")
        k_ptr, k_key_stride = slab.k.data_ptr(), 0
    a = AttnArgs(
        q=q.data_ptr(), out=out.data_ptr(), cu_q=cu_q.data_ptr(), kv_len=kv_len.data_ptr(),
        k_slab=k_ptr, vt_slab=slab.vt.data_ptr(), nseg=kv_len.numel(), nq=nq, nkv=nkv, hd=hd,
        causal=int(bool(causal)), max_q=max_q, max_kv=max_kv, nsplit=nsplit,
        workspace=None if workspace is None else workspace.data_ptr(), q_row_stride=q.stride(0), k_key_stride=k_key_stride,
        variant=int(variant), stats=None if stats is None else stats.data_ptr(), wave_split=int(wave_split), **strides, **_paging(slab))
    if _entry is not None:
        fn, chk, name = _entry
        chk(fn(C.byref(a), _stream()), name)
        return out
    check(lib.umv_attn_varlen(C.byref(a), _stream()), "umv_attn_varlen")
    return out


def decode_advance(tok_slot, tok_pos, kv_len):
    lib = _lib.load()
    check(lib.umv_decode_advance(_p(tok_slot), _p(tok_pos), _p(kv_len), tok_slot.numel(), _stream()), "umv_decode_advance")


def decode_step_end(tok_slot, tok_pos, kv_len, ids, in_ids, pred_ids, step_idx):
    """pred_ids[s] = ids, in_ids[s + 1] = ids, counters += 1, s += 1 - the whole bookkeeping of a decode step in one launch."""
    lib = _lib.load()
    for t, name in ((ids, "ids"), (in_ids, "in_ids"), (pred_ids, "pred_ids"), (step_idx, "step_idx")):
        _req(t, torch.int64, name)
    if not (in_ids.is_contiguous() and pred_ids.is_contiguous() and in_ids.shape == pred_ids.shape and in_ids.shape[1] == ids.numel()):
        raise _lib.UmvError("decode_step_end: in_ids / pred_ids must be contiguous [max_len, B]")
    check(lib.umv_decode_step_end(_p(tok_slot), _p(tok_pos), _p(kv_len), _p(ids), _p(in_ids), _p(pred_ids), _p(step_idx),
                                  ids.numel(), in_ids.shape[0], _stream()), "umv_decode_step_end")


def decode_step_end_argmax(tok_slot, tok_pos, kv_len, argmax_partial, ids, in_ids, pred_ids, step_idx):
    """ids = argmax over the per-tile keys the lm_head GEMM left in argmax_partial, then decode_step_end's bookkeeping.
    step_idx: one counter per sample ([B] int64, all equal)."""
    lib = _lib.load()
    for t, name in ((argmax_partial, "argmax_partial"), (ids, "ids"), (in_ids, "in_ids"), (pred_ids, "pred_ids"), (step_idx, "step_idx")):
        _req(t, torch.int64, name)
    B = ids.numel()
    if step_idx.numel() < B:
        raise _lib.UmvError(f"decode_step_end_argmax: step_idx holds {step_idx.numel()} counters for {B} samples")
    if not (in_ids.is_contiguous() and pred_ids.is_contiguous() and in_ids.shape == pred_ids.shape and in_ids.shape[1] == B
            and argmax_partial.is_contiguous() and argmax_partial.shape[0] == B):
        raise _lib.UmvError("decode_step_end_argmax: in_ids / pred_ids [max_len, B], argmax_partial [B, n_tiles], all contiguous")
    check(lib.umv_decode_step_end_argmax(_p(tok_slot), _p(tok_pos), _p(kv_len), _p(argmax_partial), argmax_partial.shape[1], _p(ids),
                                         _p(in_ids), _p(pred_ids), _p(step_idx), B, in_ids.shape[0], _stream()),
          "umv_decode_step_end_argmax")


def timestep_embed(t, freqs):
    """[n] fp32 timesteps (device) x [half] fp32 frequencies -> [n, 2*half] bf16 sinusoid (cos | sin)"""
    lib = _lib.load()
    _req(t, torch.float32, "t")
    _req(freqs, torch.float32, "freqs")
    n, half = t.numel(), freqs.numel()
    out = torch.empty((n, 2 * half), dtype=BF16, device=t.device)
    check(lib.umv_timestep_embed(_p(t), _p(freqs), _p(out), n, half, _stream()), "umv_timestep_embed")
    return out


def cfg_renorm_euler(x_t, v_t, v_text, v_img, rows, seg_off, nseg, s_text, s_img, renorm_min, rtype, dt):
    lib = _lib.load()
    _req(x_t, torch.float32, "x_t")
    _req(v_t, BF16, "v_t")
    check(lib.umv_cfg_renorm_euler(_p(x_t), _p(v_t), _p(v_text), _p(v_img), v_t.stride(0), _p(rows), _p(seg_off), nseg,
                                   float(s_text), float(s_img), float(renorm_min), int(rtype), float(dt), x_t.shape[1],
                                   _stream()), "umv_cfg_renorm_euler")
