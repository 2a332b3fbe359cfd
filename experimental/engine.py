"""Host side of the decode layer engine (csrc/decode_engine.hip, include/unimedvl_hip.h: umv_decode_engine).

One persistent launch runs a CHAIN of the weight-streaming linears of a decode step (Qwen2MoTDecoderLayer.forward_inference,
codes/modeling/unimedvl/qwen2_navit.py:843-902, inside the token loop of Bagel.generate_text, bagel.py:1262-1314):

    o_proj + residual -> post_attention_layernorm -> gate/up + SwiGLU -> down_proj (8 K groups, fp32 partial sums)
      -> sum + residual -> the next input_layernorm (or the final norm) -> q/k/v_proj + bias

with the weight stream of every op already in flight while the previous op's rows travel between the workgroups.
This module only builds the op tables (device memory, read with scalar loads) and owns the counters.

EXPERIMENTAL: bit-identical to the kernel chain (tests/test_decode_engine_gpu.py) but slower on MI355X (116 vs 93 us per layer
for these ops, profiles/HISTORY.md section 5b) - every hand-off between workgroups costs 3-4 loaded memory round trips.  It lives in
experimental/lib/libunimedvl_hip_experimental.so and is not wired into decode.py.
"""
import ctypes as C

import torch

from unimedvl_amd import ops

from . import _lib
from ._lib import DeOp

BF16 = torch.bfloat16
DE_GEMM, DE_REDUCE = 0, 1
EPI_BF16, EPI_RESIDUAL, EPI_PARTIAL = 0, 1, 2
SIG_XCD, SIG_UNIT_DIV, SIG_GROUP_END = 0, 1, 2
COUNTER_STRIDE = 16      # words between counter words (64 bytes: one per L2 line)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class Counters:
    """A pool of arrival counters: `block(n)` hands out n words (each on its own 64-byte line)."""

    def __init__(self, device, n_words=64):
        self.buf = torch.zeros(n_words * COUNTER_STRIDE, dtype=torch.int32, device=device)
        self.used = 0
        self.n_words = n_words

    def block(self, n):
        if self.used + n > self.n_words:
            raise _lib.UmvError("decode engine: out of counter words")
        p = self.buf.data_ptr() + self.used * COUNTER_STRIDE * 4
        self.used += n
        return p


def gemm_op(lin, x, *, out, KT=None, norm_w=None, eps=1e-6, resid=None, epi=EPI_BF16, kgroups=1, rot=0, publish=False,
            wait=None, wait_target=0, wait_mode=0, sig=None, sig_mode=SIG_XCD, sig_div=1, split_stride=0, use_bias=True,
            ss_out=None, ss_in=None, ss_n=0):
    """One GEMM op of a chain over the PackedLinear `lin` (16-row bf16 image)."""
    if lin.th != 16 or lin.wp is None:
        raise _lib.UmvError("decode engine: needs the standard bf16 weight image")
    KT = lin.K // 32 if KT is None else KT
    assert lin.K % 32 == 0 and lin.N % 16 == 0
    o = DeOp()
    o.w = lin.wp.data_ptr()
    o.x, o.ldx = x.data_ptr(), x.stride(0)
    o.norm_w, o.norm_eps = _ptr(norm_w), eps
    o.kind = DE_GEMM
    o.bias = _ptr(lin.bias) if (use_bias and lin.bias is not None and not lin.swiglu) else 0
    o.resid, o.ldr = (_ptr(resid), resid.stride(0)) if resid is not None else (0, 0)
    o.out, o.ldo, o.split_stride = out.data_ptr(), out.stride(-2), split_stride
    o.wait_cnt, o.wait_target, o.wait_mode = (wait or 0), wait_target, wait_mode
    o.sig_cnt, o.sig_mode, o.sig_div = (sig or 0), sig_mode, sig_div
    o.KT, o.ntiles, o.pair, o.kgroups, o.rot = KT, lin.N // 16, 1 if lin.swiglu else 0, kgroups, rot
    o.epi, o.publish = epi, 1 if publish else 0
    o.ss_out, o.ss_in, o.ss_n = _ptr(ss_out), _ptr(ss_in), ss_n
    return o


def reduce_op(partials, seq, *, kgroups, wait=None, wait_target=0, sig=None, sig_div=1, publish=True, ss_out=None):
    """seq[:, tile] = bf16(bf16(sum_g partials[g][:, tile]) + seq[:, tile]) - workgroup i finishes 16-column tile i."""
    o = DeOp()
    o.kind = DE_REDUCE
    o.x, o.ldx, o.split_stride = partials.data_ptr(), partials.stride(1), partials.stride(0)
    o.resid, o.ldr = seq.data_ptr(), seq.stride(0)
    o.wait_cnt, o.wait_target, o.wait_mode = (wait or 0), wait_target, 1
    o.sig_cnt, o.sig_mode, o.sig_div = (sig or 0), SIG_XCD, sig_div
    o.ntiles, o.kgroups, o.publish = seq.shape[1] // 16, kgroups, 1 if publish else 0
    o.ss_out = _ptr(ss_out)
    return o


class EngineProgram:
    """An op table in device memory plus everything a launch needs."""

    def __init__(self, op_list, M, device, counters=None, grid=256):
        self.n = len(op_list)
        arr = (DeOp * self.n)(*op_list)
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.M, self.grid = M, grid
        self.counters = counters
        self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self.dummy = torch.zeros(512, dtype=BF16, device=device)

    def launch(self, trace=None):
        lib = _lib.load()
        cnt = self.counters.buf if self.counters is not None else None
        if trace is not None:      # tuning only: [grid][64] int64 timeline of the lead service waves
            _lib.check(lib.umv_decode_engine_traced(self.table.data_ptr(), self.n, self.M, _ptr(cnt), 0 if cnt is None else cnt.numel(),
                                               self.err.data_ptr(), self.dummy.data_ptr(), self.grid, trace.data_ptr(), ops._stream()),
                  "umv_decode_engine_traced")
            return
        _lib.check(lib.umv_decode_engine(self.table.data_ptr(), self.n, self.M, _ptr(cnt), 0 if cnt is None else cnt.numel(),
                                    self.err.data_ptr(), self.dummy.data_ptr(), self.grid, ops._stream()), "umv_decode_engine")

    def check_error(self):
        e = int(self.err.item())
        if e != 0:
            raise _lib.UmvError(f"decode engine: a bounded wait timed out (code {e & 0xFFFFFFFF:#x}): results are invalid")


def layer_chain(lw, nxt_norm, nxt_qkv, *, attn_out, seq, act, p_h, qkv_out, eps, device, counters, ss, G=256, kgroups=8):
    """The op list of one decoder layer after its attention: o_proj ... (next) q/k/v_proj.  `nxt_qkv` None = last layer:
    the chain ends with the residual stream complete in `seq` (the final norm + lm_head are the caller's)."""
    H = lw.o.N
    I = lw.gate_up.N // 2
    n_o = H // 16                               # o_proj tiles: one per workgroup, G - n_o workgroups idle
    c_o = counters.block(8)                     # E1: o_proj tiles published (XCD shards)
    c_gu = counters.block(kgroups)              # E2: SwiGLU pairs published, per K group of down_proj
    c_dn = counters.block(G // kgroups)         # E3a: down_proj workgroups done, per n-group
    c_rd = counters.block(8)                    # E3b: residual tiles published (XCD shards)
    pairs_per_kg = (I // 16) // kgroups
    assert (I // 16) % kgroups == 0 and (lw.down.K // 32) % kgroups == 0 and G % kgroups == 0
    assert (H // 16) % (G // kgroups) == 0
    # ss [2][H/16][8] fp32: per-tile row sums of squares of the residual stream after o_proj / after down_proj
    chain = [
        gemm_op(lw.o, attn_out, out=seq, resid=seq, epi=EPI_RESIDUAL, publish=True, sig=c_o, sig_mode=SIG_XCD, ss_out=ss[0]),
        gemm_op(lw.gate_up, seq, out=act, norm_w=lw.post_norm, eps=eps, publish=True, wait=c_o, wait_target=n_o, wait_mode=0,
                sig=c_gu, sig_mode=SIG_UNIT_DIV, sig_div=pairs_per_kg, ss_in=ss[0], ss_n=n_o),
        gemm_op(lw.down, act, out=p_h, epi=EPI_PARTIAL, kgroups=kgroups, publish=True, split_stride=p_h.stride(0),
                wait=c_gu, wait_target=pairs_per_kg, wait_mode=1, sig=c_dn, sig_mode=SIG_GROUP_END),
        reduce_op(p_h, seq, kgroups=kgroups, wait=c_dn, wait_target=kgroups, sig=c_rd, sig_div=(H // 16) // (G // kgroups),
                  ss_out=ss[1]),
    ]
    if nxt_qkv is not None:
        # 288 q/k/v tiles on 256 workgroups: the 32 workgroups that had no o_proj tile take the second one
        chain.append(gemm_op(nxt_qkv, seq, out=qkv_out, norm_w=nxt_norm, eps=eps, rot=(G - n_o) % G, wait=c_rd, wait_target=n_o,
                             wait_mode=0, ss_in=ss[1], ss_n=n_o))
    return chain
