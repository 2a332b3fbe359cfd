"""Decode layer engine (EXPERIMENTAL; experimental/engine.py, experimental/csrc/decode_engine.hip): one persistent launch for
o_proj + residual -> RMSNorm -> gate/up + SwiGLU -> down_proj (8 K groups) -> sum + residual -> RMSNorm -> q/k/v_proj
(qwen2_navit.py:617-620,873-898,541-543; modeling_qwen2.py:234-235) must reproduce the kernel chain of decode.py::_step
bit for bit at the full widths - every value crosses workgroups through the in-launch hand-off protocol (write-through
stores, arrival counters, sc1 loads), so a stale or torn hand-off shows up as a wrong element."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _xops():
    """the experimental package (experimental/): skip when its library has not been built (python -m experimental.build)"""
    from experimental import _lib as xlib
    if not xlib.available():
        pytest.skip("experimental library not built (python -m experimental.build)")
    from experimental import ops as xops
    return xops

H, I, QKV = 3584, 18944, 4608
EPS = 1e-6
BF16 = torch.bfloat16


class _LW:
    pass


@pytest.fixture(scope="module")
def layer():
    from unimedvl_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)

    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device=dev, generator=gen) * std).to(BF16)
    lw = _LW()
    lw.o = ops.PackedLinear.from_weight(rn(H, H))
    lw.gate_up = ops.PackedLinear.from_gate_up(rn(I, H), rn(I, H))
    lw.down = ops.PackedLinear.from_weight(rn(H, I))
    lw.qkv = ops.PackedLinear.from_weight(rn(QKV, H), rn(QKV))
    lw.post_norm = (1.0 + 0.1 * torch.randn(H, device=dev, generator=gen)).to(BF16)
    lw.in_norm = (1.0 + 0.1 * torch.randn(H, device=dev, generator=gen)).to(BF16)
    return lw, dev, gen


def _classic(lw, attn, seq, B):
    from unimedvl_amd import ops
    x = torch.empty(B, H, dtype=BF16, device=seq.device)
    act = torch.empty(B, I, dtype=BF16, device=seq.device)
    p_h = torch.empty(8, B, H, dtype=torch.float32, device=seq.device)
    qkv = torch.empty(B, QKV, dtype=BF16, device=seq.device)
    ops.gemm(attn, lw.o, out=seq, residual=seq)
    ops.rmsnorm(seq, lw.post_norm, EPS, out=x)
    ops.gemm(x, lw.gate_up, out=act)
    ops.gemm_splitk(act, lw.down, p_h, 8)
    ops.residual_rmsnorm(p_h, seq, lw.in_norm, EPS, out=x)
    ops.gemm(x, lw.qkv, out=qkv)
    return seq, act, p_h, qkv


@pytest.mark.parametrize("B", [8, 3])
def test_engine_chain_equals_kernel_chain(layer, B):
    _xops()
    from experimental import engine
    lw, dev, gen = layer
    act = torch.empty(B, I, dtype=BF16, device=dev)
    p_h = torch.empty(8, B, H, dtype=torch.float32, device=dev)
    qkv = torch.empty(B, QKV, dtype=BF16, device=dev)
    ss = torch.zeros(2, H // 16, 8, dtype=torch.float32, device=dev)
    attn = torch.empty(B, H, dtype=BF16, device=dev)
    seq = torch.empty(B, H, dtype=BF16, device=dev)
    cnt = engine.Counters(dev, 128)
    prog = engine.EngineProgram(engine.layer_chain(lw, lw.in_norm, lw.qkv, attn_out=attn, seq=seq, act=act, p_h=p_h, qkv_out=qkv,
                                                   eps=EPS, device=dev, counters=cnt, ss=ss), B, dev, cnt)
    for it in range(6):      # new inputs every launch: the buffers the consumers read were written by the previous launch
        a = torch.randn(B, H, device=dev, generator=gen).to(BF16) * (1.0 + it)
        s0 = torch.randn(B, H, device=dev, generator=gen).to(BF16)
        attn.copy_(a)
        seq.copy_(s0)
        prog.launch()
        torch.cuda.synchronize()
        prog.check_error()
        rs, ra, rp, rq = _classic(lw, a.clone(), s0.clone(), B)
        assert torch.equal(seq, rs), f"residual stream differs (launch {it})"
        assert torch.equal(act, ra), f"SwiGLU activations differ (launch {it})"
        assert torch.equal(p_h, rp), f"down_proj partial sums differ (launch {it})"
        # the RMSNorm scale comes from per-tile partial sums of squares (another fp32 summation order than umv_rmsnorm's):
        # a row's scale may differ in the last bit, which can move a normalised value - and then a q/k/v output - by one bf16 ulp
        same = (qkv == rq).float().mean().item()
        assert same > 0.999, f"q/k/v rows: only {same * 100:.3f} % bit-identical (launch {it})"
        assert (qkv.float() - rq.float()).abs().max().item() <= 2.0 ** -6 * rq.float().abs().max().item()


def test_engine_rejects_more_than_eight_rows(layer):
    _xops()
    from unimedvl_amd import _lib
    from experimental import engine
    lw, dev, gen = layer
    B = 9
    act = torch.empty(B, I, dtype=BF16, device=dev)
    p_h = torch.empty(8, B, H, dtype=torch.float32, device=dev)
    qkv = torch.empty(B, QKV, dtype=BF16, device=dev)
    ss = torch.zeros(2, H // 16, 8, dtype=torch.float32, device=dev)
    attn = torch.zeros(B, H, dtype=BF16, device=dev)
    seq = torch.zeros(B, H, dtype=BF16, device=dev)
    cnt = engine.Counters(dev, 128)
    prog = engine.EngineProgram(engine.layer_chain(lw, lw.in_norm, lw.qkv, attn_out=attn, seq=seq, act=act, p_h=p_h, qkv_out=qkv,
                                                   eps=EPS, device=dev, counters=cnt, ss=ss), B, dev, cnt)
    with pytest.raises(_lib.UmvError):
        prog.launch()
