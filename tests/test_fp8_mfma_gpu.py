"""W8A8 path of the fp8 mode (umv_quantize_act_fp8 + umv_gemm_fp8a8w on v_mfma_scale_f32_16x16x128_f8f6f4), through the
C ABI, against oracle/fp8.py.  No reference counterpart exists (see oracle/fp8.py); what is pinned here:
  * the activation quantiser == the oracle's per-row quantiser bit for bit (codes and power-of-two scales);
  * the GEMM == an fp32-accumulated product of the EXACTLY dequantised operands (products and scales are exact, only the
    order of the fp32 sums differs): within 1-2 bf16 ulp of the rounded intermediates, >= 97 % bit-identical;
  * every epilogue (bias, residual, SwiGLU, MoT row routing) and ragged / padded shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops
    return ops


@pytest.mark.parametrize("M,K", [(5, 512), (130, 200), (64, 3584)])
def test_quantize_act_matches_oracle(M, K):
    ops = _ops()
    from oracle import fp8
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g) * 3)).to(BF16)
    x[1] = 0
    xq, xs = ops.quantize_act(x.cuda())
    q, scale, _ = fp8.quantize_act_rows(x)
    assert torch.equal(xs.cpu(), scale)
    assert torch.equal(xq.cpu()[:, :K], q) and not xq.cpu()[:, K:].any()
    idx = torch.tensor([3, 0, 4], dtype=torch.int32) if M >= 5 else None
    if idx is not None:
        xq2, xs2 = ops.quantize_act(x.cuda(), M=3, row_idx=idx.cuda())
        assert torch.equal(xq2.cpu()[:, :K], q[idx.long()]) and torch.equal(xs2.cpu(), scale[idx.long()])


def _close(got, ref, mag=None):
    """The dequantised operands live on a coarse power-of-two grid, so the exact sums sit on or next to bf16 rounding
    ties far more often than random reals would; the fp32 accumulation order then decides ~1 % of the roundings.  Bound:
    one ulp of the largest intermediate the epilogue rounds (|mag|), doubled when a second rounding follows (residual)."""
    mag = ref.float().abs() if mag is None else mag
    tol = mag * 2.0 ** -6 + 1e-3 * ref.float().abs().max()
    assert ((got.float() - ref.float()).abs() <= tol).all(), f"max diff {(got.float() - ref.float()).abs().max().item()}"
    assert (got == ref).float().mean() >= 0.97


@pytest.mark.parametrize("M,N,K", [(128, 256, 512), (1000, 1152, 1024), (2048, 3584, 3584), (129, 320, 200), (65, 48, 128), (700, 4608, 3584),
                                   (3072, 3584, 512), (3001, 3100, 200)])   # the last two: >= 144 tiles of 256 x 256 -> the big tile
def test_w8a8_gemm_vs_dequantised_product(M, N, K):
    ops = _ops()
    from oracle import fp8
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 2).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16)
    b = torch.randn(N, generator=g).to(BF16)
    lin = ops.PackedLinear.from_weight_fp8(w.cuda(), b.cuda()).enable_fp8_mfma()
    assert lin.wp is None and lin.w8m.numel() == ((N + 15) // 16) * ((K + 127) // 128) * 2048
    res = torch.randn(M, N, generator=g).to(BF16)
    got = ops.gemm(x.cuda(), lin, residual=res.cuda(), act8=True)
    wd = fp8.quantize_rows(w)[2]
    xd = fp8.quantize_act_rows(x)[2]
    acc = (xd.cuda().double() @ wd.cuda().double().t()).float() + b.cuda().float()      # exact products, well-conditioned sum
    ref = (acc.to(BF16).float() + res.cuda().float()).to(BF16)
    _close(got, ref, acc.abs() + res.cuda().float().abs())


@pytest.mark.parametrize("M", [100, 1024])
def test_w8a8_swiglu_and_routing(M):
    ops = _ops()
    from oracle import fp8
    g = torch.Generator().manual_seed(M)
    K, I = 640, 512
    x = torch.randn(M + 40, K, generator=g).to(BF16)
    gate = (torch.randn(I, K, generator=g) * 0.05).to(BF16)
    up = (torch.randn(I, K, generator=g) * 0.05).to(BF16)
    lin = ops.PackedLinear.from_gate_up_fp8(gate.cuda(), up.cuda()).enable_fp8_mfma()
    idx = torch.randperm(M + 40, generator=g)[:M].to(torch.int32)
    out = torch.zeros(M + 40, I, dtype=BF16, device="cuda")
    ops.gemm(x.cuda(), lin, out=out, M=M, row_idx=idx.cuda(), act8=True)
    xd = fp8.quantize_act_rows(x[idx.long()])[2].cuda().double()
    gd, ud = fp8.quantize_rows(gate)[2].cuda().double(), fp8.quantize_rows(up)[2].cuda().double()
    gg = (xd @ gd.t()).float().to(BF16).float()
    uu = (xd @ ud.t()).float().to(BF16).float()
    ref = (torch.nn.functional.silu(gg).to(BF16).float() * uu).to(BF16)
    _close(out[idx.long().cuda()], ref, (torch.nn.functional.silu(gg).abs() + 1) * (uu.abs() + 2.0 ** -7 * (xd.abs() @ ud.abs().t()).float()))
    untouched = torch.ones(M + 40, dtype=torch.bool)
    untouched[idx.long()] = False
    assert not out[untouched.cuda()].any()


@pytest.mark.parametrize("M", [1, 8, 40, 64])
def test_w8a8_small_m_rounds_activations_too(M):
    """M <= 64 rows run the weight-streaming kernels on a bf16 copy of the e4m3-rounded rows: same operands as the MFMA path."""
    ops = _ops()
    from oracle import fp8
    g = torch.Generator().manual_seed(M)
    K, N = 1024, 768
    x = (torch.randn(M + 5, K, generator=g) * 2).to(BF16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16)
    lin = ops.PackedLinear.from_weight_fp8(w.cuda()).enable_fp8_mfma()
    idx = torch.randperm(M + 5, generator=g)[:M].to(torch.int32)
    out = torch.zeros(M + 5, N, dtype=BF16, device="cuda")
    ops.gemm(x.cuda(), lin, out=out, M=M, row_idx=idx.cuda(), act8=True)
    xd = fp8.quantize_act_rows(x[idx.long()])[2].cuda().double()
    ref = (xd @ fp8.quantize_rows(w)[2].cuda().double().t()).float().to(BF16)
    _close(out[idx.long().cuda()], ref)
    # without act8 the same linear still serves decode rows (bf16 activations), but refuses M > 64
    ops.gemm(x.cuda(), lin, out=out, M=M, row_idx=idx.cuda())
    from unimedvl_amd import _lib
    with pytest.raises(_lib.UmvError, match="act8"):
        ops.gemm(torch.randn(100, K).to(BF16).cuda(), lin)


def test_w8a8_rejects_bad_arguments():
    ops = _ops()
    from unimedvl_amd import _lib
    lin = ops.PackedLinear.from_weight(torch.randn(64, 128).to(BF16).cuda())
    with pytest.raises(_lib.UmvError, match="fp8 weights"):
        lin.enable_fp8_mfma()
    with pytest.raises(_lib.UmvError, match="act8 needs"):
        ops.gemm(torch.randn(4, 128).to(BF16).cuda(), lin, act8=True)
