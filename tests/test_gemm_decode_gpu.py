"""Persistent decode GEMM (umv_gemm_decode, EXPERIMENTAL, experimental/csrc/gemm_decode.hip) through the C ABI.
The decode image only re-tiles the weight; the K split and summation order are those of the one-tile-per-workgroup
kernels, so the results must be BIT-IDENTICAL to umv_gemm_bf16 / umv_gemm_fp8w on the same weight (which are themselves
parity-tested against the fp32 / oracle references in test_kernels_gpu.py and test_fp8_gpu.py).  The fused RMSNorm
prologue is compared with the standalone rmsnorm kernel + GEMM at 1 bf16 ulp of the normalised activations
(the row sum of squares is accumulated in a different order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _xops():
    """the experimental package (experimental/): skip when its library has not been built (python -m experimental.build)"""
    from experimental import _lib as xlib
    if not xlib.available():
        pytest.skip("experimental library not built (python -m experimental.build)")
    from experimental import ops as xops
    return xops


def _ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops
    return ops


def _mk(ops, N, K, swiglu, fp8, g, bias=True):
    if swiglu:
        gate = (torch.randn(N // 2, K, generator=g) * 0.05).to(BF16).cuda()
        up = (torch.randn(N // 2, K, generator=g) * 0.05).to(BF16).cuda()
        return (ops.PackedLinear.from_gate_up_fp8 if fp8 else ops.PackedLinear.from_gate_up)(gate, up)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).cuda()
    b = torch.randn(N, generator=g).to(BF16).cuda() if bias else None
    return (ops.PackedLinear.from_weight_fp8 if fp8 else ops.PackedLinear.from_weight)(w, b)


SHAPES = [
    # N, K, swiglu         (the model's decode shapes + ragged / tiny / non-multiple cases)
    (4608, 3584, False), (3584, 3584, False), (37888, 3584, True), (3584, 18944, False),
    (1000, 512, False), (96, 256, True), (40, 64, False), (4960, 1024, True), (320, 4096, False),
]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("M", [1, 8, 11, 16])
@pytest.mark.parametrize("N,K,swiglu", SHAPES)
def test_gemm_decode_bit_exact(N, K, swiglu, M, fp8):
    ops = _ops()
    if fp8 and K % 512 != 0:
        pytest.skip("umv_gemm_fp8w and the decode image agree bit for bit only when the 8 K slices coincide (K % 512 == 0)")
    g = torch.Generator().manual_seed(N + K + M)
    lin = _mk(ops, N, K, swiglu, fp8, g)
    dl = _xops().DecodeLinear(lin)
    L = dl.layout
    rows = N // 2 if swiglu else N
    assert L.G * L.C >= rows and L.tpw * L.th >= L.C and L.th <= 16
    x = torch.randn(M, K, generator=g).to(BF16).cuda()
    res = None if swiglu else torch.randn(M, N, generator=g).to(BF16).cuda()
    ref = ops.gemm(x, lin, residual=res)
    got = _xops().gemm_decode(x, dl, residual=res)
    assert torch.equal(got, ref)


def test_gemm_decode_layouts_for_model_shapes():
    ops = _ops()
    from unimedvl_amd import _lib
    import ctypes as C
    _xops()
    from experimental import _lib as xlib
    lib = xlib.load()
    want = {4608: (18, 9, 2), 3584: (14, 14, 1), 18944: (74, 15, 5), 152064: (594, None, None)}
    for rows, (c, th, tpw) in want.items():
        L = xlib.DecodeLayout()
        assert lib.umv_decode_layout_for(rows, 256, C.byref(L)) == 0
        assert L.G == 256 and L.C == c
        if th is not None:
            assert (L.th, L.tpw) == (th, tpw)
        assert (L.tpw * L.th - L.C) / L.C <= 0.015, "at most 1.5% padding rows per slab on the model shapes"


@pytest.mark.parametrize("M", [1, 8, 13, 16])
@pytest.mark.parametrize("N,K,swiglu", [(4608, 3584, False), (37888, 3584, True), (1000, 512, False), (152064, 3584, False)])
def test_gemm_decode_fused_rmsnorm(N, K, swiglu, M):
    ops = _ops()
    g = torch.Generator().manual_seed(N + K + M + 1)
    lin = _mk(ops, N, K, swiglu, False, g, bias=not swiglu and N < 100000)
    dl = _xops().DecodeLinear(lin)
    x = (torch.randn(M, K, generator=g) * 3).to(BF16).cuda()
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF16).cuda()
    xn = ops.rmsnorm(x, nw, 1e-6)
    ref = _xops().gemm_decode(xn, dl)
    got = _xops().gemm_decode(x, dl, norm_w=nw, norm_eps=1e-6)
    # 1 ulp flips of single normalised activations perturb an output by <= 2^-8 * |x_k w_nk|; bound it loosely by the scale
    scale = ref.float().abs().max().clamp_min(1e-3)
    err = (got.float() - ref.float()).abs().max()
    assert err <= 2e-2 * scale, f"fused norm: max err {err:.4g} vs scale {scale:.4g}"
    exact = (got == ref).float().mean().item()
    assert exact > 0.9, f"only {exact:.3f} of the outputs are bit-identical to rmsnorm + GEMM"


def test_gemm_decode_row_idx_and_errors():
    ops = _ops()
    from unimedvl_amd import _lib
    g = torch.Generator().manual_seed(5)
    lin = _mk(ops, 512, 1024, False, False, g)
    dl = _xops().DecodeLinear(lin)
    x = torch.randn(12, 1024, generator=g).to(BF16).cuda()
    idx = torch.tensor([7, 2, 9, 0, 11], dtype=torch.int32).cuda()
    o1 = torch.zeros(12, 512, dtype=BF16, device="cuda")
    o2 = torch.zeros(12, 512, dtype=BF16, device="cuda")
    ops.gemm(x, lin, out=o1, M=5, row_idx=idx)
    _xops().gemm_decode(x, dl, out=o2, M=5, row_idx=idx)
    assert torch.equal(o1, o2)
    with pytest.raises(_lib.UmvError, match="M <= 16"):
        _xops().gemm_decode(torch.randn(17, 1024).to(BF16).cuda(), dl)
    big = _xops().DecodeLinear(_mk(ops, 64, 8192, False, False, g))
    with pytest.raises(_lib.UmvError, match="K <= 4096"):
        _xops().gemm_decode(torch.randn(4, 8192).to(BF16).cuda(), big, norm_w=torch.ones(8192, dtype=BF16, device="cuda"))
