"""umv_attn_decode_fused (EXPERIMENTAL, experimental/csrc/attention_decode.hip: q/k RMSNorm + RoPE + KV append folded into the decode attention) against the two-kernel path
umv_qkv_post + umv_attn_varlen it replaces, through the C ABI.  Both run the same MFMAs on the same operands; only the
row sum of squares of the q/k norms is accumulated in another order, so: V^T column bit-exact, K row / attention output
within 1 bf16 ulp of the value and >= 99 % / 97 % bit-identical.  Ragged lengths put the new key at every position of a
32-key block and in the first / a middle / the last split."""
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


@pytest.mark.parametrize("nsplit", [1, 4, 17, 32])
def test_fused_decode_attention_matches_two_kernel_path(nsplit):
    ops = _ops()
    nq, nkv, hd = 28, 4, 128
    lens = [1, 2, 31, 32, 33, 64, 65, 100, 511, 1060, 1061, 2049]     # kv_len INCLUDING the new token
    B = len(lens)
    g = torch.Generator().manual_seed(nsplit)
    cap = 2080
    N = (nq + 2 * nkv) * hd
    qkv = (torch.randn(B, N, generator=g) * 2).to(BF16).cuda()
    qn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    kn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    ang = torch.rand(4096, hd, generator=g) * 6.28
    cos, sin = ang.cos().to(BF16).cuda(), ang.sin().to(BF16).cuda()
    pos = torch.tensor([l + 7 for l in lens], dtype=torch.int32).cuda()
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    slot = kv_len - 1
    seg = torch.arange(B, dtype=torch.int32).cuda()
    cu = torch.arange(B + 1, dtype=torch.int32).cuda()
    hist_k = torch.randn(B, nkv, cap, hd, generator=g).to(BF16)
    hist_v = torch.randn(B, nkv, hd, cap, generator=g).to(BF16)
    ws = ops.attn_workspace(B, nq, hd, 1, nsplit, "cuda") if nsplit > 1 else None

    def fresh():
        s = ops.KVSlab(B, nkv, cap, hd, "cuda")
        s.k.copy_(hist_k)
        s.vt.copy_(hist_v)
        return s
    ref_slab, q = fresh(), torch.zeros(B, nq, hd, dtype=BF16, device="cuda")
    ops.qkv_post(qkv, q, ref_slab, seg, slot, pos, nq, nkv, hd, 1e-6, qn, kn, cos_tab=cos, sin_tab=sin)
    ref = torch.zeros(B, nq * hd, dtype=BF16, device="cuda")
    ops.attention(q, ref, ref_slab, cu, kv_len, nq, nkv, hd, True, 1, cap, nsplit, ws)
    got_slab = fresh()
    got = torch.zeros(B, nq * hd, dtype=BF16, device="cuda")
    _xops().attn_decode_fused(qkv, got, got_slab, cu, kv_len, pos, nq, nkv, hd, 1e-6, qn, kn, cos, sin, nsplit, ws)
    torch.cuda.synchronize()
    assert torch.equal(got_slab.vt, ref_slab.vt), "V^T column"
    dk = (got_slab.k.float() - ref_slab.k.float()).abs()
    assert (dk <= ref_slab.k.float().abs() * 2.0 ** -7 + 1e-6).all() and (got_slab.k == ref_slab.k).float().mean() > 0.999
    for b, l in enumerate(lens):   # only the new row of every segment was touched
        assert torch.equal(got_slab.k[b, :, :l - 1], hist_k[b, :, :l - 1].cuda()) and torch.equal(got_slab.k[b, :, l:], hist_k[b, :, l:].cuda())
    d = (got.float() - ref.float()).abs()
    tol = ref.float().abs() * 2.0 ** -7 + 2e-3 * ref.float().abs().max()
    assert (d <= tol).all(), f"max diff {d.max().item()}"
    assert (got == ref).float().mean() > 0.97


@pytest.mark.parametrize("n_splits,with_bias", [(1, False), (3, True), (4, False)])
def test_fused_decode_attention_from_splitk_partials(n_splits, with_bias):
    """QKV row handed over as the fp32 partial sums of a split-K GEMM: the kernel's own sum (split order, + bias, one bf16
    rounding) must give bit for bit what it gives on the bf16 row holding those sums."""
    ops = _ops()
    nq, nkv, hd = 28, 4, 128
    lens = [1, 33, 64, 100, 777, 1060]
    B = len(lens)
    g = torch.Generator().manual_seed(n_splits)
    cap, nsplit = 1088, 17
    N = (nq + 2 * nkv) * hd
    part = torch.randn(n_splits, B, N, generator=g).cuda()
    bias = torch.randn(N, generator=g).to(BF16).cuda() if with_bias else None
    acc = torch.zeros(B, N, device="cuda")
    for s in range(n_splits):
        acc = acc + part[s]
    if with_bias:
        acc = acc + bias.float()
    qkv = acc.to(BF16)
    qn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    kn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    ang = torch.rand(2048, hd, generator=g) * 6.28
    cos, sin = ang.cos().to(BF16).cuda(), ang.sin().to(BF16).cuda()
    pos = torch.tensor([l + 3 for l in lens], dtype=torch.int32).cuda()
    kv_len = torch.tensor(lens, dtype=torch.int32).cuda()
    cu = torch.arange(B + 1, dtype=torch.int32).cuda()
    hist_k = torch.randn(B, nkv, cap, hd, generator=g).to(BF16)
    hist_v = torch.randn(B, nkv, hd, cap, generator=g).to(BF16)
    ws = ops.attn_workspace(B, nq, hd, 1, nsplit, "cuda")
    outs = []
    for kw in (dict(), dict(partials=part, bias=bias)):
        slab = ops.KVSlab(B, nkv, cap, hd, "cuda")
        slab.k.copy_(hist_k)
        slab.vt.copy_(hist_v)
        out = torch.zeros(B, nq * hd, dtype=BF16, device="cuda")
        _xops().attn_decode_fused(None if kw else qkv, out, slab, cu, kv_len, pos, nq, nkv, hd, 1e-6, qn, kn, cos, sin, nsplit, ws, **kw)
        torch.cuda.synchronize()
        outs.append((out, slab.k.clone(), slab.vt.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert outs[0][0].abs().sum() > 0
