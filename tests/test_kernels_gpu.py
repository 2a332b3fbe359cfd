"""Kernel-level parity: each C-ABI entry point against the CPU oracle's restatement of
the same reference op, on seeded inputs.  Needs an MI355X (-m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m 'not gpu' on CPU-only machines)")
    from unimedvl_amd import ops as o
    return o


def ulp_diff(a, b):
    """distance in bf16 ulps between two bf16 tensors (sign-magnitude ordering)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a) - key(b)).abs()


def assert_close_bf16(got, ref, max_ulp=1, frac_exact=0.98, what=""):
    got, ref = got.cpu(), ref.cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    d = ulp_diff(got, ref)
    # values near zero can be many "ulps" apart while absolutely tiny: fall back to abs tol there
    absd = (got.float() - ref.float()).abs()
    scale = ref.float().abs().max().clamp_min(1e-6)
    bad = (d > max_ulp) & (absd > scale * 2 ** -8)
    assert bad.sum() == 0, f"{what}: {int(bad.sum())} elements off by more than {max_ulp} ulp; worst abs {absd.max():.4g}"
    exact = (d == 0).float().mean().item()
    assert exact >= frac_exact, f"{what}: only {exact:.4f} bit-exact"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF16)


@pytest.mark.parametrize("M,N,K", [(1, 48, 64), (8, 512, 256), (8, 3584, 3584), (16, 4608, 3584), (20, 144, 144),
                                   (33, 320, 256), (64, 1152, 4304), (7, 100, 72)])
def test_gemm_skinny(ops, M, N, K):
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, 1 / math.sqrt(K)), rnd((N,), 3)
    lin = ops.PackedLinear.from_weight(w.cuda(), b.cuda())
    out = ops.gemm(x.cuda(), lin)
    ref = (x.float() @ w.float().T + b.float()).to(BF16)
    assert_close_bf16(out, ref, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M,N,K", [(65, 128, 64), (200, 320, 256), (1030, 1152, 1152), (300, 144, 208), (2048, 3584, 1152)])
def test_gemm_tiled(ops, M, N, K):
    x, w, b = rnd((M, K), 4), rnd((N, K), 5, 1 / math.sqrt(K)), rnd((N,), 6)
    lin = ops.PackedLinear.from_weight(w.cuda(), b.cuda())
    out = ops.gemm(x.cuda(), lin)
    ref = (x.float() @ w.float().T + b.float()).to(BF16)
    assert_close_bf16(out, ref, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("M", [8, 40, 300])
def test_gemm_epilogues(ops, M):
    K, N, I = 256, 320, 384
    x = rnd((M, K), 7)
    w, b, res = rnd((N, K), 8, 1 / 16), rnd((N,), 9), rnd((M, N), 10)
    lin = ops.PackedLinear.from_weight(w.cuda(), b.cuda())
    lin_nb = ops.PackedLinear.from_weight(w.cuda())
    base = x.float() @ w.float().T
    # bias + gelu_tanh (siglip_navit.py:256-257)
    out = ops.gemm(x.cuda(), lin, act="gelu_tanh")
    ref = F.gelu((base + b.float()).to(BF16), approximate="tanh")
    assert_close_bf16(out, ref, what="gelu")
    # bias + silu (modeling_utils.py:80-81)
    out = ops.gemm(x.cuda(), lin, act="silu")
    assert_close_bf16(out, F.silu((base + b.float()).to(BF16)), what="silu")
    # residual (qwen2_navit.py:883)
    out = ops.gemm(x.cuda(), lin_nb, residual=res.cuda())
    assert_close_bf16(out, res + base.to(BF16), what="residual")
    # swiglu (modeling_qwen2.py:235)
    wg, wu = rnd((I, K), 11, 1 / 16), rnd((I, K), 12, 1 / 16)
    sw = ops.PackedLinear.from_gate_up(wg.cuda(), wu.cuda())
    out = ops.gemm(x.cuda(), sw)
    ref = F.silu((x.float() @ wg.float().T).to(BF16)) * (x.float() @ wu.float().T).to(BF16)
    assert_close_bf16(out, ref, what="swiglu")
    # row-indexed (MoT routing): rows 1,3,.. of a larger buffer
    big = rnd((2 * M, K), 13)
    idx = torch.arange(1, 2 * M, 2, dtype=torch.int32)
    outbuf = torch.zeros((2 * M, N), dtype=BF16, device="cuda")
    ops.gemm(big.cuda(), lin, out=outbuf, M=M, row_idx=idx.cuda())
    ref = torch.zeros((2 * M, N), dtype=BF16)
    ref[idx.long()] = (big[idx.long()].float() @ w.float().T + b.float()).to(BF16)
    assert_close_bf16(outbuf, ref, what="row_idx")


@pytest.mark.parametrize("M,N,K,swiglu", [(8, 4608, 3584, False), (8, 37888, 3584, True), (3, 320, 256, False),
                                          (16, 512, 1152, False), (11, 2048, 4096, True), (1, 64, 32, False)])
def test_gemm_fused_rmsnorm(ops, M, N, K, swiglu):
    """decode path: Qwen2RMSNorm folded into the skinny GEMM prologue == rmsnorm kernel then GEMM."""
    from oracle.unimedvl_cpu import rmsnorm
    x, nw = rnd((M, K), 60, 1.5), (1 + 0.1 * torch.randn(K, generator=torch.Generator().manual_seed(61))).to(BF16)
    xn = rmsnorm(x, nw, 1e-6)
    if swiglu:
        wg, wu = rnd((N // 2, K), 62, 1 / math.sqrt(K)), rnd((N // 2, K), 63, 1 / math.sqrt(K))
        lin = ops.PackedLinear.from_gate_up(wg.cuda(), wu.cuda())
        ref = F.silu((xn.float() @ wg.float().T).to(BF16)) * (xn.float() @ wu.float().T).to(BF16)
    else:
        w, b = rnd((N, K), 64, 1 / math.sqrt(K)), rnd((N,), 65)
        lin = ops.PackedLinear.from_weight(w.cuda(), b.cuda())
        ref = (xn.float() @ w.float().T + b.float()).to(BF16)
    out = ops.gemm(x.cuda(), lin, norm_w=nw.cuda(), norm_eps=1e-6)
    two_step = ops.gemm(ops.rmsnorm(x.cuda(), nw.cuda(), 1e-6), lin)
    assert_close_bf16(out, ref, what=f"fused norm gemm {M}x{N}x{K}", frac_exact=0.97)
    assert_close_bf16(out, two_step.cpu(), what="fused vs two-step", frac_exact=0.97)


@pytest.mark.parametrize("M,N,K", [(8, 3584, 3584), (8, 4608, 3584), (8, 3584, 18944), (20, 2560, 256), (3, 3584, 64)])
def test_gemm_exact_partition_tiles(ops, M, N, K):
    """decode-only th-row weight image (th = N/256): same result as the standard 16-row image."""
    x, w, b, res = rnd((M, K), 70), rnd((N, K), 71, 1 / math.sqrt(K)), rnd((N,), 72), rnd((M, N), 73)
    lin = ops.PackedLinear.from_weight(w.cuda(), b.cuda())
    dec = lin.for_decode()
    assert dec.th < 16 and N % dec.th == 0 and (N // dec.th) % 256 == 0
    ref = ops.gemm(x.cuda(), lin, residual=res.cuda())
    out = ops.gemm(x.cuda(), dec, residual=res.cuda())
    assert torch.equal(out.cpu(), ref.cpu()), "exact-partition tiles must not change a single bit"
    with pytest.raises(Exception):
        ops.gemm(rnd((100, K), 74).cuda(), dec)      # decode-only layout


def test_norms(ops):
    from oracle.unimedvl_cpu import rmsnorm
    for T, H in [(8, 3584), (5, 256), (300, 128), (1000, 1152)]:
        x, w = rnd((T, H), 20, 2.0), (1 + 0.1 * torch.randn(H, generator=torch.Generator().manual_seed(21))).to(BF16)
        out = ops.rmsnorm(x.cuda(), w.cuda(), 1e-6)
        assert_close_bf16(out, rmsnorm(x, w, 1e-6), what=f"rmsnorm {T}x{H}", frac_exact=0.995)
    x = rnd((40, 256), 22)
    w0, w1 = rnd((256,), 23), rnd((256,), 24)
    ex = (torch.arange(40) % 3 == 0).to(torch.int32)
    out = ops.rmsnorm(x.cuda(), w0.cuda(), 1e-6, w_gen=w1.cuda(), expert=ex.cuda())
    ref = rmsnorm(x, w0, 1e-6)
    ref[ex.bool()] = rmsnorm(x[ex.bool()], w1, 1e-6)
    assert_close_bf16(out, ref, what="rmsnorm expert", frac_exact=0.995)
    for T, H in [(100, 1152), (7, 144)]:
        x, w, b = rnd((T, H), 25, 1.5), rnd((H,), 26), rnd((H,), 27)
        out = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
        assert_close_bf16(out, F.layer_norm(x, (H,), w, b, 1e-6), what=f"layernorm {T}x{H}", frac_exact=0.99)


def test_gather_add_argmax(ops):
    table = rnd((500, 256), 30)
    ids = torch.randint(0, 500, (37,), generator=torch.Generator().manual_seed(31))
    assert torch.equal(ops.embed_gather(table.cuda(), ids.cuda()).cpu(), table[ids])
    rows = torch.randperm(60)[:37].to(torch.int32)
    buf = torch.zeros((60, 256), dtype=BF16, device="cuda")
    ops.embed_gather(table.cuda(), ids.cuda(), out=buf, out_rows=rows.cuda())
    ref = torch.zeros((60, 256), dtype=BF16)
    ref[rows.long()] = table[ids]
    assert torch.equal(buf.cpu(), ref)
    a, bc = rnd((37, 256), 32), rnd((256,), 33)
    out = torch.zeros((60, 256), dtype=BF16, device="cuda")
    ops.add_rows(a.cuda(), out, bcast=bc.cuda(), table=table.cuda(), idx=ids.cuda(), out_rows=rows.cuda())
    ref = torch.zeros((60, 256), dtype=BF16)
    ref[rows.long()] = a + bc + table[ids]
    assert torch.equal(out.cpu(), ref)
    logits = rnd((8, 152064), 34)
    logits[3, 777] = logits[3].max()  # tie: lowest index must win
    first = int((logits[3] == logits[3].max()).nonzero()[0])
    got = ops.argmax(logits.cuda()).cpu()
    assert torch.equal(got, torch.argmax(logits.float(), -1))
    assert int(got[3]) == first
    px = torch.randn(10, 588, generator=torch.Generator().manual_seed(35))
    cp = ops.cast_pad(px.cuda(), 608).cpu()
    assert torch.equal(cp[:, :588], px.to(BF16)) and (cp[:, 588:] == 0).all()


def test_sampling_matches_softmax_distribution(ops):
    """do_sample path (bagel.py:1297-1299): empirical frequencies of umv_sample_bf16 follow
    softmax(logits / T); same seed -> same draw; different steps -> different draws."""
    V, T = 40, 0.7
    g = torch.Generator().manual_seed(80)
    logits = (torch.randn(1, V, generator=g) * 2).to(BF16)
    p = torch.softmax((logits.float() / T).to(BF16).float(), -1)[0]
    rows = 4096
    big = logits.repeat(rows, 1).cuda()
    counts = torch.zeros(V)
    for s in range(4):
        ids = ops.sample(big, T, seed=1234 + s).cpu()
        assert int(ids.min()) >= 0 and int(ids.max()) < V
        counts += torch.bincount(ids, minlength=V).float()
    n = counts.sum()
    exp = p * n
    mask = exp > 5
    chi2 = (((counts - exp) ** 2) / exp)[mask].sum().item()
    dof = int(mask.sum()) - 1
    assert chi2 < dof + 6 * (2 * dof) ** 0.5, f"chi2 {chi2:.1f} for {dof} dof"
    a = ops.sample(big, T, seed=99).cpu()
    b = ops.sample(big, T, seed=99).cpu()
    assert torch.equal(a, b)
    step = torch.tensor([7], dtype=torch.int64, device="cuda")
    c = ops.sample(big, T, seed=99, step=step).cpu()
    assert not torch.equal(a, c)
    # temperature -> 0 approaches greedy
    cold = ops.sample(big[:64], 0.01, seed=5).cpu()
    assert (cold == int(logits.float().argmax())).all()
    with pytest.raises(Exception):
        ops.sample(big, 0.0, seed=1)


def test_sampling_in_the_lm_head_epilogue_matches_softmax_distribution(ops):
    """do_sample fused into the lm_head GEMM (umv_gemm_args.sample_temperature: Gumbel-max over bf16(logit / T), bagel.py:1297-1299):
    the maximum key of a row must be distributed as softmax(logits / T).  64 rows x 64 steps = 4096 draws from ONE logits vector
    (x = e_0, so logits = W[:, 0]) against the softmax with a chi-square bound; the same (seed, step) gives the same draw, another step
    another one; temperature -> 0 is the greedy token; the logits written to `out` are those of the plain GEMM."""
    import numpy as np
    V, K, T, rows = 40, 32, 0.7, 64
    g = torch.Generator().manual_seed(81)
    w = torch.zeros((V, K))
    w[:, 0] = torch.randn(V, generator=g) * 2
    w = w.to(BF16).cuda()
    lin = ops.PackedLinear.from_weight(w)
    x = torch.zeros((rows, K), dtype=BF16, device="cuda")
    x[:, 0] = 1.0
    logits = w[:, 0].float().cpu()
    p = torch.softmax((logits / T).to(BF16).float(), -1)
    keys = torch.zeros((rows, (V + 15) // 16), dtype=torch.int64, device="cuda")
    out = torch.empty((rows, V), dtype=BF16, device="cuda")
    step = torch.zeros(1, dtype=torch.int64, device="cuda")

    def draw(seed, s, temp=T):
        step.fill_(s)
        keys.zero_()
        ops.gemm(x, lin, out=out, argmax_partial=keys, sample=(temp, seed, step))
        k = keys.cpu().numpy().view(np.uint64).max(axis=1)
        return torch.from_numpy((np.uint64(0xFFFFFFFF) - (k & np.uint64(0xFFFFFFFF))).astype(np.int64))
    counts = torch.zeros(V)
    for s in range(64):
        ids = draw(1234, s)
        assert int(ids.min()) >= 0 and int(ids.max()) < V
        counts += torch.bincount(ids, minlength=V).float()
    assert torch.equal(out.float().cpu(), logits.to(BF16).float().expand(rows, V)), "the logits themselves must be the plain GEMM's"
    n = counts.sum()
    exp = p * n
    mask = exp > 5
    chi2 = (((counts - exp) ** 2) / exp)[mask].sum().item()
    dof = int(mask.sum()) - 1
    assert chi2 < dof + 6 * (2 * dof) ** 0.5, f"chi2 {chi2:.1f} for {dof} dof"
    a, b, c = draw(99, 3), draw(99, 3), draw(99, 4)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert len(set(a.tolist())) > 1, "rows must draw independently"
    assert (draw(5, 0, temp=0.01) == int(logits.argmax())).all()
    greedy = torch.zeros_like(keys)
    ops.gemm(x, lin, out=out, argmax_partial=greedy)
    kg = greedy.cpu().numpy().view(np.uint64).max(axis=1)
    assert ((np.uint64(0xFFFFFFFF) - (kg & np.uint64(0xFFFFFFFF))).astype(np.int64) == int(logits.argmax())).all()
    with pytest.raises(Exception):
        ops.gemm(x, lin, out=out, sample=(T, 1, step))          # sampling without the key buffer


def _rope_tables(max_pos, hd, theta=1e6):
    from oracle.unimedvl_cpu import rope_cos_sin
    return rope_cos_sin(torch.arange(max_pos), hd, theta, BF16)


@pytest.mark.parametrize("gen", [False, True])
def test_qkv_post(ops, gen):
    from oracle.unimedvl_cpu import rmsnorm, apply_rope
    nq, nkv, hd, T, cap = 4, 2, 128, 11, 64
    qkv = rnd((T, (nq + 2 * nkv) * hd), 40)
    qn, kn, qg, kg = (rnd((hd,), 41 + i) + 1 for i in range(4))
    cos, sin = _rope_tables(4096, hd)
    pos = torch.tensor([0, 1, 2, 3, 3, 3, 3, 3, 1000, 1001, 4000], dtype=torch.int32)
    seg = torch.tensor([0] * 6 + [1] * 5, dtype=torch.int32)
    slot = torch.tensor([3, 4, 5, 6, 7, 8, 0, 1, 2, 3, 4], dtype=torch.int32)
    expert = (torch.arange(T) % 2).to(torch.int32) if gen else None
    slab = ops.KVSlab(2, nkv, cap, hd, "cuda")
    q_out = torch.empty((T, nq, hd), dtype=BF16, device="cuda")
    ops.qkv_post(qkv.cuda(), q_out, slab, seg.cuda(), slot.cuda(), pos.cuda(), nq, nkv, hd, 1e-6, qn.cuda(), kn.cuda(),
                 qg.cuda(), kg.cuda(), None if expert is None else expert.cuda(), cos.cuda(), sin.cuda(), fp32_chain=gen)
    q = qkv[:, :nq * hd].view(T, nq, hd)
    k = qkv[:, nq * hd:(nq + nkv) * hd].view(T, nkv, hd)
    v = qkv[:, (nq + nkv) * hd:].view(T, nkv, hd)
    c, s = cos[pos.long()], sin[pos.long()]
    if not gen:
        qr, kr = apply_rope(rmsnorm(q, qn, 1e-6), rmsnorm(k, kn, 1e-6), c, s)
    else:
        e = expert.bool()
        qf, kf = q.float(), k.float()
        qf[~e] = rmsnorm(qf[~e], qn, 1e-6); qf[e] = rmsnorm(qf[e], qg, 1e-6)
        kf[~e] = rmsnorm(kf[~e], kn, 1e-6); kf[e] = rmsnorm(kf[e], kg, 1e-6)
        # text tokens of a gen-mode call take the fp32 chain too (qwen2_navit.py:568-579)
        qr, kr = apply_rope(qf, kf, c, s)
    assert_close_bf16(q_out, qr.to(BF16), what="q", frac_exact=0.99)
    kc, vtc = slab.k.cpu(), slab.vt.cpu()
    for t in range(T):
        assert_close_bf16(kc[seg[t], :, slot[t]], kr[t].to(BF16), what=f"k[{t}]", frac_exact=0.97)
        assert torch.equal(vtc[seg[t], :, :, slot[t]], v[t])


def _attn_case(ops, nq, nkv, hd, q_lens, k_lens, causal, nsplit=1, seed=50):
    from oracle.unimedvl_cpu import attention_segment
    nseg = len(q_lens)
    cap = (max(k_lens) + 31) // 32 * 32
    slab = ops.KVSlab(nseg, nkv, cap, hd, "cuda")
    T = sum(q_lens)
    q = rnd((T, nq, hd), seed)
    ks = [rnd((lk, nkv, hd), seed + 1 + i) for i, lk in enumerate(k_lens)]
    vs = [rnd((lk, nkv, hd), seed + 100 + i) for i, lk in enumerate(k_lens)]
    kh, vh = slab.k.cpu(), slab.vt.cpu()
    kh += float("nan") if False else 0
    for i, lk in enumerate(k_lens):
        kh[i, :, :lk] = ks[i].transpose(0, 1)
        vh[i, :, :, :lk] = vs[i].permute(1, 2, 0)
    slab.k.copy_(kh); slab.vt.copy_(vh)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32)
    out = torch.zeros((T, nq, hd), dtype=BF16, device="cuda")
    ws = ops.attn_workspace(nseg, nq, hd, max(q_lens), nsplit, "cuda") if nsplit > 1 else None
    ops.attention(q.cuda(), out, slab, cu.cuda(), torch.tensor(k_lens, dtype=torch.int32).cuda(), nq, nkv, hd, causal,
                  max(q_lens), max(k_lens), nsplit, ws)
    ref = torch.empty_like(q)
    t0 = 0
    for i, lq in enumerate(q_lens):
        ref[t0:t0 + lq] = attention_segment(q[t0:t0 + lq], ks[i], vs[i], causal, impl="flash")
        t0 += lq
    got = out.cpu()
    err = (got.float() - ref.float()).abs().max().item()
    assert err < 0.03, f"attention max abs err {err}"
    assert_close_bf16(got, ref, max_ulp=4, frac_exact=0.5, what="attention")


def test_attention_prefill(ops):
    _attn_case(ops, 28, 4, 128, [34, 130], [34 + 50, 130], causal=True)
    _attn_case(ops, 28, 4, 128, [70, 9], [70, 9 + 40], causal=False)
    _attn_case(ops, 2, 1, 128, [14, 33], [14, 33 + 14], causal=True)
    _attn_case(ops, 16, 16, 72, [100, 37, 16], [100, 37, 16], causal=False)
    _attn_case(ops, 2, 2, 72, [12], [12], causal=False)


def test_attention_decode_split(ops):
    _attn_case(ops, 28, 4, 128, [1] * 8, [1061, 1070, 33, 1, 500, 777, 1572, 64], causal=True, nsplit=1)
    _attn_case(ops, 28, 4, 128, [1] * 8, [1061, 1070, 33, 1, 500, 777, 1572, 64], causal=True, nsplit=8)
    _attn_case(ops, 2, 1, 128, [1, 1], [20, 45], causal=True, nsplit=4)


def test_f2bf_exhaustive(ops):
    """The fp32 -> bf16 rounding every kernel uses (common.h f2bf / pack2bf on v_cvt_pk_bf16_f32) against torch's own
    conversion over ALL 2^32 fp32 bit patterns (through umv_cast_pad_f32_bf16): identical bits for every non-NaN input
    (round to nearest even, overflow to inf, denormals kept), NaN in -> NaN out."""
    chunk = 1 << 26
    base = torch.arange(chunk, dtype=torch.int64, device="cuda")
    for c in range((1 << 32) // chunk):
        bits = (base + c * chunk).to(torch.int32) if c * chunk < (1 << 31) else (base + c * chunk - (1 << 32)).to(torch.int32)
        x = bits.view(torch.float32).view(chunk // 4096, 4096)
        got = ops.cast_pad(x, 4096).view(torch.int16)
        ref = x.to(torch.bfloat16).view(torch.int16)
        nan = torch.isnan(x)
        assert torch.equal(got[~nan], ref[~nan]), f"chunk {c}"
        assert torch.isnan(got.view(torch.bfloat16)[nan].float()).all()


@pytest.mark.parametrize("nq,nkv,hd", [(16, 16, 72), (1, 1, 512), (4, 2, 128), (3, 3, 40), (2, 1, 20)])
def test_qkv_split_without_norm(ops, nq, nkv, hd):
    """No-norm split (ViT / VAE attention, siglip_navit.py:222-231, autoencoder.py:51-60): q rows, K rows and V^T columns are
    plain copies.  The 8-token tile path must handle aligned runs, a run starting at a slot that is not a multiple of 8,
    a segment change inside a group of 8 and a ragged tail; hd = 20 (not a multiple of 8) takes the element kernel."""
    g = torch.Generator().manual_seed(nq * 100 + hd)
    lens, starts = [24, 13, 16, 5], [0, 3, 8, 40]          # tokens per segment and the slot their run starts at
    T, cap = sum(lens), 64
    seg = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)])
    slot = torch.cat([torch.arange(st, st + n, dtype=torch.int32) for st, n in zip(starts, lens)])
    N = (nq + 2 * nkv) * hd
    qkv = torch.randn(T, N, generator=g).to(BF16)
    slab = ops.KVSlab(len(lens), nkv, cap, hd, "cuda")
    slab.k.fill_(7.0)
    slab.vt.fill_(7.0)
    q_out = torch.zeros((T, nq, hd), dtype=BF16, device="cuda")
    ops.qkv_post(qkv.cuda(), q_out, slab, seg.cuda(), slot.cuda(), None, nq, nkv, hd)
    assert torch.equal(q_out.cpu(), qkv[:, :nq * hd].view(T, nq, hd))
    k_ref = torch.full((len(lens), nkv, cap, hd), 7.0, dtype=BF16)
    v_ref = torch.full((len(lens), nkv, hd, cap), 7.0, dtype=BF16)
    k = qkv[:, nq * hd:(nq + nkv) * hd].view(T, nkv, hd)
    v = qkv[:, (nq + nkv) * hd:].view(T, nkv, hd)
    for t in range(T):
        k_ref[seg[t], :, slot[t]] = k[t]
        v_ref[seg[t], :, :, slot[t]] = v[t]
    assert torch.equal(slab.k.cpu(), k_ref)          # untouched slots keep their content
    assert torch.equal(slab.vt.cpu(), v_ref)


@pytest.mark.parametrize("gen", [False, True])
def test_qkv_post_many_tokens_matches_small_calls(ops, gen):
    """T >= 64 tokens of a bf16 row: the V heads go through the 8-token tile kernel (16-byte V^T stores) and the norm + RoPE
    kernel visits only q / k heads.  Must equal, bit for bit, the same tokens pushed through in chunks of < 64 (one kernel)."""
    nq, nkv, hd, cap = 28, 4, 128, 160
    lens, starts = [70, 37, 9], [0, 5, 130]
    T = sum(lens)
    seg = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(lens)]).cuda()
    slot = torch.cat([torch.arange(st, st + n, dtype=torch.int32) for st, n in zip(starts, lens)]).cuda()
    pos = (slot + 11).to(torch.int32)
    qkv = rnd((T, (nq + 2 * nkv) * hd), 77).cuda()
    qn, kn, qg, kg = ((rnd((hd,), 78 + i) + 1).cuda() for i in range(4))
    cos, sin = _rope_tables(512, hd)
    cos, sin = cos.cuda(), sin.cuda()
    expert = (torch.arange(T) % 3 == 0).to(torch.int32).cuda() if gen else None

    def run(chunks):
        slab = ops.KVSlab(len(lens), nkv, cap, hd, "cuda")
        slab.k.fill_(3.0)
        slab.vt.fill_(3.0)
        q_out = torch.zeros((T, nq, hd), dtype=BF16, device="cuda")
        for a, b in chunks:
            ops.qkv_post(qkv[a:b], q_out[a:b], slab, seg[a:b], slot[a:b], pos[a:b], nq, nkv, hd, 1e-6, qn, kn, qg, kg,
                         None if expert is None else expert[a:b], cos, sin, fp32_chain=gen)
        return q_out, slab.k.clone(), slab.vt.clone()
    whole = run([(0, T)])
    parts = run([(0, 50), (50, 100), (100, T)])
    for x, y in zip(whole, parts):
        assert torch.equal(x, y)
    written = (whole[2] != 3.0).sum().item()      # a few random values may equal the fill by chance
    assert T * nkv * hd - 64 <= written <= T * nkv * hd


@pytest.mark.parametrize("M,N,K,fp8", [(8, 152064, 3584, False), (1, 1000, 256, False), (33, 4112, 512, False), (64, 37888, 1152, False),
                                       (8, 152064, 3584, True), (20, 1000, 256, True)])
def test_gemm_argmax_epilogue(ops, M, N, K, fp8):
    """greedy argmax as the lm_head epilogue (bagel.py:1295-1301): the per-tile keys + umv_decode_step_end_argmax pick
    torch.argmax of the stored bf16 logits, lowest index on ties (two identical weight rows -> identical logits), and do
    decode_step_end's bookkeeping."""
    x, w = rnd((M, K), 90), rnd((N, K), 91, 1 / math.sqrt(K))
    w[N // 3] = w[N - 5]                       # exact ties between two far-apart columns
    w[7] = w[N - 5]
    mk = ops.PackedLinear.from_weight_fp8 if fp8 else ops.PackedLinear.from_weight
    lin = mk(w.cuda())
    part = torch.zeros((M, (N + 15) // 16), dtype=torch.int64, device="cuda")
    logits = ops.gemm(x.cuda(), lin, argmax_partial=part)
    plain = ops.gemm(x.cuda(), lin)
    if M <= 32 or N < 16384:
        assert torch.equal(logits, plain), "the argmax epilogue must not change the logits"
    else:    # 33..64 rows on a wide-N linear: without the argmax keys the call goes to the LDS-staged 128x64 tile (other fp32
        assert_close_bf16(logits, plain, max_ulp=1, frac_exact=0.98, what="argmax-epilogue kernel vs tiled kernel")   # summation order)
    B, max_len = M, 5
    i32 = dict(dtype=torch.int32, device="cuda")
    slot, pos, kvl = torch.arange(B, **i32), torch.arange(B, **i32) + 100, torch.arange(B, **i32) + 1
    ids = torch.full((B,), -1, dtype=torch.int64, device="cuda")
    in_ids = torch.zeros((max_len, B), dtype=torch.int64, device="cuda")
    pred = torch.zeros((max_len, B), dtype=torch.int64, device="cuda")
    step = torch.full((B,), 2, dtype=torch.int64, device="cuda")          # one counter per sample
    ops.decode_step_end_argmax(slot, pos, kvl, part, ids, in_ids, pred, step)
    ref = torch.argmax(logits.float(), -1)
    assert torch.equal(ids, ref), (ids.tolist()[:8], ref.tolist()[:8])
    tie_rows = (logits[:, 7] == logits.float().max(-1).values.to(logits.dtype))
    assert torch.equal(ids[tie_rows], torch.full_like(ids[tie_rows], 7))
    assert torch.equal(pred[2], ref) and torch.equal(in_ids[3], ref) and int(pred[3].abs().sum()) == 0
    assert torch.equal(step, torch.full_like(step, 3))
    assert torch.equal(slot, torch.arange(B, **i32) + 1) and torch.equal(pos, torch.arange(B, **i32) + 101)
    # a second step through the same buffers
    ops.decode_step_end_argmax(slot, pos, kvl, part, ids, in_ids, pred, step)
    assert torch.equal(step, torch.full_like(step, 4)) and torch.equal(pred[3], ref) and torch.equal(in_ids[4], ref)
    with pytest.raises(Exception):
        ops.gemm(rnd((100, K), 92).cuda(), lin if not fp8 else ops.PackedLinear.from_weight(w.cuda()),
                 argmax_partial=torch.zeros((100, (N + 15) // 16), dtype=torch.int64, device="cuda"))


def test_timestep_embed(ops):
    """umv_timestep_embed against torch's own restatement of modeling_utils.py:87-109 on the flow schedule's timesteps
    (shifted linspace, bagel.py:937-940) and on t = 0 (forward_cache_update_vae): same bf16 bits except where libm's and torch's
    cos / sin differ in the last fp32 ulp right at a bf16 rounding boundary (<= 1 bf16 ulp, a handful of elements)."""
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    ts = torch.linspace(1, 0, 50)
    ts = (3.0 * ts / (1 + 2.0 * ts))[:-1]
    t = torch.cat([ts, torch.tensor([0.0, 1.0, 0.5, 1000.0, 999.0, 37.25])])
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(BF16)
    got = ops.timestep_embed(t.cuda(), freqs.cuda())
    assert got.shape == ref.shape == (t.numel(), 256)
    assert_close_bf16(got, ref, max_ulp=1, frac_exact=0.999, what="timestep sinusoid")
    assert torch.equal(got[49].cpu(), ref[49])          # t = 0: cos = 1, sin = 0 exactly


def _splitmix64(x):
    import numpy as np
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def test_sampling_uniform_extremes_cannot_pick_a_token_by_themselves(ops):
    """with u in (0, 1] the draw u == 1 (probability 2^-24 per column, ~1 % per draw over a 152 k vocabulary) gave +inf Gumbel
    noise / a zero exponential: that column won WHATEVER its logit.  u is now (r + 0.5) 2^-23, r = the top 23 bits of the hash.  The test
    searches the counter-based generator's stream (splitmix64 keyed by seed, step, row, column - restated here) for (seed, column) pairs
    whose r is all ones (the draw nearest 1) resp. zero (nearest 0), gives that column a logit 30 below the rest (p ~ 1e-13 after the
    softmax) and -inf, and checks that neither sampler picks it - umv_sample_bf16 and the lm_head epilogue draw from the same stream."""
    import numpy as np
    V, K = 4096, 32
    cols = np.arange(V, dtype=np.uint64)
    found = {}
    with np.errstate(over="ignore"):
        for seed in range(1, 20000):
            row_key = _splitmix64(np.uint64(seed))            # step 0, row 0: seed ^ 0 ^ 0
            r = _splitmix64(row_key + cols) >> np.uint64(41)
            for want, name in ((0x7FFFFF, "top"), (0, "bottom")):
                hit = np.nonzero(r == np.uint64(want))[0]
                if len(hit) and name not in found:
                    found[name] = (seed, int(hit[0]))
            if len(found) == 2:
                break
    assert len(found) == 2, found
    for name, (seed, col) in found.items():
        for low in (-30.0, float("-inf")):
            logits = torch.zeros((1, V))
            logits[0, col] = low
            lg = logits.to(BF16).cuda()
            tok = int(ops.sample(lg, 1.0, seed=seed)[0])
            assert tok != col, f"umv_sample_bf16 picked column {col} (logit {low}) on the '{name}' extreme of the uniform"
            # the same through the lm_head epilogue: logits = W[:, 0] with x = e_0
            w = torch.zeros((V, K))
            w[:, 0] = logits[0].clamp_min(-3e38)
            lin = ops.PackedLinear.from_weight(w.to(BF16).cuda())
            x = torch.zeros((1, K), dtype=BF16, device="cuda")
            x[0, 0] = 1.0
            keys = torch.zeros((1, (V + 15) // 16), dtype=torch.int64, device="cuda")
            out = torch.empty((1, V), dtype=BF16, device="cuda")
            ops.gemm(x, lin, out=out, argmax_partial=keys, sample=(1.0, seed, None))
            k = keys.cpu().numpy().view(np.uint64).max(axis=1)
            tok2 = int(np.uint64(0xFFFFFFFF) - (k[0] & np.uint64(0xFFFFFFFF)))
            assert tok2 != col, f"the lm_head epilogue picked column {col} (logit {low}) on the '{name}' extreme of the uniform"
            if name == "top" and low == -30.0:
                assert tok == tok2, "both samplers draw from one stream: same seed, same token"
