"""Split-K decode GEMM (umv_gemm_args.k_splits) and its two consumers through the C ABI.

  * partial sums: sum_s P[s] against an fp32 torch GEMM of the same bf16 operands (rtol 1e-3 of the output scale: only the
    association of the fp32 sums differs), and bf16(sum + bias) against umv_gemm_bf16's output (<= 1 bf16 ulp, >= 99 % equal);
  * umv_residual_rmsnorm_bf16: the new residual stream must be BIT-EXACT bf16(bf16(p0 + p1 + ...) + seq) (sequential fp32
    adds in split order) and the normalised output bit-identical to umv_rmsnorm_bf16 of that stream;
  * umv_qkv_post with a partials input == umv_qkv_post on bf16(sum + bias), bit for bit (q rows, K and V^T slabs);
  * the decode session in split-K mode against the default session: same greedy tokens on the tiny model wherever the
    default session's top-2 logit margin exceeds 0.25, logits within 0.25 (the tolerance of every engine test)."""
import pytest
import torch

from conftest import NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops
    return ops


def _seq_sum(p):
    acc = p[0].clone()
    for s in range(1, p.shape[0]):
        acc = acc + p[s]
    return acc


@pytest.mark.parametrize("M", [1, 8, 16, 32, 64, 65, 96, 128])       # > 64 rows: the tiled 128 x 128 kernel over k_splits K ranges
@pytest.mark.parametrize("N,K,S", [(4608, 3584, 4), (3584, 3584, 4), (3584, 18944, 8), (320, 1000, 3), (48, 64, 2), (3584, 18944, 16)])
def test_splitk_partials(M, N, K, S):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(BF16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).cuda()
    b = torch.randn(N, generator=g).to(BF16).cuda()
    lin = ops.PackedLinear.from_weight(w, b)
    p = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_splitk(x, lin, p, S)
    total = _seq_sum(p)
    ref = x.float() @ w.float().t()
    scale = ref.abs().max().clamp_min(1e-3)
    assert torch.isfinite(p).all()
    assert (total - ref).abs().max() <= 1e-3 * scale
    got = (total + b.float()).to(BF16)
    one = ops.gemm(x, lin)
    # one bf16 ulp of the value, or the fp32 association noise where sum and bias cancel
    tol = torch.maximum(one.float().abs() * 2.0 ** -7, 1e-3 * scale)
    assert ((got.float() - one.float()).abs() <= tol).all() and (got == one).float().mean() >= 0.99


@pytest.mark.parametrize("M", [8, 32, 64])
@pytest.mark.parametrize("N,K,S", [(4608, 3584, 3), (3584, 18944, 4), (320, 1000, 3)])
def test_splitk_partials_fp8_weights(M, N, K, S):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K + 7)
    x = torch.randn(M, K, generator=g).to(BF16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).cuda()
    lin = ops.PackedLinear.from_weight_fp8(w)
    p = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_splitk(x, lin, p, S)
    # reference: the bf16 split-K GEMM on the dequantised weights (same operands, possibly another K partition)
    lin16 = ops.PackedLinear(lin.wp, None, lin.N, lin.K)
    p16 = torch.empty_like(p)
    ops.gemm_splitk(x, lin16, p16, S)
    scale = _seq_sum(p16).abs().max().clamp_min(1e-3)
    assert torch.isfinite(p).all() and (_seq_sum(p) - _seq_sum(p16)).abs().max() <= 1e-3 * scale


@pytest.mark.parametrize("T,H,S", [(1, 3584, 4), (8, 3584, 8), (32, 3584, 8), (5, 256, 2), (64, 4096, 3)])
def test_residual_rmsnorm(T, H, S):
    ops = _ops()
    g = torch.Generator().manual_seed(T + H + S)
    p = torch.randn(S, T, H, generator=g).cuda()
    seq = torch.randn(T, H, generator=g).to(BF16).cuda()
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(BF16).cuda()
    want_seq = (_seq_sum(p).to(BF16).float() + seq.float()).to(BF16)
    out = torch.empty_like(seq)
    seq2 = seq.clone()
    ops.residual_rmsnorm(p, seq2, w, 1e-6, out)
    assert torch.equal(seq2, want_seq)
    ref = ops.rmsnorm(want_seq, w, 1e-6)
    if H >= 1024:      # same thread mapping and reduction order as the decode rmsnorm kernel
        assert torch.equal(out, ref)
    else:              # the generic rmsnorm kernel reduces the row in another order: <= 1 bf16 ulp
        assert (out.view(torch.int16).int() - ref.view(torch.int16).int()).abs().max() <= 1


def test_qkv_post_from_partials():
    ops = _ops()
    nq, nkv, hd, T, S = 28, 4, 128, 8, 4
    g = torch.Generator().manual_seed(3)
    N = (nq + 2 * nkv) * hd
    p = torch.randn(S, T, N, generator=g).cuda()
    bias = torch.randn(N, generator=g).to(BF16).cuda()
    qkv = (_seq_sum(p) + bias.float()).to(BF16)
    qn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    kn = (1 + 0.1 * torch.randn(hd, generator=g)).to(BF16).cuda()
    pos = torch.arange(100, 100 + T, dtype=torch.int32).cuda()
    ang = torch.rand(256, hd, generator=g) * 6.28
    cos, sin = ang.cos().to(BF16).cuda(), ang.sin().to(BF16).cuda()
    seg = torch.arange(T, dtype=torch.int32).cuda()
    slot = torch.full((T,), 5, dtype=torch.int32).cuda()
    outs = []
    for use_p in (False, True):
        slab = ops.KVSlab(T, nkv, 32, hd, "cuda")
        slab.k.zero_()
        slab.vt.zero_()
        q = torch.zeros(T, nq, hd, dtype=BF16, device="cuda")
        if use_p:
            ops.qkv_post(None, q, slab, seg, slot, pos, nq, nkv, hd, 1e-6, qn, kn, cos_tab=cos, sin_tab=sin, partials=p, bias=bias)
        else:
            ops.qkv_post(qkv, q, slab, seg, slot, pos, nq, nkv, hd, 1e-6, qn, kn, cos_tab=cos, sin_tab=sin)
        outs.append((q, slab.k.clone(), slab.vt.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][0].abs().sum() > 0 and outs[0][1].abs().sum() > 0 and outs[0][2].abs().sum() > 0


def test_decode_session_splitk_matches_default(tiny_weights, monkeypatch):
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.decode import DecodeSession
    from unimedvl_amd.kvcache import NaiveCache
    cfg, sd, _, _ = tiny_weights
    model = Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda", visual_gen=False)
    B = 5
    g = torch.Generator().manual_seed(9)
    prompts = [[int(v) for v in torch.randint(5, 290, (3 + i,), generator=g)] for i in range(B)]

    class Tok:
        def encode(self, s):
            return prompts[int(s)]

    runs = {}
    modes = ("0", "2,2,3", "3,1,2", "2,1,3")   # every GEMM split or not (the QKV partial sums go into umv_qkv_post)
    for mode in modes:
        monkeypatch.setenv("UMV_DECODE_SPLITK", mode)
        cache = NaiveCache(cfg["layers"])
        gi, kvl, rope = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], Tok(), NEW_TOKEN_IDS)
        cache = model.forward_cache_update_text(cache, **gi)
        gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
        for use_graph in (False, True):
            from copy import deepcopy
            sess = DecodeSession(model.language_model, deepcopy(cache), gi["packed_start_tokens"], gi["packed_query_position_ids"], 6,
                                 use_graph=use_graph)
            assert sess.sk == ((1, 1, 1) if mode == "0" else tuple(int(v) for v in mode.split(",")))
            logits = []
            for _ in range(5):
                sess.step(1)
                logits.append(sess.logits.float().clone())
            runs[(mode, use_graph)] = (sess.pred_ids[:5].clone(), torch.stack(logits))
    for mode in modes:   # graph replay == eager, bit for bit, in every mode
        assert torch.equal(runs[(mode, False)][0], runs[(mode, True)][0])
        assert torch.equal(runs[(mode, False)][1], runs[(mode, True)][1])
    ids0, lg0 = runs[("0", False)]
    for mode in modes[1:]:
        ids1, lg1 = runs[(mode, False)]
        for s in range(5):
            assert (lg0[s] - lg1[s]).abs().max() <= 0.25
            top2 = lg0[s].topk(2, dim=-1).values
            sure = (top2[:, 0] - top2[:, 1]) > 0.25
            assert torch.equal(ids0[s][sure], ids1[s][sure])
            if not torch.equal(ids0[s], ids1[s]):
                break


def test_decode_session_above_64_samples_matches_oracle(tiny_weights):
    """65..128 samples per step: split-K on the tiled kernel (6, 8, 8 splits), consumers summing 6 / 8 partials, separate argmax
    (the argmax epilogue is a <= 64-row feature); logits against the CPU oracle, teacher-forced, and graph == eager."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache, OracleBagel
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.decode import DecodeSession
    from unimedvl_amd.kvcache import NaiveCache
    cfg, sd, vae_sd, _ = tiny_weights
    model = Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda", visual_gen=False)
    oracle = OracleBagel(cfg, sd, vae_sd)
    B = 70
    g = torch.Generator().manual_seed(10)
    prompts = [[int(v) for v in torch.randint(5, 290, (2 + i % 7,), generator=g)] for i in range(B)]

    class Tok:
        def encode(self, s):
            return prompts[int(s)]
    bos, eos = NEW_TOKEN_IDS["bos_token_id"], NEW_TOKEN_IDS["eos_token_id"]
    cache = NaiveCache(cfg["layers"])
    gi, kvl, rope = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], Tok(), NEW_TOKEN_IDS)
    cache = model.forward_cache_update_text(cache, **gi)
    oc = KVCache(cfg["layers"], B)
    okv, orope = oracle.update_text(oc, [0] * B, [0] * B, [[bos] + p + [eos] for p in prompts])
    assert okv == kvl and orope == rope
    gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    runs = []
    for use_graph in (True, False):
        sess = DecodeSession(model.language_model, deepcopy(cache), gi["packed_start_tokens"], gi["packed_query_position_ids"], 5,
                             use_graph=use_graph)
        assert sess.sk == (6, 8, 8) and not sess.fused_argmax
        lg = []
        for _ in range(4):
            sess.step(1)
            lg.append(sess.logits.float().cpu())
        runs.append((sess.in_ids[:4].cpu(), torch.stack(lg)))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    pos = torch.tensor(rope, dtype=torch.long)
    for s in range(4):
        h = oracle.llm_forward(oracle.embed(runs[0][0][s]), [1] * B, pos, oc, True, True, "und")
        ref = oracle.lm_head(h).float()
        pos = pos + 1
        assert (runs[0][1][s] - ref).abs().max() <= 0.25, f"step {s}"
