"""SURVEY 8f-4 as built: GROWABLE KV slabs instead of block tables.  What a paged cache must guarantee - a context
can outgrow any reservation, nothing already cached changes, attention over the grown cache is right - is pinned here at 32 k tokens:
  * a 32 768-token context prefilled in 4 096-token chunks into a cache that starts at 256 tokens (seven doublings on the way) leaves
    K / V^T slabs and next-token logits BIT-IDENTICAL to the same prefill into a cache reserved at full size up front;
  * the attention kernels over 32 k keys (decode: one query row, split-KV + combine; prefill: 4 096 causal rows on top of 28 k cached
    keys, the LDS-shared kernel with its 32-bit buffer offsets) against exact fp64 attention on sampled rows.
The reference's NaiveCache has no bound either: it re-merges the whole cache every forward (qwen2_navit.py:585-600, inferencer.py:261)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def test_growth_to_32k_changes_nothing(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from conftest import NEW_TOKEN_IDS
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    cfg_d, sd, _, _ = tiny_weights
    cfg = UniMedVLConfig.from_dict(cfg_d)
    cfg.max_position = 36864                                   # rotary tables for 32 k tokens of context + the decoded ones
    model = Bagel(cfg, lambda n: sd[n], device="cuda:0", visual_gen=False, visual_und=False)
    total, chunk = 32768, 4096
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 290, (total,), generator=g).tolist()

    class Tok:
        def __init__(self, t):
            self.t = t

        def encode(self, s):
            return self.t[int(s)]

    def prefill(cache):
        kvl, rope, caps = [0], [0], []
        for c0 in range(0, total, chunk):
            piece = ids[c0:c0 + chunk - 2]                    # + bos / eos = `chunk` tokens per call
            gi, kvl, rope = model.prepare_prompts(kvl, rope, ["0"], Tok([piece]), NEW_TOKEN_IDS)
            cache = model.forward_cache_update_text(cache, **gi)
            caps.append(cache.cap)
        gs = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
        _, logits = model.generate_text(past_key_values=cache, max_length=2, return_logits=True, **gs)
        return cache, kvl, caps, logits

    grown, kvl, caps, lg = prefill(NaiveCache(cfg.layers))
    assert kvl == [total]
    assert len(set(caps)) >= 4, f"the cache was meant to double on the way: capacities {caps}"
    fixed = NaiveCache(cfg.layers)
    fixed.reserve(1, total + 64, cfg.kv_heads, cfg.head_dim, model.device)
    fixed, kvl2, caps2, lg2 = prefill(fixed)
    assert len(set(caps2)) == 1
    for l in range(cfg.layers):
        assert torch.equal(grown.slabs[l].k[:, :, :total], fixed.slabs[l].k[:, :, :total]), f"layer {l}: K differs after growth"
        assert torch.equal(grown.slabs[l].vt[:, :, :, :total], fixed.slabs[l].vt[:, :, :, :total]), f"layer {l}: V^T differs after growth"
    assert torch.equal(lg[0], lg2[0]), "next-token logits differ between the grown and the reserved cache"
    assert torch.isfinite(lg[0].float()).all()


def _exact_rows(q, k, v, rows, limit_of):
    """fp64 attention of the sampled query rows: q [T, nq, hd], k / v [Lk, nkv, hd]; row t sees keys <= limit_of(t)"""
    rep = q.shape[1] // k.shape[1]
    kf = k.double().transpose(0, 1).repeat_interleave(rep, 0)           # [nq, Lk, hd]
    vf = v.double().transpose(0, 1).repeat_interleave(rep, 0)
    out = []
    for t in rows:
        s = torch.einsum("hd,hkd->hk", q[t].double(), kf) / math.sqrt(q.shape[-1])
        s[:, limit_of(t) + 1:] = float("-inf")
        out.append(torch.einsum("hk,hkd->hd", torch.softmax(s, -1), vf))
    return torch.stack(out)


@pytest.mark.parametrize("scale", [1.0, 8.0])
def test_attention_over_32k_keys(scale):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import _lib, ops
    nq, nkv, hd, Lk, Lq = 28, 4, 128, 32768, 4096
    g = torch.Generator(device="cuda").manual_seed(11)
    k = torch.randn(Lk, nkv, hd, generator=g, device="cuda").to(BF16)
    v = torch.randn(Lk, nkv, hd, generator=g, device="cuda").to(BF16)
    slab = ops.KVSlab(1, nkv, Lk, hd, "cuda")
    slab.k[0] = k.transpose(0, 1)
    slab.vt[0] = v.permute(1, 2, 0)
    kvl = torch.tensor([Lk], dtype=torch.int32).cuda()
    # prefill: the last 4 096 tokens of the context attend causally over everything before them (28 k cached keys + themselves)
    q = (torch.randn(Lq, nq, hd, generator=g, device="cuda") * scale).to(BF16)
    cu = torch.tensor([0, Lq], dtype=torch.int32).cuda()
    assert _lib.load().umv_attn_prefill_tq(1, nq, nkv, hd, Lq) in (1, 2)
    out = torch.zeros_like(q)
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, True, Lq, Lk)
    rows = [0, 1, 777, 2048, 4094, 4095]
    ref = _exact_rows(q, k, v, rows, lambda t: Lk - Lq + t)
    err = (out[rows].double() - ref).abs().max().item()
    ulp = 2.0 ** (math.floor(math.log2(float(ref.abs().max()))) - 7)
    assert err <= 2 * ulp, f"prefill over 32 k keys: max error {err:.4g} (2 ulp of the output range = {2 * ulp:.4g})"
    # decode: one query row over the 32 k keys, split 16 ways + combine
    q1 = q[-1:].contiguous()
    ws = torch.zeros(_lib.load().umv_attn_workspace_bytes(1, nq, hd, 1, 16) // 4, dtype=torch.float32, device="cuda")
    o1 = torch.zeros_like(q1)
    ops.attention(q1, o1, slab, torch.tensor([0, 1], dtype=torch.int32).cuda(), kvl, nq, nkv, hd, True, 1, Lk, nsplit=16, workspace=ws)
    ref1 = _exact_rows(q1, k, v, [0], lambda t: Lk - 1)
    err1 = (o1.double() - ref1).abs().max().item()
    assert err1 <= 2 * ulp, f"decode over 32 k keys: max error {err1:.4g}"
