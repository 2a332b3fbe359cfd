"""Paged KV (block tables instead of the reference's re-merged cache, qwen2_navit.py:585-600): 256-token pages in
one pool per layer, a page table per segment (include/unimedvl_hip.h: umv_attn_args / umv_qkv_post_args .page_table).
  * kernels: attention through a page table (pages handed out in SCRAMBLED order) is bit-identical to the slab form on the same keys -
    decode (split-KV + combine, with and without the in-workgroup wave split) and prefill (causal, ragged) - and umv_qkv_post writes K / V^T
    through the table to the same values the slab form holds;
  * cache: prefix snapshots share pages and copy the partially filled last page on the first append (either side);
  * engine: prefill + captured greedy decode on a PagedCache give the same tokens and logits as on NaiveCache slabs;
  * serving: ContinuousBatcher(paged=True) - 7 requests of different lengths through 3 slots, pages returned and re-used - gives the
    answers of the slab batcher, which the reference-pinned tests hold to one-request-at-a-time decoding."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops as o
    return o


def _paged_copy(ops, slab, lens, seed):
    """the keys of a KVSlab re-housed in page pools, the pages of each segment taken in a shuffled order"""
    nseg, nkv, cap, hd = slab.k.shape
    P = ops.KV_PAGE
    need = [(n + P - 1) // P for n in lens]
    npages = sum(need) + 3
    g = torch.Generator().manual_seed(seed)
    order = (torch.randperm(npages - 1, generator=g) + 1).tolist()        # page 0 is never handed out
    table = torch.zeros((nseg, max(need) + 1), dtype=torch.int32, device="cuda")
    ps = ops.PagedSlab(npages, nkv, hd, "cuda", table)
    ps.k.fill_(777.0)
    ps.vt.fill_(777.0)
    it = iter(order)
    for s, n in enumerate(lens):
        for p in range(need[s]):
            pg = next(it)
            table[s, p] = pg
            m = min(P, cap - p * P)
            ps.k[pg, :, :m] = slab.k[s, :, p * P:p * P + m]
            ps.vt[pg, :, :, :m] = slab.vt[s, :, :, p * P:p * P + m]
    return ps


@pytest.mark.parametrize("wave_split", [0, 2, 4])
def test_paged_decode_attention_bit_identical(ops, wave_split):
    from unimedvl_amd import _lib
    nq, nkv, hd = 28, 4, 128
    lens = [1060, 777, 33, 300, 256, 513]
    nseg = len(lens)
    g = torch.Generator(device="cuda").manual_seed(3)
    slab = ops.KVSlab(nseg, nkv, 1088, hd, "cuda")
    slab.k.copy_(torch.randn(slab.k.shape, generator=g, device="cuda").to(BF16))
    slab.vt.copy_(torch.randn(slab.vt.shape, generator=g, device="cuda").to(BF16))
    q = (torch.randn(nseg, nq, hd, generator=g, device="cuda") * 3).to(BF16)
    cu = torch.arange(nseg + 1, dtype=torch.int32).cuda()
    kvl = torch.tensor(lens, dtype=torch.int32).cuda()
    ps = _paged_copy(ops, slab, lens, 9)
    for nsplit in (1, 4, 12):
        ws = None
        if nsplit > 1:
            ws = torch.zeros(_lib.load().umv_attn_workspace_bytes(nseg, nq, hd, 1, nsplit) // 4, dtype=torch.float32, device="cuda")
        ref = torch.zeros_like(q)
        ops.attention(q, ref, slab, cu, kvl, nq, nkv, hd, True, 1, max(lens), nsplit=nsplit, workspace=ws, wave_split=wave_split)
        out = torch.zeros_like(q)
        ops.attention(q, out, ps, cu, kvl, nq, nkv, hd, True, 1, max(lens), nsplit=nsplit, workspace=ws, wave_split=wave_split)
        assert torch.isfinite(out.float()).all()
        assert torch.equal(out, ref), f"nsplit {nsplit} wave_split {wave_split}: paged attention differs from the slab form"
    # the wave split changes the summation tree, not the answer: against the single-wave kernel within one bf16 ulp of the output range
    if wave_split:
        one = torch.zeros_like(q)
        ws = torch.zeros(_lib.load().umv_attn_workspace_bytes(nseg, nq, hd, 1, 4) // 4, dtype=torch.float32, device="cuda")
        ops.attention(q, one, slab, cu, kvl, nq, nkv, hd, True, 1, max(lens), nsplit=4, workspace=ws)
        two = torch.zeros_like(q)
        ops.attention(q, two, slab, cu, kvl, nq, nkv, hd, True, 1, max(lens), nsplit=4, workspace=ws, wave_split=wave_split)
        assert float((one.float() - two.float()).abs().max()) <= 2.0 ** -7 * float(one.float().abs().max())


def test_paged_prefill_attention_and_qkv_post(ops):
    """causal, ragged prefill on top of cached context: keys appended through the page table by umv_qkv_post, attention through it"""
    nq, nkv, hd = 28, 4, 128
    ctx, qlens = [300, 0, 513], [200, 129, 77]
    lens = [c + n for c, n in zip(ctx, qlens)]
    nseg, T = 3, sum(qlens)
    g = torch.Generator(device="cuda").manual_seed(4)
    slab = ops.KVSlab(nseg, nkv, 640, hd, "cuda")
    for s, c in enumerate(ctx):
        slab.k[s, :, :c] = torch.randn(nkv, c, hd, generator=g, device="cuda").to(BF16)
        slab.vt[s, :, :, :c] = torch.randn(nkv, hd, c, generator=g, device="cuda").to(BF16)
    ps = _paged_copy(ops, slab, lens, 10)          # pages for the FINAL lengths; the new tokens' slots still hold the fill value
    qkv = torch.randn(T, (nq + 2 * nkv) * hd, generator=g, device="cuda").to(BF16)
    seg = torch.cat([torch.full((n,), s, dtype=torch.int32) for s, n in enumerate(qlens)]).cuda()
    slot = torch.cat([torch.arange(c, c + n, dtype=torch.int32) for c, n in zip(ctx, qlens)]).cuda()
    pos = slot.clone()
    w = torch.ones(hd, dtype=BF16, device="cuda")
    cos = torch.randn(1024, hd, generator=g, device="cuda").to(BF16)
    sin = torch.randn(1024, hd, generator=g, device="cuda").to(BF16)
    q1, q2 = torch.zeros(T, nq, hd, dtype=BF16, device="cuda"), torch.zeros(T, nq, hd, dtype=BF16, device="cuda")
    ops.qkv_post(qkv, q1, slab, seg, slot, pos, nq, nkv, hd, 1e-6, w, w, cos_tab=cos, sin_tab=sin)
    ops.qkv_post(qkv, q2, ps, seg, slot, pos, nq, nkv, hd, 1e-6, w, w, cos_tab=cos, sin_tab=sin)
    assert torch.equal(q1, q2)
    P = ops.KV_PAGE
    for s, n in enumerate(lens):
        for p in range((n + P - 1) // P):
            m = min(P, n - p * P)
            pg = int(ps.table[s, p])
            assert torch.equal(ps.k[pg, :, :m], slab.k[s, :, p * P:p * P + m]), f"segment {s} page {p}: K"
            assert torch.equal(ps.vt[pg, :, :, :m], slab.vt[s, :, :, p * P:p * P + m]), f"segment {s} page {p}: V^T"
    cu = torch.tensor([0, 200, 329, 406], dtype=torch.int32).cuda()
    kvl = torch.tensor(lens, dtype=torch.int32).cuda()
    from unimedvl_amd import _lib as L
    # the library's own policy (LDS-shared lazy kernels, their PAGED instantiation), TQ = 1 and 2, and the per-wave kernel: paged == slab, bit for bit
    for variant in (0, L.ATTN_FORCE | L.ATTN_TQ1, L.ATTN_FORCE | L.ATTN_TQ2, L.ATTN_FORCE | L.ATTN_STREAM):
        ref = torch.zeros(T, nq, hd, dtype=BF16, device="cuda")
        ops.attention(q1, ref, slab, cu, kvl, nq, nkv, hd, True, max(qlens), max(lens), variant=variant)
        out = torch.zeros_like(ref)
        ops.attention(q1, out, ps, cu, kvl, nq, nkv, hd, True, max(qlens), max(lens), variant=variant)
        assert torch.isfinite(out.float()).all() and float(out.float().abs().max()) > 0
        assert torch.equal(out, ref), f"variant {variant}: paged prefill attention differs from the slab form"
    # a long, non-causal span over several pages (8 x 1026 image spans on top of 300 cached keys: TQ = 2 by policy)
    lens2 = [300 + 1026] * 8
    slab2 = ops.KVSlab(8, nkv, 1344, hd, "cuda")
    slab2.k.copy_(torch.randn(slab2.k.shape, generator=g, device="cuda").to(BF16))
    slab2.vt.copy_(torch.randn(slab2.vt.shape, generator=g, device="cuda").to(BF16))
    ps2 = _paged_copy(ops, slab2, lens2, 11)
    qq = torch.randn(8 * 1026, nq, hd, generator=g, device="cuda").to(BF16)
    cu2 = torch.arange(0, 9 * 1026, 1026, dtype=torch.int32).cuda()
    kvl2 = torch.tensor(lens2, dtype=torch.int32).cuda()
    assert L.load().umv_attn_prefill_tq(8, nq, nkv, hd, 1026) == 2
    ref = torch.zeros_like(qq)
    ops.attention(qq, ref, slab2, cu2, kvl2, nq, nkv, hd, False, 1026, 1326)
    out = torch.zeros_like(qq)
    ops.attention(qq, out, ps2, cu2, kvl2, nq, nkv, hd, False, 1026, 1326)
    assert torch.equal(out, ref)


def _tiny_model(tiny_weights):
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    cfg_d, sd, _, _ = tiny_weights
    cfg = UniMedVLConfig.from_dict(cfg_d)
    return cfg, Bagel(cfg, lambda n: sd[n], device="cuda:0", visual_gen=False, visual_und=True)


class _Tok:
    def __init__(self, table):
        self.table = table

    def encode(self, s):
        return self.table[int(s)]

    def decode(self, ids):
        return "<|im_start|>" + " ".join(str(int(v)) for v in ids[1:]) + "<|im_end|>"


def test_paged_cache_engine_and_snapshots(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from conftest import NEW_TOKEN_IDS
    from unimedvl_amd.kvcache import NaiveCache, PagedCache
    cfg, model = _tiny_model(tiny_weights)
    g = torch.Generator().manual_seed(21)
    prompts = [torch.randint(0, 290, (n,), generator=g).tolist() for n in (300, 41, 530)]
    tok = _Tok(prompts)

    def run(cache):
        gi, kvl, rope = model.prepare_prompts([0] * 3, [0] * 3, ["0", "1", "2"], tok, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_text(cache, **gi)
        snap = cache.snapshot()
        gs = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
        ids, logits = model.generate_text(past_key_values=cache, max_length=20, return_logits=True, **gs)
        return cache, snap, kvl, rope, ids, logits

    c1, s1, kvl, rope, ids1, lg1 = run(NaiveCache(cfg.layers))
    c2, s2, _, _, ids2, lg2 = run(PagedCache(cfg.layers, pool_pages=64, max_context=2048))
    assert torch.equal(ids1, ids2)
    for a, b in zip(lg1, lg2):
        assert torch.equal(a, b), "logits differ between the slab and the paged cache"
    for l in range(cfg.layers):
        assert torch.equal(c1.packed_keys(l), c2.packed_keys(l)) and torch.equal(c1.packed_values(l), c2.packed_values(l))
    # the snapshot still shows the prefix (the decode appended to copies of the shared last pages), and decoding from it repeats the run
    assert s2.lens == kvl
    for l in range(cfg.layers):
        assert torch.equal(s1.packed_keys(l), s2.packed_keys(l)) and torch.equal(s1.packed_values(l), s2.packed_values(l))
    # page accounting: 2 + 1 + 3 prefix pages; the decode appended to the shared, partially filled last page of each segment, which
    # copied that page (3 more) and left the originals to the snapshot - whose own decode then writes into pages it alone holds
    used = c2.pages_in_use()
    assert used == (2 + 1 + 3) + 3, used
    gs = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    ids3, _ = model.generate_text(past_key_values=s2, max_length=20, return_logits=True, **gs)
    assert torch.equal(ids3, ids1)
    assert c2.pages_in_use() == used
    for l in range(cfg.layers):          # ... and the first cache is untouched by the snapshot's decode
        assert torch.equal(c1.packed_keys(l), c2.packed_keys(l))
    for s in range(3):
        s2.release(s)
        c2.release(s)
    assert c2.pages_in_use() == 0


def test_paged_continuous_batcher_matches_slabs(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from conftest import NEW_TOKEN_IDS
    from unimedvl_amd.serving import ContinuousBatcher
    cfg, model = _tiny_model(tiny_weights)
    g = torch.Generator().manual_seed(22)
    lens = [30, 700, 12, 300, 1500, 64, 257]
    prompts = [torch.randint(0, 290, (n,), generator=g).tolist() for n in lens]
    budgets = [24, 40, 8, 33, 20, 48, 16]
    ntid = dict(NEW_TOKEN_IDS)
    ntid["eos_token_id"] = 301

    def serve(**kw):
        b = ContinuousBatcher(model, _Tok(prompts), ntid, lambda x: x, slots=3, max_context=256, max_new_tokens=16, check_every=8, **kw)
        rids = [b.submit(None, str(i), max_new_tokens=budgets[i]) for i in range(len(prompts))]
        out = b.run()
        return [out[r] for r in rids], b

    ref, bs = serve()
    got, bp = serve(paged=True, pool_pages=24)
    assert got == ref
    assert bs.stats["cache_grows"] >= 1 and bp.stats["cache_grows"] == 0, "the slab batcher had to re-allocate; the paged one must not"
    # 24 pages of 256 tokens serve requests of up to 1 500 + 20 tokens through 3 slots only because finished requests return their pages
    assert bp.cache.pages_in_use() <= 3
