"""Continuous batching (unimedvl_amd/serving.py): more requests than slots, ragged contexts, per-request EOS / token
budgets, slots refilled in flight.  Every request must get exactly the answer that one-request-at-a-time greedy decoding
(Bagel.chat, the mirror of bagel.py:1321-1392) gives it: samples are independent rows of every kernel, whichever slot and
neighbours a request lands on."""
import pytest
import torch

from conftest import NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    cfg, sd, _, _ = tiny_weights
    return Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda", visual_gen=False)


def _requests(n):
    g = torch.Generator().manual_seed(21)
    reqs = []
    for i in range(n):
        h, w = [(42, 56), (28, 70), (56, 56), (42, 42)][i % 4]
        images = [] if i % 5 == 4 else [torch.randn(3, h, w, generator=g).clamp(-1, 1)]
        if i % 7 == 3:
            images.append(torch.randn(3, 28, 28, generator=g).clamp(-1, 1))
        prompt = " ".join(str(int(v)) for v in torch.randint(5, 290, (2 + i % 6,), generator=g))
        reqs.append((images, prompt))
    return reqs


@pytest.mark.parametrize("slots,check_every,use_graph", [(3, 4, True), (8, 3, True), (2, 5, False)])
def test_continuous_batching_matches_single_requests(model, slots, check_every, use_graph):
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.serving import ContinuousBatcher
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    reqs = _requests(7)
    budgets = [6, 3, 6, 5, 6, 2, 6]
    ident = lambda x: x   # noqa: E731  (images are already [3,H,W] tensors in [-1,1])
    want = []
    for (images, prompt), nb in zip(reqs, budgets):
        want.append(model.chat(tok, NEW_TOKEN_IDS, ident, images, prompt, max_length=nb + 1))
    srv = ContinuousBatcher(model, tok, NEW_TOKEN_IDS, ident, slots=slots, max_context=256, max_new_tokens=8,
                            check_every=check_every, use_graph=use_graph)
    rids = [srv.submit(images, prompt, max_new_tokens=nb) for (images, prompt), nb in zip(reqs, budgets)]
    got = srv.run()
    assert sorted(got) == sorted(rids)
    for rid, w in zip(rids, want):
        assert got[rid] == w, (rid, got[rid], w)
    assert srv.stats["prefills"] == 7 and srv.stats["tokens"] == sum(len(tok.encode(w)) for w in want)


def test_batcher_rejects_oversized_requests(model):
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.serving import ContinuousBatcher
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    srv = ContinuousBatcher(model, tok, NEW_TOKEN_IDS, lambda x: x, slots=2, max_context=16, max_new_tokens=4, growable=False)
    with pytest.raises(ValueError, match="reserve"):
        srv.submit(None, "5 6", max_new_tokens=9)
    srv.submit(None, " ".join(["7"] * 40))
    with pytest.raises(ValueError, match="max_context"):
        srv.run()


@pytest.mark.parametrize("slots,check_every,use_graph", [(2, 3, True), (3, 4, True), (2, 5, False)])
def test_batcher_grows_its_cache_for_long_requests(model, slots, check_every, use_graph):
    """growable=True (the default; the reference's NaiveCache grows without bound, qwen2_navit.py:585-600): the slots are
    reserved for the minimum (256 tokens per slot); a 280-token request in the middle of the queue and a 600-token one at its
    end enlarge the slabs between decode rounds (256 -> 512 -> 1024), the decode step is re-captured and the slots that were
    mid-answer continue.  Answers must be exactly those of one-request-at-a-time decoding, and the cache must have grown."""
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.serving import ContinuousBatcher
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    reqs = _requests(6)
    g = torch.Generator().manual_seed(5)
    reqs.append(([torch.randn(3, 56, 70, generator=g).clamp(-1, 1), torch.randn(3, 70, 70, generator=g).clamp(-1, 1)],
                 " ".join(str(int(v)) for v in torch.randint(5, 290, (600,), generator=g))))      # the one that forces a second doubling
    reqs[2] = (reqs[2][0], " ".join(str(int(v)) for v in torch.randint(5, 290, (280,), generator=g)))
    budgets = [6, 3, 6, 5, 12, 2, 7]
    ident = lambda x: x   # noqa: E731
    want = [model.chat(tok, NEW_TOKEN_IDS, ident, images, prompt, max_length=nb + 1) for (images, prompt), nb in zip(reqs, budgets)]
    srv = ContinuousBatcher(model, tok, NEW_TOKEN_IDS, ident, slots=slots, max_context=16, max_new_tokens=8, check_every=check_every,
                            use_graph=use_graph)
    cap0 = srv.cache.cap
    rids = [srv.submit(images, prompt, max_new_tokens=nb) for (images, prompt), nb in zip(reqs, budgets)]
    got = srv.run()
    for rid, w in zip(rids, want):
        assert got[rid] == w, (rid, got[rid], w)
    assert cap0 == 256 and srv.stats["cache_grows"] >= 2 and srv.cache.cap >= 1024
    # a limit on the context is still available
    lim = ContinuousBatcher(model, tok, NEW_TOKEN_IDS, ident, slots=2, max_context=16, max_new_tokens=4, context_limit=64)
    lim.submit(None, " ".join(["7"] * 100))
    with pytest.raises(ValueError, match="context_limit"):
        lim.run()


@pytest.mark.parametrize("slots,check_every", [(4, 3), (8, 4)])
def test_batched_admission_matches_single_requests(model, slots, check_every):
    """requests with the same item structure (one image + a prompt) that are admitted in the same round are prefilled by ONE
    packed ViT + LLM forward into their (non-contiguous) slots while the other slots keep decoding; every answer must equal
    the one-request-at-a-time answer (bagel.py:1321-1392 per request)."""
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.serving import ContinuousBatcher
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    g = torch.Generator().manual_seed(33)
    sizes = [(42, 56), (28, 70), (56, 56), (42, 42), (56, 28), (28, 28), (70, 28), (42, 70), (56, 42), (28, 56), (42, 28)]
    reqs = [([torch.randn(3, h, w, generator=g).clamp(-1, 1)], " ".join(str(int(v)) for v in torch.randint(5, 290, (2 + i % 5,), generator=g)))
            for i, (h, w) in enumerate(sizes)]
    budgets = [6, 2, 5, 2, 6, 3, 2, 6, 4, 2, 5]         # several slots finish in the same round -> batched refills
    ident = lambda x: x   # noqa: E731
    want = [model.chat(tok, NEW_TOKEN_IDS, ident, images, prompt, max_length=nb + 1) for (images, prompt), nb in zip(reqs, budgets)]
    srv = ContinuousBatcher(model, tok, NEW_TOKEN_IDS, ident, slots=slots, max_context=256, max_new_tokens=8, check_every=check_every)
    rids = [srv.submit(images, prompt, max_new_tokens=nb) for (images, prompt), nb in zip(reqs, budgets)]
    got = srv.run()
    for rid, w in zip(rids, want):
        assert got[rid] == w, (rid, got[rid], w)
    assert srv.stats["prefills"] == len(reqs) and srv.stats.get("batched_prefills", 0) >= 2


def test_image_prefill_graph_equals_eager(model):
    """the image-span prefill into a RESERVED cache replays from a HIP graph keyed by the patch grid (Bagel._vit_graph_run);
    K / V and the decoded answer must equal the eager path bit for bit, for a second image of the same size (replay with new
    pixels), at another cache offset, and in another segment of a multi-slot cache."""
    from unimedvl_amd.kvcache import NaiveCache
    cfg = model.cfg
    g = torch.Generator().manual_seed(44)
    imgs = [torch.randn(3, 42, 56, generator=g).clamp(-1, 1) for _ in range(3)]

    class Tok:
        def encode(self, s):
            return [int(x) for x in s.split()]
    ident = lambda x: x   # noqa: E731

    def run(cache, images, slot=None, nslots=1):
        kvl, rope = [0] * nslots, [0] * nslots
        for im in images:
            if slot is None:
                gi, kvl, rope = model.prepare_vit_images(kvl, rope, [im], ident, NEW_TOKEN_IDS)
            else:       # one image into segment `slot` of a multi-slot cache: the other segments take part with zero tokens
                gi, k1, r1 = model.prepare_vit_images([kvl[slot]], [rope[slot]], [im], ident, NEW_TOKEN_IDS)
                full = torch.zeros(nslots, dtype=gi["packed_seqlens"].dtype)
                full[slot] = gi["packed_seqlens"][0]
                gi["packed_seqlens"] = full
                gi["key_values_lens"] = gi["packed_key_value_indexes"] = gi["packed_indexes"] = None
                kvl[slot], rope[slot] = k1[0], r1[0]
            cache = model.forward_cache_update_vit(cache, **gi)
        return cache, kvl, rope

    eager, kvl, rope = run(NaiveCache(cfg.layers), imgs[:2])              # not reserved -> eager kernels
    assert not model._vit_graphs or True
    model._vit_graphs.clear()          # the graph store is bounded (oldest entry dropped): count from empty
    n_before = len(model._vit_graphs)
    pooled = NaiveCache(cfg.layers)
    pooled.reserve(1, 512, cfg.kv_heads, cfg.head_dim, model.device)
    pooled, kvl2, rope2 = run(pooled, imgs[:2])                           # capture at offset 0, replay at offset 14
    assert len(model._vit_graphs) == n_before + 1, "same patch grid, same cache: one graph serves both offsets"
    assert kvl2 == kvl and rope2 == rope and pooled.lens == eager.lens
    for l in range(cfg.layers):
        assert torch.equal(pooled.packed_keys(l), eager.packed_keys(l)) and torch.equal(pooled.packed_values(l), eager.packed_values(l))
    gi, kv3, rp3 = model.prepare_prompts(kvl, rope, ["5 6 7"], Tok(), NEW_TOKEN_IDS)
    a = model.generate_text(past_key_values=model.forward_cache_update_text(eager, **gi), max_length=5,
                            **model.prepare_start_tokens(kv3, rp3, NEW_TOKEN_IDS))
    b = model.generate_text(past_key_values=model.forward_cache_update_text(pooled, **gi), max_length=5,
                            **model.prepare_start_tokens(kv3, rp3, NEW_TOKEN_IDS))
    assert torch.equal(a, b)
    # a 4-slot cache: one graph, any slot
    multi = NaiveCache(cfg.layers)
    multi.reserve(4, 256, cfg.kv_heads, cfg.head_dim, model.device)
    multi.lens = [0] * 4
    n0 = len(model._vit_graphs)
    multi, _, _ = run(multi, [imgs[2]], slot=2, nslots=4)
    multi, _, _ = run(multi, [imgs[0]], slot=0, nslots=4)
    assert len(model._vit_graphs) == n0 + 1
    one, _, _ = run(NaiveCache(cfg.layers), [imgs[2]])
    two, _, _ = run(NaiveCache(cfg.layers), [imgs[0]])
    for l in range(cfg.layers):
        assert torch.equal(multi.view_segments(2, 3).packed_keys(l), one.packed_keys(l))
        assert torch.equal(multi.view_segments(0, 1).packed_values(l), two.packed_values(l))
    assert multi.lens == [one.lens[0] if False else two.lens[0], 0, one.lens[0], 0]


def test_chat_with_pooled_cache_equals_plain_chat(model):
    from oracle.toy_tokenizer import ToyTokenizer
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    g = torch.Generator().manual_seed(45)
    imgs = [torch.randn(3, 56, 42, generator=g).clamp(-1, 1) for _ in range(3)]
    ident = lambda x: x   # noqa: E731
    want = [model.chat(tok, NEW_TOKEN_IDS, ident, [im], "9 8 7 6", max_length=6) for im in imgs]
    model.chat_cache_tokens = 256
    try:
        got = [model.chat(tok, NEW_TOKEN_IDS, ident, [im], "9 8 7 6", max_length=6) for im in imgs]
    finally:
        model.chat_cache_tokens = 0
    assert got == want


@pytest.mark.parametrize("paged", [False, True])
def test_mixed_vqa_and_t2i_requests_interleaved(tiny_weights, paged):
    """(paged: the VQA slots on the block-table cache, kvcache.PagedCache.)
    serving.MixedBatcher (BASELINE.json configs[4]: mixed VQA + T2I interleaved batch): VQA decode slots and a
    text-to-image group advance in the same step stream.  Every answer equals Bagel.chat's for that request alone and every
    image's final latent equals Bagel.generate_image's for that prompt and starting noise alone, bit for bit."""
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    from unimedvl_amd.serving import MixedBatcher
    from unimedvl_amd.vae import AutoEncoder
    cfg, sd, vae_sd, _ = tiny_weights
    ucfg = UniMedVLConfig.from_dict(cfg)
    model = Bagel(ucfg, lambda n: sd[n], device="cuda", visual_gen=True)
    vae = AutoEncoder(ucfg, lambda n: vae_sd[n], device="cuda")
    tok = ToyTokenizer(NEW_TOKEN_IDS)
    ident = lambda x: x   # noqa: E731
    reqs = _requests(5)
    budgets = [6, 4, 6, 5, 3]
    want_text = [model.chat(tok, NEW_TOKEN_IDS, ident, images, prompt, max_length=nb + 1) for (images, prompt), nb in zip(reqs, budgets)]
    hw = (32, 32)
    down = model.latent_downsample
    ntok, D = (hw[0] // down) * (hw[1] // down), model.latent_patch_size ** 2 * model.latent_channel
    g = torch.Generator().manual_seed(5)
    t2i = [("7 8 9", torch.randn(ntok, D, generator=g)), ("10 11", torch.randn(ntok, D, generator=g)), ("12 13 14 15", torch.randn(ntok, D, generator=g))]
    kw = dict(num_timesteps=6, timestep_shift=3.0, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0,
              cfg_renorm_type="global")

    def alone(prompt, noise):
        gen = NaiveCache(ucfg.layers)
        gi, kvl, rope = model.prepare_prompts([0], [0], [prompt], tok, NEW_TOKEN_IDS)
        gen = model.forward_cache_update_text(gen, **gi)
        gl = model.prepare_vae_latent(kvl, rope, [hw], NEW_TOKEN_IDS)
        gl["packed_init_noises"] = noise.clone()
        gt = model.prepare_vae_latent_cfg([0], [0], [hw])
        gim = model.prepare_vae_latent_cfg(kvl, rope, [hw])
        lat = model.generate_image(
            past_key_values=gen, cfg_text_past_key_values=NaiveCache(ucfg.layers), cfg_img_past_key_values=gen.snapshot(), **kw, **gl,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_img_packed_position_ids=gim["cfg_packed_position_ids"])
        return lat[0].clone()
    want_lat = [alone(p, n) for p, n in t2i]

    srv = MixedBatcher(model, vae, tok, NEW_TOKEN_IDS, ident, slots=3, t2i_batch=2, flow_steps_per_round=2, max_context=256,
                       max_new_tokens=8, check_every=3, paged=paged)
    rids = [srv.submit(images, prompt, max_new_tokens=nb) for (images, prompt), nb in zip(reqs, budgets)]
    iids = [srv.submit_t2i(p, hw, init_noise=n, **kw) for p, n in t2i]
    got = srv.run()
    assert sorted(got) == sorted(rids + iids)
    for rid, w in zip(rids, want_text):
        assert got[rid] == w, (rid, got[rid], w)
    for iid, w in zip(iids, want_lat):
        assert got[iid].dtype == torch.uint8 and tuple(got[iid].shape) == (hw[0], hw[1], 3)
        assert torch.equal(srv.latents[iid], w), f"image request {iid}: latent differs from the request served alone"
    st = srv.stats
    assert st["images"] == 3 and st["t2i_groups"] == 2 and st["flow_steps"] == 2 * 5 and st["interleaved_rounds"] >= 6
    assert st["decode_steps"] > 0 and st["prefills"] == 5

    # image requests only: the same loop without a decode session
    srv2 = MixedBatcher(model, vae, tok, NEW_TOKEN_IDS, ident, slots=2, t2i_batch=4, flow_steps_per_round=3, max_context=64, max_new_tokens=4)
    iid = srv2.submit_t2i(t2i[0][0], hw, init_noise=t2i[0][1], **kw)
    out = srv2.run()
    assert torch.equal(srv2.latents[iid], want_lat[0]) and out[iid].shape == (hw[0], hw[1], 3)
