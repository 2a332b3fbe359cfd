"""Edge cases of the hot path on the GPU against the CPU oracle (tiny synthetic model, tolerances of test_engine_gpu.py):
long contexts (tiled GEMM with M in the thousands, long causal prefill attention, decode with the maximum KV split),
the smallest inputs (empty prompt, a one-patch image, max_length 1), and ragged batches that mix them."""
import pytest
import torch

from conftest import NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu
BOS, EOS = NEW_TOKEN_IDS["bos_token_id"], NEW_TOKEN_IDS["eos_token_id"]


@pytest.fixture(scope="module")
def pair(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from oracle.unimedvl_cpu import OracleBagel
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    cfg, sd, vae_sd, _ = tiny_weights
    return Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda", visual_gen=False), OracleBagel(cfg, sd, vae_sd), cfg


class Tok:
    def __init__(self, prompts):
        self.prompts = prompts

    def encode(self, s):
        return self.prompts[int(s)]


def _run(model, oracle, cfg, images, prompts, steps):
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    B = len(prompts)
    cache = NaiveCache(cfg["layers"])
    kvl, rope = [0] * B, [0] * B
    oc = KVCache(cfg["layers"], B)
    okv, orope = [0] * B, [0] * B
    if images is not None:
        gi, kvl, rope = model.prepare_vit_images(kvl, rope, images, lambda x: x, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_vit(cache, **gi)
        okv, orope = oracle.update_vit(oc, okv, orope, images, NEW_TOKEN_IDS)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], Tok(prompts), NEW_TOKEN_IDS)
    cache = model.forward_cache_update_text(cache, **gi)
    okv, orope = oracle.update_text(oc, okv, orope, [[BOS] + p + [EOS] for p in prompts])
    assert okv == kvl and orope == rope
    gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    ids, logits = model.generate_text(past_key_values=cache, max_length=steps, return_logits=True, **gi)
    oids, ologits = oracle.generate_text(oc, orope, BOS, steps, return_logits=True)
    lg, rl = logits.float().cpu(), ologits.float()
    assert ids.shape == (steps, B)
    for s in range(steps):
        assert torch.equal(ids[s].cpu(), oids[s]), f"fed token differs at step {s}"
        d = (lg[s] - rl[s]).abs().max().item()
        assert d <= 0.25, f"logits differ by {d} at step {s}"
        top2 = rl[s].topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > 0.25
        assert torch.equal(lg[s].argmax(-1)[sure], rl[s].argmax(-1)[sure])
        if not torch.equal(lg[s].argmax(-1), rl[s].argmax(-1)):
            break
    return kvl


def test_long_text_context(pair):
    model, oracle, cfg = pair
    g = torch.Generator().manual_seed(1)
    prompts = [[int(v) for v in torch.randint(5, 290, (2100,), generator=g)], [int(v) for v in torch.randint(5, 290, (700,), generator=g)]]
    kvl = _run(model, oracle, cfg, None, prompts, 3)
    assert kvl == [2102, 702]      # ragged, > 2048 keys: decode attention runs its maximum split


def test_smallest_inputs(pair):
    model, oracle, cfg = pair
    g = torch.Generator().manual_seed(2)
    one_patch = torch.randn(3, 14, 14, generator=g).clamp(-1, 1)
    wide = torch.randn(3, 14, 112, generator=g).clamp(-1, 1)        # 1 x 8 patches, the side limit of the tiny ViT table
    kvl = _run(model, oracle, cfg, [one_patch, wide], [[], [7]], 2)  # an empty prompt is just <|im_start|><|im_end|>
    assert kvl == [1 + 2 + 2, 8 + 2 + 3]
    _run(model, oracle, cfg, None, [[9]], 1)                         # max_length 1: one step, B = 1


def test_generate_text_zero_and_eos(pair):
    model, _, cfg = pair
    from unimedvl_amd.kvcache import NaiveCache
    cache = NaiveCache(cfg["layers"])
    gi, kvl, rope = model.prepare_prompts([0], [0], ["0"], Tok([[5, 6]]), NEW_TOKEN_IDS)
    cache = model.forward_cache_update_text(cache, **gi)
    gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    out = model.generate_text(past_key_values=cache, max_length=0, **gi)
    assert out.shape[0] == 0 and cache.lens == kvl
    # an end token that is produced immediately stops after one row (bagel.py:1313)
    first = model.generate_text(past_key_values=cache, max_length=2, **gi)
    nxt = int(first[1, 0])
    cache2 = NaiveCache(cfg["layers"])
    gi2, kvl2, rope2 = model.prepare_prompts([0], [0], ["0"], Tok([[5, 6]]), NEW_TOKEN_IDS)
    cache2 = model.forward_cache_update_text(cache2, **gi2)
    out = model.generate_text(past_key_values=cache2, max_length=5, end_token_id=nxt, **model.prepare_start_tokens(kvl2, rope2, NEW_TOKEN_IDS))
    assert out.shape[0] == 1 and int(out[0, 0]) == BOS


def test_out_of_range_inputs_raise_instead_of_faulting(pair):
    """the kernels index the embedding and rotary tables unchecked; the host rejects ids / positions they do not hold
    (position >= max_position read out of bounds, NaN logits fed an id of 0x7fffffff back into the gather)"""
    from unimedvl_amd.decode import DecodeSession
    from unimedvl_amd.kvcache import NaiveCache
    model, _, cfg = pair
    cache = NaiveCache(cfg["layers"])
    with pytest.raises(ValueError, match="token ids"):
        gi, _, _ = model.prepare_prompts([0], [0], ["0"], Tok([[5, cfg["vocab"] + 3]]), NEW_TOKEN_IDS)
        model.forward_cache_update_text(cache, **gi)
    maxpos = model.cfg.max_position
    with pytest.raises(ValueError, match="position"):
        gi, _, _ = model.prepare_prompts([0], [maxpos - 2], ["0"], Tok([[5, 6, 7]]), NEW_TOKEN_IDS)
        model.forward_cache_update_text(NaiveCache(cfg["layers"]), **gi)
    gi, kvl, rope = model.prepare_prompts([0], [0], ["0"], Tok([[5, 6]]), NEW_TOKEN_IDS)
    cache = model.forward_cache_update_text(NaiveCache(cfg["layers"]), **gi)
    with pytest.raises(ValueError, match="rope position"):
        DecodeSession(model.language_model, cache, torch.tensor([BOS]), torch.tensor([maxpos - 3]), 8)
