"""NaiveCache.snapshot(): O(1) prefix sharing with copy-on-write must behave exactly like the
deepcopy it replaces (inferencer.py:261,587,600,607)."""
from copy import deepcopy

import pytest
import torch

from conftest import NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu


class ListTokenizer:
    def encode(self, s):
        return [int(x) for x in s.split()]


@pytest.fixture(scope="module")
def engine(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    cfg, sd, _, _ = tiny_weights
    return Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda")


def _text(model, cache, kvl, rope, prompt):
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [prompt], ListTokenizer(), NEW_TOKEN_IDS)
    return model.forward_cache_update_text(cache, **gi), kvl, rope


def test_snapshot_is_a_frozen_prefix_and_copies_on_write(engine):
    from unimedvl_amd.kvcache import NaiveCache
    L = engine.cfg.layers
    cache, kvl, rope = _text(engine, NaiveCache(L), [0], [0], "5 6 7 8 9 10 11")
    snap, deep = cache.snapshot(), deepcopy(cache)
    assert snap.shares_storage_with(cache) and not deep.shares_storage_with(cache) and snap.lens == cache.lens == [9]
    # the source keeps appending in place: the snapshot still shows the old prefix, bit for bit
    cache, kvl2, rope2 = _text(engine, cache, kvl, rope, "12 13 14")
    assert cache.lens == [14] and snap.lens == [9]
    for l in range(L):
        assert torch.equal(snap.packed_keys(l), deep.packed_keys(l)) and torch.equal(snap.packed_values(l), deep.packed_values(l))
        assert torch.equal(cache.packed_keys(l)[:9], deep.packed_keys(l))
    # writing through the snapshot copies first: the source's tokens 9..13 survive, the snapshot equals a deepcopy + same append
    before = [cache.packed_keys(l).clone() for l in range(L)]
    snap, _, _ = _text(engine, snap, kvl, rope, "20 21")
    deep, _, _ = _text(engine, deep, kvl, rope, "20 21")
    assert not snap.shares_storage_with(cache) and snap.lens == deep.lens == [13]
    for l in range(L):
        assert torch.equal(cache.packed_keys(l), before[l])
        assert torch.equal(snap.packed_keys(l), deep.packed_keys(l)) and torch.equal(snap.packed_values(l), deep.packed_values(l))
    # empty cache: a snapshot is just another empty cache
    e = NaiveCache(L)
    s = e.snapshot()
    assert s.slabs is None and s.seq_lens == 0
    s, _, _ = _text(engine, s, [0], [0], "1 2")
    assert e.slabs is None and s.lens == [4]


def test_decode_in_place_then_rewind_equals_decode_on_a_copy(engine):
    from unimedvl_amd.kvcache import NaiveCache
    L = engine.cfg.layers
    cache, kvl, rope = _text(engine, NaiveCache(L), [0], [0], "30 31 32 33")
    gi = engine.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    ref_ids = engine.generate_text(past_key_values=deepcopy(cache), max_length=6, **gi)
    snap = cache.snapshot()
    lens0 = list(cache.lens)
    ids = engine.generate_text(past_key_values=cache, max_length=6, **gi)
    cache.lens = lens0
    assert torch.equal(ids, ref_ids)
    again = engine.generate_text(past_key_values=cache, max_length=6, **gi)      # the rewound context decodes the same again
    cache.lens = lens0
    assert torch.equal(again, ref_ids)
    for l in range(L):
        assert torch.equal(snap.packed_keys(l), cache.packed_keys(l))


def test_pure_t2i_contexts_share_storage_and_skip_the_compare(engine):
    """after text-only inputs cfg_img_context is gen_context's own prefix: generate_image sees identical contexts without
    comparing KV on the device"""
    from unimedvl_amd.kvcache import NaiveCache
    L = engine.cfg.layers
    cache, kvl, rope = _text(engine, NaiveCache(L), [0], [0], "40 41 42")
    snap = cache.snapshot()
    pos = torch.tensor([rope[0]] * 6)
    assert engine._contexts_identical(cache, pos, snap, pos.clone())
    cache2, _, _ = _text(engine, cache, kvl, rope, "43")
    assert not engine._contexts_identical(cache2, pos, snap, pos)
