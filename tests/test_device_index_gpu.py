"""An engine built on cuda:N (the scripts' `target_gpu_device`, interactive_vqa_inferencer.py:60,105) must launch its
kernels on device N's stream whatever torch's current device is (every umv_* launch used the CURRENT device's
stream).  Needs two GPUs; on the one-GPU test box it checks the guard logic on device 0 only."""
import pytest
import torch

from conftest import NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu


def _run(model, cfg):
    from unimedvl_amd.kvcache import NaiveCache

    class Tok:
        def encode(self, s):
            return [11, 22, 33, 44]
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 42, 56, generator=g).clamp(-1, 1)
    cache = NaiveCache(cfg["layers"])
    gi, kvl, rope = model.prepare_vit_images([0], [0], [img], lambda x: x, NEW_TOKEN_IDS)
    cache = model.forward_cache_update_vit(cache, **gi)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, ["q"], Tok(), NEW_TOKEN_IDS)
    cache = model.forward_cache_update_text(cache, **gi)
    gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
    return model.generate_text(past_key_values=cache, max_length=5, **gi).cpu()


def test_on_device_guard_switches_and_restores(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops

    class Obj:
        device = torch.device("cuda", 0)

        @ops.on_device
        def where(self):
            return torch.cuda.current_device()
    assert Obj().where() == 0
    with ops.device_scope("cuda:0"):
        assert torch.cuda.current_device() == 0
    o = Obj()
    o.device = "cuda"                 # no index: whatever is current
    assert o.where() == torch.cuda.current_device()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_engine_on_second_gpu_matches_first(tiny_weights):
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    cfg, sd, _, _ = tiny_weights
    m0 = Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda:0", visual_gen=False)
    m1 = Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda:1", visual_gen=False)
    assert torch.cuda.current_device() == 0          # building / running on cuda:1 must not leak the device switch
    a, b = _run(m0, cfg), _run(m1, cfg)
    assert torch.cuda.current_device() == 0
    assert torch.equal(a, b)
    assert m1.language_model.w.embed.device.index == 1
