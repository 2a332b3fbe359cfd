#!/usr/bin/env python
"""Single-request latency at full 14B dims (random weights): one 448x448 image + 32-token question -> prefill wall time with
the eager image span vs the HIP-graph replay into a pooled (reserved) cache, then 16 decode steps.  profiles/HISTORY.md section 7.4."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd.bagel import Bagel  # noqa: E402
from unimedvl_amd.config import UniMedVLConfig  # noqa: E402
from unimedvl_amd.kvcache import NaiveCache  # noqa: E402
from unimedvl_amd.weights import random_getter  # noqa: E402

cfg = UniMedVLConfig()
dev = torch.device("cuda", 0)
model = Bagel(cfg, random_getter(cfg, dev, seed=1234), device=dev, visual_gen=False)
ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
g = torch.Generator().manual_seed(1)
images = [torch.rand(3, 448, 448, generator=g) * 2 - 1 for _ in range(6)]
prompt = torch.randint(1000, 150000, (32,), generator=g).tolist()


class Tok:
    def encode(self, s):
        return prompt


def request(img, cache):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache.lens = [0] * max(1, len(cache.lens)) if cache.slabs is not None else []
    gi, kvl, rope = model.prepare_vit_images([0], [0], [img], lambda x: x, ntid)
    t_prep = time.perf_counter() - t0
    cache = model.forward_cache_update_vit(cache, **gi)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, ["q"], Tok(), ntid)
    cache = model.forward_cache_update_text(cache, **gi)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    gi = model.prepare_start_tokens(kvl, rope, ntid)
    ids = model.generate_text(past_key_values=cache, max_length=16, **gi)
    torch.cuda.synchronize()
    return t_prep, t_prefill, time.perf_counter() - t0, ids


for mode in ("eager", "graph"):
    model.prefill_graph = mode == "graph"
    cache = NaiveCache(cfg.layers)
    cache.reserve(1, 2048, cfg.kv_heads, cfg.head_dim, dev)
    out = [request(im, cache) for im in images]
    pre = sorted(t[1] for t in out[2:])
    tot = sorted(t[2] for t in out[2:])
    print(f"{mode:6s}: host prep {out[-1][0] * 1e3:5.1f} ms | prefill wall (image + question) median {pre[len(pre) // 2] * 1e3:6.1f} ms "
          f"min {pre[0] * 1e3:6.1f} | prefill + 16 decode steps median {tot[len(tot) // 2] * 1e3:6.1f} ms | first request {out[0][1] * 1e3:.0f} ms")
