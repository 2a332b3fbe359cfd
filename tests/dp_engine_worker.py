"""Worker of tests/test_dp_engine_gpu.py: one data-parallel rank running the REAL engine (tiny weights) behind
unimedvl_amd.parallel.DataParallelVQA.  Launched by unimedvl_amd.launch.spawn_ranks (or alone, WORLD_SIZE unset)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NTID = dict(bos_token_id=300, eos_token_id=301, start_of_image=302, end_of_image=303)


class ListTokenizer:
    def encode(self, s):
        return [int(x) for x in s.split()]


def items():
    g = torch.Generator().manual_seed(21)
    sizes = [(42, 56), (56, 56), (28, 70), (56, 42), (70, 28)]
    images = [torch.randn(3, h, w, generator=g).clamp(-1, 1) for h, w in sizes]
    prompts = [" ".join(str(int(v)) for v in torch.randint(5, 290, (n,), generator=g)) for n in (4, 9, 6, 3, 7)]
    return images, prompts


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        # two ranks share the one GPU of the test box, so the result gather runs over gloo (RCCL needs one device per rank);
        # the engine path of each rank is the real HIP one
        dist.init_process_group("gloo")
    from oracle.weights import TINY, make_weights
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    from unimedvl_amd.parallel import DataParallelVQA
    sd, _ = make_weights(TINY)
    model = Bagel(UniMedVLConfig.from_dict(TINY), lambda n: sd[n], device="cuda:0", visual_gen=False)
    tok = ListTokenizer()
    calls = []

    def engine(images, prompts):
        B = len(prompts)
        calls.append(B)
        cache = NaiveCache(TINY["layers"])
        gi, kvl, rope = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, NTID)
        cache = model.forward_cache_update_vit(cache, **gi)
        gi, kvl, rope = model.prepare_prompts(kvl, rope, prompts, tok, NTID)
        cache = model.forward_cache_update_text(cache, **gi)
        gi = model.prepare_start_tokens(kvl, rope, NTID)
        ids = model.generate_text(past_key_values=cache, max_length=6, **gi).cpu()
        return [" ".join(str(int(v)) for v in ids[:, b]) for b in range(B)]

    images, prompts = items()
    out = DataParallelVQA(engine)(images, prompts)
    balanced = DataParallelVQA(engine)(images, prompts, lengths=[len(p.split()) for p in prompts])
    if rank == 0:
        print(json.dumps({"world": world, "out": out, "balanced": balanced, "calls": calls}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
