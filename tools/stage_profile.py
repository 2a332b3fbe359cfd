#!/usr/bin/env python
"""Run ONE stage of the path repeatedly at the 14B dims so that a rocprofv3 kernel trace isolates it:
    python tools/stage_profile.py vit [B]        ViT tower + connector on B 448x448 images
    python tools/stage_profile.py prefill [B]    forward_cache_update_vit + forward_cache_update_text (image span + question)
    python tools/stage_profile.py t2i [B]        bench.run_t2i: 50 guided flow steps at 256x256 + VAE decode (one timed run)
    python tools/stage_profile.py edit [B]       bench.run_edit: 448x448 -> 512x512 edit flow over three distinct contexts (STEPS timesteps)
Prints the wall time per repetition; under `rocprofv3 --kernel-trace --stats` the per-kernel table is that stage's alone
(plus the one-time weight initialisation, which only uses torch / pack kernels)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import IdTokenizer, run_edit, run_t2i, synth_image  # noqa: E402
from unimedvl_amd.bagel import Bagel  # noqa: E402
from unimedvl_amd.config import UniMedVLConfig  # noqa: E402
from unimedvl_amd.kvcache import NaiveCache  # noqa: E402
from unimedvl_amd.weights import random_getter  # noqa: E402


def main():
    stage = sys.argv[1] if len(sys.argv) > 1 else "vit"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    reps = int(os.environ.get("REPS", "10"))
    dev = torch.device("cuda", 0)
    cfg = UniMedVLConfig()
    if stage in ("vae", "vae_enc"):       # the full-size AutoEncoder alone: B images of HW x HW (decode) / (encode)
        from bench import synth_vae
        hw = int(os.environ.get("HW", "256"))
        vae = synth_vae(cfg, dev)
        g = torch.Generator(device=dev).manual_seed(1)
        if stage == "vae":
            lats = [torch.randn((hw // 16) ** 2, 4 * cfg.z_channels, device=dev, generator=g) for _ in range(B)]
            fn = lambda: vae.decode_tokens_batch_to_uint8(lats, (hw, hw), 16, 2)   # noqa: E731
        else:
            img = torch.randn(B, 3, hw, hw, device=dev, generator=g).clamp(-1, 1)
            fn = lambda: vae.encode(img)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(f"{stage}: {(time.time() - t0) / reps * 1e3:.3f} ms per repetition ({reps} reps), B={B} {hw}x{hw}")
        return
    cfg.llm_weight_dtype = os.environ.get("WEIGHTS", "bf16")
    if os.environ.get("ACT8", "0") != "0":
        cfg.llm_act_dtype = "fp8"
    model = Bagel(cfg, random_getter(cfg, dev, seed=1234), device=dev, visual_gen=stage in ("t2i", "edit"), visual_und=True)
    if stage == "t2i":
        r = run_t2i(model, cfg, dev, 0, 1, None, batch=B if len(sys.argv) > 2 else 4)
        print(f"t2i: {r['s_per_batch'] * 1e3:.3f} ms per repetition (1 reps) {r['images_per_s']} images/s {r['llm_tflops']} TF/s")
        return
    if stage == "edit":
        r = run_edit(model, cfg, dev, 0, 1, None, batch=B if len(sys.argv) > 2 else 4, num_timesteps=int(os.environ.get("STEPS", "50")))
        print(f"edit: {r['s_per_batch'] * 1e3:.3f} ms per repetition (1 reps) {r['images_per_s']} images/s {r['llm_tflops']} TF/s, context build {r['context_build_s']} s")
        return
    ids = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    images = [synth_image(448, 448, i) for i in range(B)]
    g = torch.Generator().manual_seed(1234)
    prompts = [torch.randint(1000, 150000, (32,), generator=g).tolist() for _ in range(B)]

    def vit():
        gi, _, _ = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ids)
        px, pos = gi["packed_vit_tokens"].to(dev), gi["packed_vit_position_ids"].to(dev)
        return lambda: model.encode_vit(px, pos, gi["vit_token_seqlens"])

    def prefill():
        def run():
            cache = NaiveCache(cfg.layers)
            gi, kvl, rope = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ids)
            cache.reserve(B, max(kvl) + 64, cfg.kv_heads, cfg.head_dim, dev)
            cache = model.forward_cache_update_vit(cache, **gi)
            gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTokenizer(prompts), ids)
            model.forward_cache_update_text(cache, **gi)
        return run

    if os.environ.get("PREFETCH_N"):      # experiment: pull the packed weights of the GEMMs with these N through the caches right before the GEMM
        from experimental import ops as xops
        from unimedvl_amd import ops as _ops
        ns = {int(v) for v in os.environ["PREFETCH_N"].split(",")}
        real = _ops.gemm

        def gemm_pf(x, lin, *a, **k):
            if lin.N in ns and x.shape[0] > 128 and lin.wp is not None:
                xops.prefetch(lin.wp, blocks=256)
            return real(x, lin, *a, **k)
        _ops.gemm = gemm_pf
    fn = {"vit": vit, "prefill": prefill}[stage]()
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{stage} B={B}: {(time.time() - t0) / reps * 1e3:.3f} ms per repetition ({reps} reps)")


if __name__ == "__main__":
    main()
