"""BUILD-CONTAINER ONLY (needs /root/reference): time the ACTUAL reference (imported over the shims of ref_import.py)
against the CPU restatement (oracle/unimedvl_cpu.py) on the same mid-size config, weights and inputs - BASELINE.md section 3.5:
the restatement is what bench.py's cpu_baseline times on the GPU box (the reference's Python cannot travel), so its cost
must match the reference's own within noise.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.time_ref_vs_port

Prints one JSON object (copied into BASELINE.md).  TEST INFRASTRUCTURE."""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.gen_golden import NEW_TOKEN_IDS, ListTokenizer, build_reference, synth_image  # noqa: E402
from oracle.unimedvl_cpu import KVCache, OracleBagel  # noqa: E402

MID = dict(hidden=1024, layers=4, heads=8, kv_heads=2, inter=4096, vocab=4096, vit_hidden=384, vit_layers=4, vit_heads=6,
           vit_inter=1536, patch=14, vit_side=70, max_latent=64, vae_ch=32, vae_mult=(1, 2, 4, 4), vae_res=1, z_channels=16,
           rope_theta=1e6, rms_eps=1e-6, ln_eps=1e-6, latent_patch=2, scale_factor=0.3611, shift_factor=0.1159)
NTID = dict(bos_token_id=4000, eos_token_id=4001, start_of_image=4002, end_of_image=4003)


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


@torch.no_grad()
def main():
    ns, model, vae, sd, vae_sd = build_reference(MID)
    o = OracleBagel(MID, sd, vae_sd, attn_impl="sdpa")
    tok = ListTokenizer()
    img = synth_image(448, 448, 5)
    prompt_ids = torch.randint(10, 3900, (32,), generator=torch.Generator().manual_seed(1)).tolist()
    prompt = " ".join(map(str, prompt_ids))
    ac = torch.autocast("cpu", dtype=torch.bfloat16)
    L = MID["layers"]
    ident = lambda x: x   # noqa: E731
    out = {"config": {k: MID[k] for k in ("hidden", "layers", "heads", "kv_heads", "inter", "vocab", "vit_hidden", "vit_layers")},
           "threads": torch.get_num_threads(), "image": "448x448 (1024 patches)", "prompt_tokens": 32, "decode_steps": 16}

    # ---- reference
    def ref_prefill():
        with ac:
            cache = ns.NaiveCache(L)
            gi, kvl, rope = model.prepare_vit_images([0], [0], [img], ident, NTID)
            cache = model.forward_cache_update_vit(cache, **gi)
            gi, kvl, rope = model.prepare_prompts(kvl, rope, [prompt], tok, NTID)
            cache = model.forward_cache_update_text(cache, **gi)
        return cache, kvl, rope

    def ref_decode():
        cache, kvl, rope = ref_prefill()
        t0 = time.perf_counter()
        with ac:
            gi = model.prepare_start_tokens(kvl, rope, NTID)
            ids = model.generate_text(past_key_values=cache, max_length=16, do_sample=False, end_token_id=None, **gi)
        return time.perf_counter() - t0, ids

    # ---- restatement
    def port_prefill():
        c = KVCache(L, 1)
        kvl, rope = o.update_vit(c, [0], [0], [img], NTID)
        kvl, rope = o.update_text(c, kvl, rope, [[NTID["bos_token_id"]] + prompt_ids + [NTID["eos_token_id"]]])
        return c, kvl, rope

    def port_decode():
        c, kvl, rope = port_prefill()
        t0 = time.perf_counter()
        ids = o.generate_text(c, rope, NTID["bos_token_id"], 16)
        return time.perf_counter() - t0, ids

    out["prefill_s"] = {"reference": round(best(ref_prefill), 4), "restatement": round(best(port_prefill), 4)}
    rd = [ref_decode() for _ in range(3)]
    pd = [port_decode() for _ in range(3)]
    out["decode16_s"] = {"reference": round(min(t for t, _ in rd), 4), "restatement": round(min(t for t, _ in pd), 4)}
    out["same_tokens"] = bool(torch.equal(rd[0][1], pd[0][1]))
    out["ratio_restatement_over_reference"] = {k: round(out[k]["restatement"] / out[k]["reference"], 3) for k in ("prefill_s", "decode16_s")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
