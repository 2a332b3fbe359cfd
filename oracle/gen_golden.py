"""Generate golden vectors from the IMPORTED REFERENCE (this container only).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Runs /root/reference/codes (via oracle/ref_import.py shims) on a tiny
random-weight UniMedVL under torch.autocast("cpu", bf16) and stores inputs and
outputs as tests/golden/*.npz.  bf16 tensors are stored as their uint16 bit
patterns (suffix ``__bf16``).  The weights are NOT stored: they are re-made
from oracle/weights.py by seed and guarded by a sha256 in each fixture.
"""
import os
import sys
from copy import deepcopy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference  # noqa: E402
from oracle.weights import TINY, make_weights, digest  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
NEW_TOKEN_IDS = dict(bos_token_id=300, eos_token_id=301, start_of_image=302, end_of_image=303)


class ListTokenizer:
    """Stands in for the HF tokenizer (absent: vocab files ship with the
    checkpoint).  'prompts' are space-separated integers."""

    def encode(self, s):
        return [int(x) for x in s.split()]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def pack(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.contiguous().view(torch.int16).numpy().view(np.uint16)
            else:
                out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


def build_reference(cfg, seed=1234):
    ns = import_reference()
    c = cfg
    llm_cfg = ns.Qwen2Config(
        vocab_size=c["vocab"], hidden_size=c["hidden"], intermediate_size=c["inter"],
        num_hidden_layers=c["layers"], num_attention_heads=c["heads"], num_key_value_heads=c["kv_heads"],
        max_position_embeddings=4096, rms_norm_eps=c["rms_eps"], rope_theta=c["rope_theta"], qk_norm=True,
        tie_word_embeddings=False, layer_module="Qwen2MoTDecoderLayer", pad_token_id=None)
    vit_cfg = ns.SiglipVisionConfig(
        hidden_size=c["vit_hidden"], intermediate_size=c["vit_inter"], num_hidden_layers=c["vit_layers"],
        num_attention_heads=c["vit_heads"], num_channels=3, image_size=c["patch"] * c["vit_side"],
        patch_size=c["patch"], hidden_act="gelu_pytorch_tanh", layer_norm_eps=c["ln_eps"], rope=False)
    vae_params = ns.AutoEncoderParams(
        resolution=256, in_channels=3, downsample=2 ** (len(c["vae_mult"]) - 1), ch=c["vae_ch"], out_ch=3,
        ch_mult=list(c["vae_mult"]), num_res_blocks=c["vae_res"], z_channels=c["z_channels"],
        scale_factor=c["scale_factor"], shift_factor=c["shift_factor"])
    vae = ns.AutoEncoder(vae_params)
    lm = ns.Qwen2ForCausalLM(llm_cfg)
    vit = ns.SiglipVisionModel(vit_cfg)
    bcfg = ns.BagelConfig(visual_gen=True, visual_und=True, llm_config=llm_cfg, vit_config=vit_cfg,
                          vae_config=vae_params, vit_max_num_patch_per_side=c["vit_side"],
                          connector_act="gelu_pytorch_tanh", latent_patch_size=c["latent_patch"],
                          max_latent_size=c["max_latent"])
    model = ns.Bagel(lm, vit, bcfg)
    model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_cfg)
    sd, vae_sd = make_weights(c, seed)
    model = model.to(torch.bfloat16).eval()
    vae = vae.to(torch.bfloat16).eval()
    # .to(bf16) also rounds the non-persistent rotary inv_freq buffer; the deployed
    # model keeps it fp32 (it is created outside the checkpoint and accelerate's
    # dtype cast only touches loaded tensors, interactive_vqa_inferencer.py:158-161),
    # so restore the fp32 values the module saved at construction.
    rot = model.language_model.model.rotary_emb
    rot.inv_freq = rot.original_inv_freq.float()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing
    vae.load_state_dict(vae_sd, strict=True)
    return ns, model, vae, sd, vae_sd


def synth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 1, h // 7 + 2, w // 7 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)[0]
    img = (img / img.abs().max()).clamp(-1, 1)
    return img.repeat(3, 1, 1).contiguous()


def to_dev(d):
    return d


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    cfg = dict(TINY)
    ns, model, vae, sd, vae_sd = build_reference(cfg)
    wdig = digest(sd) + ":" + digest(vae_sd)
    tok = ListTokenizer()
    ident = lambda x: x  # images are passed as already-transformed tensors
    L = cfg["layers"]
    logits_log = []
    model.language_model.lm_head.register_forward_hook(lambda m, i, o: logits_log.append(o.detach().clone()))

    ac = torch.autocast("cpu", dtype=torch.bfloat16)

    def fresh():
        return ns.NaiveCache(L), [0], [0]

    # ------------------------------------------------------------------ A: ViT tower only
    img_a = synth_image(56, 42, 11)
    with ac:
        gi, _, _ = model.prepare_vit_images([0], [0], [img_a], ident, NEW_TOKEN_IDS)
        cu = torch.nn.functional.pad(torch.cumsum(gi["vit_token_seqlens"], 0), (1, 0)).to(torch.int32)
        vit_out = model.vit_model(packed_pixel_values=gi["packed_vit_tokens"],
                                  packed_flattened_position_ids=gi["packed_vit_position_ids"],
                                  cu_seqlens=cu, max_seqlen=int(gi["vit_token_seqlens"].max()))
        conn = model.connector(vit_out) + model.vit_pos_embed(gi["packed_vit_position_ids"])
    np.savez(os.path.join(OUT, "vit.npz"), **pack(dict(
        weights_sha=wdig, image=img_a, vit_out=vit_out, connector_out=conn)))

    # ------------------------------------------------------------------ B: VQA, B=1
    prompt = "17 45 99 3 250 8"
    with ac:
        cache, kvl, rope = fresh()
        gi, kvl, rope = model.prepare_vit_images(kvl, rope, [img_a], ident, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_vit(cache, **gi)
        k0_vit = cache.key_cache[0].clone(); vL_vit = cache.value_cache[L - 1].clone()
        gi, kvl, rope = model.prepare_prompts(kvl, rope, [prompt], tok, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_text(cache, **gi)
        k0_txt = cache.key_cache[0].clone(); vL_txt = cache.value_cache[L - 1].clone()
        gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
        logits_log.clear()
        ids = model.generate_text(past_key_values=cache, max_length=8, do_sample=False,
                                  end_token_id=None, **gi)
    np.savez(os.path.join(OUT, "vqa_b1.npz"), **pack(dict(
        weights_sha=wdig, image=img_a, prompt_ids=torch.tensor(tok.encode(prompt)),
        kv_lens=torch.tensor(kvl), ropes=torch.tensor(rope),
        k0_after_vit=k0_vit, vL_after_vit=vL_vit, k0_after_text=k0_txt, vL_after_text=vL_txt,
        token_ids=ids, logits=torch.stack(logits_log, 0))))

    # ------------------------------------------------------------------ C: VQA, B=2 ragged
    img_c0, img_c1 = synth_image(42, 70, 21), synth_image(28, 28, 22)
    prompts = ["5 6 7 8 9 10 11", "200 100"]
    with ac:
        cache = ns.NaiveCache(L); kvl, rope = [0, 0], [0, 0]
        gi, kvl, rope = model.prepare_vit_images(kvl, rope, [img_c0, img_c1], ident, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_vit(cache, **gi)
        gi, kvl, rope = model.prepare_prompts(kvl, rope, prompts, tok, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_text(cache, **gi)
        gi = model.prepare_start_tokens(kvl, rope, NEW_TOKEN_IDS)
        logits_log.clear()
        ids = model.generate_text(past_key_values=cache, max_length=6, do_sample=False,
                                  end_token_id=None, **gi)
    np.savez(os.path.join(OUT, "vqa_b2.npz"), **pack(dict(
        weights_sha=wdig, image0=img_c0, image1=img_c1,
        prompt0=torch.tensor(tok.encode(prompts[0])), prompt1=torch.tensor(tok.encode(prompts[1])),
        kv_lens=torch.tensor(kvl), ropes=torch.tensor(rope),
        token_ids=ids, logits=torch.stack(logits_log, 0))))

    # ------------------------------------------------------------------ D: VAE
    g = torch.Generator().manual_seed(5)
    z = torch.randn(1, 16, 8, 8, generator=g)
    img_v = synth_image(64, 48, 31).unsqueeze(0)
    with ac:
        dec = vae.decode(z.to(torch.bfloat16))
        torch.manual_seed(77)
        enc = vae.encode(img_v)
        torch.manual_seed(77)
        noise = torch.randn_like(enc)
    np.savez(os.path.join(OUT, "vae.npz"), **pack(dict(
        weights_sha=wdig, z=z, decoded=dec, image=img_v, enc_noise=noise, encoded=enc)))

    # ------------------------------------------------------------------ E: T2I (text -> image latents + pixels)
    t2i_prompt = "40 41 42 43 44"
    H = W = 64
    out_e = dict(weights_sha=wdig, prompt_ids=torch.tensor(tok.encode(t2i_prompt)), image_shape=torch.tensor([H, W]))
    with ac:
        gen_cache, gkv, grope = fresh()
        cfg_text_cache, ckv, crope = fresh()        # context without the text (inferencer.py:600)
        gi, gkv, grope = model.prepare_prompts(gkv, grope, [t2i_prompt], tok, NEW_TOKEN_IDS)
        gen_cache = model.forward_cache_update_text(gen_cache, **gi)
        cfg_img_cache = deepcopy(gen_cache); ikv, irope = list(gkv), list(grope)   # text only (inferencer.py:602)
        torch.manual_seed(123)
        gi = model.prepare_vae_latent(gkv, grope, [(H, W)], NEW_TOKEN_IDS)
        out_e["init_noise"] = gi["packed_init_noises"].clone()
        gct = model.prepare_vae_latent_cfg(ckv, crope, [(H, W)])
        gci = model.prepare_vae_latent_cfg(ikv, irope, [(H, W)])
        for rtype in ("global", "channel", "text_channel"):
            lat = model.generate_image(
                past_key_values=gen_cache, cfg_text_past_key_values=cfg_text_cache,
                cfg_img_past_key_values=cfg_img_cache, num_timesteps=6, cfg_text_scale=4.0, cfg_img_scale=1.5,
                cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type=rtype, timestep_shift=3.0, **gi,
                cfg_text_packed_position_ids=gct["cfg_packed_position_ids"],
                cfg_text_packed_query_indexes=gct["cfg_packed_query_indexes"],
                cfg_text_key_values_lens=gct["cfg_key_values_lens"],
                cfg_text_packed_key_value_indexes=gct["cfg_packed_key_value_indexes"],
                cfg_img_packed_position_ids=gci["cfg_packed_position_ids"],
                cfg_img_packed_query_indexes=gci["cfg_packed_query_indexes"],
                cfg_img_key_values_lens=gci["cfg_key_values_lens"],
                cfg_img_packed_key_value_indexes=gci["cfg_packed_key_value_indexes"])
            out_e["latent_" + rtype] = lat[0].clone()
        # no-CFG run and a one-step velocity for fine-grained checks
        lat = model.generate_image(past_key_values=gen_cache, num_timesteps=4, cfg_text_scale=1.0,
                                   cfg_img_scale=1.0, timestep_shift=3.0, **gi)
        out_e["latent_nocfg"] = lat[0].clone()
        # decode (inferencer.py:234-256)
        h = w = H // model.latent_downsample
        latent = out_e["latent_global"].reshape(1, h, w, 2, 2, 16)
        latent = torch.einsum("nhwpqc->nchpwq", latent).reshape(1, 16, h * 2, w * 2)
        image = vae.decode(latent.to(torch.bfloat16))
        out_e["decoded"] = image.clone()
        image = (image * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255
        out_e["pixels_u8"] = image.to(torch.uint8)
    np.savez(os.path.join(OUT, "t2i.npz"), **pack(out_e))

    # ------------------------------------------------------------------ F: edit path (VAE-encoded image in context, gen-mode prefill)
    img_f = synth_image(64, 64, 41)
    with ac:
        cache, kvl, rope = fresh()
        torch.manual_seed(9)
        gi, kvl, rope = model.prepare_vae_images(kvl, rope, [img_f], ident, NEW_TOKEN_IDS)
        cache = model.forward_cache_update_vae(vae, cache, **gi)
        k0 = cache.key_cache[0].clone(); vL = cache.value_cache[L - 1].clone()
    # the reference drew randn_like(mean) with mean bf16 [1,16,8,8]; record exactly that draw
    torch.manual_seed(9)
    enc_noise = torch.randn_like(torch.empty(1, 16, 8, 8, dtype=torch.bfloat16))
    np.savez(os.path.join(OUT, "edit_prefill.npz"), **pack(dict(
        weights_sha=wdig, image=img_f, enc_noise=enc_noise, kv_lens=torch.tensor(kvl), ropes=torch.tensor(rope),
        k0=k0, vL=vL)))

    # ------------------------------------------------------------------ G: host packing (prepare_* index tensors)
    out_g = dict(weights_sha=wdig, image0=img_c0, image1=img_c1)

    def put(prefix, gi):
        for k, v in gi.items():
            if torch.is_tensor(v):
                out_g[prefix + k] = v.clone()
            elif isinstance(v, list):
                out_g[prefix + k] = torch.tensor(v)
    kv0, rp0 = [3, 0], [2, 0]
    gi, kv1, rp1 = model.prepare_vit_images(kv0, rp0, [img_c0, img_c1], ident, NEW_TOKEN_IDS)
    put("vit.", gi)
    gi, kv2, rp2 = model.prepare_prompts(kv1, rp1, prompts, tok, NEW_TOKEN_IDS)
    put("txt.", gi)
    img_g0, img_g1 = synth_image(64, 48, 51), synth_image(32, 32, 52)
    out_g["vimage0"], out_g["vimage1"] = img_g0, img_g1
    gi, kv3, rp3 = model.prepare_vae_images(kv2, rp2, [img_g0, img_g1], ident, NEW_TOKEN_IDS)
    put("vae.", gi)
    torch.manual_seed(77)
    gi = model.prepare_vae_latent(kv3, rp3, [(64, 64), (32, 48)], NEW_TOKEN_IDS)
    put("lat.", gi)
    put("cfg.", model.prepare_vae_latent_cfg(kv1, rp1, [(64, 64), (32, 48)]))
    put("start.", model.prepare_start_tokens(kv3, rp3, NEW_TOKEN_IDS))
    out_g["counters"] = torch.tensor([kv1, rp1, kv2, rp2, kv3, rp3])
    np.savez(os.path.join(OUT, "prep.npz"), **pack(out_g))

    # ------------------------------------------------------------------ H: top-level InterleaveInferencer API (PIL in, str / PIL out)
    from PIL import Image
    import inferencer as ref_inferencer
    from oracle.toy_tokenizer import ToyTokenizer
    from oracle.ref_import import import_reference_transforms
    ref_tf = import_reference_transforms()               # the REFERENCE's data/transforms.py over torchvision / cv2 stand-ins
    ttok = ToyTokenizer(NEW_TOKEN_IDS)
    vae_tf, vit_tf = ref_tf.ImageTransform(64, 32, 16), ref_tf.ImageTransform(56, 28, 14)
    inf = ref_inferencer.InterleaveInferencer(model, vae, ttok, vae_tf, vit_tf, NEW_TOKEN_IDS)
    arr = ((synth_image(50, 40, 61)[0] * 0.5 + 0.5) * 255).clamp(0, 255).to(torch.uint8).numpy()
    pil = Image.fromarray(np.stack([arr, arr, arr], -1))
    # the reference wraps these calls in autocast("cuda") which is a no-op on CPU; supply the CPU autocast
    with ac:
        und = inf(image=pil, text="5 6 7 8", understanding_output=True, max_think_token_n=6)
        torch.manual_seed(11)
        t2i = inf(text="40 41 42", image_shapes=(64, 64), num_timesteps=4, cfg_text_scale=4.0, cfg_img_scale=1.5,
                  cfg_interval=(0.4, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
        torch.manual_seed(12)
        edit = inf(image=pil, text="9 10", image_shapes=(64, 48), num_timesteps=3, cfg_text_scale=4.0, cfg_img_scale=2.0,
                   cfg_interval=(0.0, 1.0), timestep_shift=3.0, cfg_renorm_type="text_channel")
        # VQA + reconstruction variants (inferencer.py:282-549): ver1 through __call__, ver0_1 / ver0 called directly
        rec = dict(reconstruct_image=True, max_think_token_n=6, num_timesteps=3, cfg_text_scale=4.0, cfg_img_scale=2.0,
                   cfg_interval=(0.0, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
        torch.manual_seed(13)
        ver1 = inf(image=pil, text="5 6 7 8", inference_ver=1, **rec)
        torch.manual_seed(14)
        ver01 = inf.interleave_inference_for_vqa_reconstruction_ver0_1([pil, "5 6 7 8"], **rec)
        # ver0 (inferencer.py:466-549): two input images, only the FIRST is reconstructed
        torch.manual_seed(16)
        ver0 = inf.interleave_inference_for_vqa_reconstruction_ver0([pil, pil, "5 6 7 8"], **rec)
        # think=True (inferencer.py:23-28,590-596,617-620): the English system prompts go through the tokenizer
        think_und = inf(image=pil, text="5 6 7 8", think=True, understanding_output=True, max_think_token_n=6)
        torch.manual_seed(15)
        think_gen = inf(text="40 41 42", think=True, max_think_token_n=5, image_shapes=(64, 64), num_timesteps=3,
                        cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
        # Bagel.chat (bagel.py:1321-1392): two images then the prompt, ViT-only context, greedy, stops at eos of sample 0
        from data.data_utils import pil_img2rgb as ref_pil_img2rgb
        arr2 = ((synth_image(30, 64, 62)[0] * 0.5 + 0.5) * 255).clamp(0, 255).to(torch.uint8).numpy()
        pil2 = Image.fromarray(arr2)                     # mode "L": pil_img2rgb converts
        chat_text = model.chat(ttok, dict(NEW_TOKEN_IDS), vit_tf, [ref_pil_img2rgb(pil), ref_pil_img2rgb(pil2)], "5 6 7 8", max_length=8)
    np.savez(os.path.join(OUT, "inferencer.npz"), **pack(dict(
        weights_sha=wdig, pil_image=torch.from_numpy(np.asarray(pil).copy()), und_text=und["text"],
        t2i_image=torch.from_numpy(np.asarray(t2i["image"]).copy()),
        edit_image=torch.from_numpy(np.asarray(edit["image"]).copy()),
        ver1_text=ver1["text"], ver1_image=torch.from_numpy(np.asarray(ver1["image"]).copy()),
        ver01_text=ver01[0], ver01_image=torch.from_numpy(np.asarray(ver01[1]).copy()),
        ver0_text=ver0[0], ver0_image=torch.from_numpy(np.asarray(ver0[1]).copy()), ver0_len=len(ver0),
        think_und_text=think_und["text"], think_gen_text=think_gen["text"],
        think_gen_image=torch.from_numpy(np.asarray(think_gen["image"]).copy()),
        pil_image2=torch.from_numpy(arr2.copy()), chat_text=chat_text)))

    # ------------------------------------------------------------------ I: host image transform (data/transforms.py:15-115)
    import hashlib
    psets = [(980, 378, 14, 2_007_040), (980, 387, 14, 14 * 14 * 9 * 1024), (1024, 512, 16, 14 * 14 * 9 * 1024),
             (64, 32, 16, 14 * 14 * 9 * 1024), (56, 28, 14, 14 * 14 * 9 * 1024), (518, 224, 14, 14 * 14 * 400)]
    dims = [(448, 448), (1024, 1024), (2000, 1500), (1500, 2000), (3000, 200), (200, 3000), (37, 41), (100, 7), (7, 100),
            (979, 981), (1400, 1433), (512, 384), (640, 480), (4096, 4096), (13, 13), (980, 378), (377, 979)]
    rows, outs = [], []
    for pi, (mx, mn, st, mp) in enumerate(psets):
        rz = ref_tf.MaxLongEdgeMinShortEdgeResize(max_size=mx, min_size=mn, stride=st, max_pixels=mp)
        for (w, h) in dims:
            for img_num in (1, 2, 5):
                o = rz(Image.new("L", (w, h)), img_num=img_num)
                rows.append([pi, w, h, img_num])
                outs.append(list(o.size))
    rng = np.random.default_rng(77)
    rgb = Image.fromarray(rng.integers(0, 256, (50, 40, 3), dtype=np.uint8))
    gray = Image.fromarray(rng.integers(0, 256, (33, 71), dtype=np.uint8))
    rgba = Image.fromarray(rng.integers(0, 256, (45, 45, 4), dtype=np.uint8), mode="RGBA")
    big = Image.fromarray(np.random.default_rng(78).integers(0, 256, (600, 437, 3), dtype=np.uint8))   # re-made by seed in the test
    t_small, t_vae = ref_tf.ImageTransform(56, 28, 14), ref_tf.ImageTransform(64, 32, 16)
    big_out = ref_tf.ImageTransform(980, 378, 14, max_pixels=2_007_040)(big)
    u8 = torch.from_numpy(rng.integers(0, 256, (3, 50, 40), dtype=np.uint8))
    f32 = torch.from_numpy(rng.random((3, 21, 90), dtype=np.float32))
    np.savez(os.path.join(OUT, "transforms.npz"), **pack(dict(
        psets=torch.tensor(psets), size_in=torch.tensor(rows), size_out=torch.tensor(outs),
        rgb=torch.from_numpy(np.asarray(rgb).copy()), gray=torch.from_numpy(np.asarray(gray).copy()),
        rgba=torch.from_numpy(np.asarray(rgba).copy()),
        rgb_vit=t_small(ref_pil_img2rgb(rgb)), gray_vit=t_small(ref_pil_img2rgb(gray)), rgba_vae=t_vae(ref_pil_img2rgb(rgba)),
        rgb_vit_num3=t_small(ref_pil_img2rgb(rgb), img_num=3),
        big_shape=torch.tensor(list(big_out.shape)),
        big_sha256=hashlib.sha256(big_out.contiguous().numpy().tobytes()).hexdigest(),
        u8_in=u8, u8_resized=t_small.resize_transform(u8), f32_in=f32, f32_resized=t_vae.resize_transform(f32))))
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(" ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
