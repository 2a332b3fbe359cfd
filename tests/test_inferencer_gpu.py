"""Top-level drop-in API on the GPU: InterleaveInferencer.__call__ (PIL / str in, str / PIL out),
VQAInferencer.infer_single and ImageGenerator, against outputs of the REFERENCE's own
InterleaveInferencer run on the same PIL image, prompts, seeds and synthetic weights
(tests/golden/inferencer.npz, oracle/gen_golden.py section H).
Pixel tolerance: t2i (CFG amplification 4*1.5 = 6x): >= 97% of uint8 values within 6 grey levels
and mean abs diff < 2.0; edit (4*2 = 8x amplification on every step, VAE encode + decode): >= 90%
within 6 and mean < 3.0.  The generated image passes through 3-4 guided Euler steps and ~30-60
bf16 VAE stages; rounding-order noise of that size also separates two CPU formulations of
the same math (see tests/test_engine_gpu.py)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import load_golden, NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stack(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from oracle.toy_tokenizer import ToyTokenizer
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.vae import AutoEncoder
    cfg, sd, vae_sd, _ = tiny_weights
    c = UniMedVLConfig.from_dict(cfg)
    model = Bagel(c, lambda n: sd[n], device="cuda")
    vae = AutoEncoder(c, lambda n: vae_sd[n], device="cuda")
    return model, vae, ToyTokenizer(NEW_TOKEN_IDS)


def pixel_close(got, ref, what, min_frac=0.97, max_mean=2.0):
    got, ref = np.asarray(got).astype(np.int32), ref.numpy().astype(np.int32)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    d = np.abs(got - ref)
    frac = float((d <= 6).mean())
    assert frac >= min_frac and d.mean() < max_mean, f"{what}: {100 * frac:.2f}% within 6 levels, mean {d.mean():.3f}, max {d.max()}"


def test_interleave_inferencer_call(stack):
    from unimedvl_amd.inferencer import InterleaveInferencer
    from unimedvl_amd.transforms import ImageTransform
    model, vae, tok = stack
    g = load_golden("inferencer")
    pil = Image.fromarray(g["pil_image"].numpy())
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16), ImageTransform(56, 28, 14), NEW_TOKEN_IDS)
    assert inf() == {"image": None, "text": None}
    und = inf(image=pil, text="5 6 7 8", understanding_output=True, max_think_token_n=6)
    assert und["image"] is None and isinstance(und["text"], str)
    assert und["text"] == g["und_text"], (und["text"], g["und_text"])
    torch.manual_seed(11)
    t2i = inf(text="40 41 42", image_shapes=(64, 64), num_timesteps=4, cfg_text_scale=4.0, cfg_img_scale=1.5,
              cfg_interval=(0.4, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
    assert t2i["text"] is None and isinstance(t2i["image"], Image.Image)
    pixel_close(t2i["image"], g["t2i_image"], "t2i")
    torch.manual_seed(12)
    edit = inf(image=pil, text="9 10", image_shapes=(64, 48), num_timesteps=3, cfg_text_scale=4.0, cfg_img_scale=2.0,
               cfg_interval=(0.0, 1.0), timestep_shift=3.0, cfg_renorm_type="text_channel")
    pixel_close(edit["image"], g["edit_image"], "edit", min_frac=0.90, max_mean=3.0)
    with pytest.raises(ValueError):
        inf(text="1", inference_ver=7)
    with pytest.raises(ValueError):
        inf.interleave_inference([3.14])


def test_vqa_reconstruction_variants(stack):
    """inference_ver=1 through __call__ and the older ver0_1 / ver0 methods (inferencer.py:282-549) against the
    reference's own runs.  ver0_1 / ver0 hard-code both guidance scales to 7.0 (49x amplification of rounding noise
    per guided step), so their pixel tolerance is wider; texts must match exactly."""
    from unimedvl_amd.inferencer import InterleaveInferencer
    from unimedvl_amd.transforms import ImageTransform
    model, vae, tok = stack
    g = load_golden("inferencer")
    pil = Image.fromarray(g["pil_image"].numpy())
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16), ImageTransform(56, 28, 14), NEW_TOKEN_IDS)
    rec = dict(reconstruct_image=True, max_think_token_n=6, num_timesteps=3, cfg_text_scale=4.0, cfg_img_scale=2.0,
               cfg_interval=(0.0, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
    torch.manual_seed(13)
    v1 = inf(image=pil, text="5 6 7 8", inference_ver=1, **rec)
    assert v1["text"] == g["ver1_text"]
    pixel_close(v1["image"], g["ver1_image"], "ver1 reconstruction", min_frac=0.90, max_mean=3.0)
    torch.manual_seed(14)
    v01 = inf.interleave_inference_for_vqa_reconstruction_ver0_1([pil, "5 6 7 8"], **rec)
    assert len(v01) == 2 and v01[0] == g["ver01_text"]
    pixel_close(v01[1], g["ver01_image"], "ver0_1 reconstruction", min_frac=0.80, max_mean=6.0)
    torch.manual_seed(16)
    v0 = inf.interleave_inference_for_vqa_reconstruction_ver0([pil, pil, "5 6 7 8"], **rec)
    assert len(v0) == int(g["ver0_len"]) == 2 and v0[1].size == v01[1].size          # first image only (inferencer.py:466-549)
    assert v0[0] == g["ver0_text"], (v0[0], g["ver0_text"])                          # the reference's own answer ...
    pixel_close(v0[1], g["ver0_image"], "ver0 reconstruction", min_frac=0.80, max_mean=6.0)   # ... and image (both scales 7.0)
    torch.manual_seed(14)   # the VQA context holds a SAMPLED VAE latent of the image: the answer depends on the seed
    assert inf.interleave_inference_for_vqa_reconstruction_ver0([pil, "5 6 7 8"], max_think_token_n=6) == [g["ver01_text"]]


def test_think_mode_and_chat_vs_reference(stack):
    """think=True (inferencer.py:23-28,590-596,617-620: system prompt prefilled into gen / cfg_img contexts, think text generated
    and fed back before the image) and Bagel.chat (bagel.py:1321-1392: images then prompt, ViT-only context, stops at sample
    0's eos) against the REFERENCE's own outputs (tests/golden/inferencer.npz, oracle/gen_golden.py section H)."""
    from unimedvl_amd.data_utils import pil_img2rgb
    from unimedvl_amd.inferencer import GEN_THINK_SYSTEM_PROMPT, VLM_THINK_SYSTEM_PROMPT, InterleaveInferencer
    from unimedvl_amd.transforms import ImageTransform
    model, vae, tok = stack
    g = load_golden("inferencer")
    pil = Image.fromarray(g["pil_image"].numpy())
    assert len(tok.encode(VLM_THINK_SYSTEM_PROMPT)) > 20 and tok.encode(GEN_THINK_SYSTEM_PROMPT) != tok.encode(VLM_THINK_SYSTEM_PROMPT)
    vit_tf = ImageTransform(56, 28, 14)
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16), vit_tf, NEW_TOKEN_IDS)
    und = inf(image=pil, text="5 6 7 8", think=True, understanding_output=True, max_think_token_n=6)
    assert und["image"] is None and und["text"] == g["think_und_text"], (und["text"], g["think_und_text"])
    assert und["text"] != g["und_text"], "the think system prompt must change the context"
    torch.manual_seed(15)
    gen = inf(text="40 41 42", think=True, max_think_token_n=5, image_shapes=(64, 64), num_timesteps=3, cfg_text_scale=4.0,
              cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), timestep_shift=3.0, cfg_renorm_type="global")
    assert gen["text"] == g["think_gen_text"], (gen["text"], g["think_gen_text"])
    pixel_close(gen["image"], g["think_gen_image"], "think + t2i")
    pil2 = Image.fromarray(g["pil_image2"].numpy())
    assert pil2.mode == "L"
    ans = model.chat(tok, dict(NEW_TOKEN_IDS), vit_tf, [pil_img2rgb(pil), pil_img2rgb(pil2)], "5 6 7 8", max_length=8)
    assert ans == g["chat_text"], (ans, g["chat_text"])


def test_batched_call_matches_single_calls(stack):
    """Batch extension: lists in, list of dicts out, per-sample EOS.  Samples are independent
    segments, so a batched greedy VQA run must give each sample the answer its own single-sample call gives
    (decode rows are computed independently and identically; prefill goes through the same kernels)."""
    from unimedvl_amd.inferencer import InterleaveInferencer
    from unimedvl_amd.transforms import ImageTransform
    model, vae, tok = stack
    g = load_golden("inferencer")
    pil = Image.fromarray(g["pil_image"].numpy())
    rng = np.random.default_rng(5)
    pil2 = Image.fromarray(rng.integers(0, 255, (40, 72, 3), dtype=np.uint8))
    inf = InterleaveInferencer(model, vae, tok, ImageTransform(64, 32, 16), ImageTransform(56, 28, 14), NEW_TOKEN_IDS)
    images, prompts = [pil, pil2, pil], ["5 6 7 8", "9 10", "11 12 13 14 15"]
    singles = [inf(image=im, text=tx, understanding_output=True, max_think_token_n=6) for im, tx in zip(images, prompts)]
    batch = inf(image=images, text=prompts, understanding_output=True, max_think_token_n=6)
    assert isinstance(batch, list) and len(batch) == 3
    for b, s in zip(batch, singles):
        assert b["image"] is None and b["text"] == s["text"], (b, s)
    assert batch[0]["text"] == g["und_text"]
    # text-to-image: sample 0 of the batch draws the same init noise as a single call with the same seed
    kw = dict(image_shapes=(64, 64), num_timesteps=4, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0),
              timestep_shift=3.0, cfg_renorm_type="global")
    torch.manual_seed(11)
    out = inf(text=["40 41 42", "43 44"], **kw)
    assert len(out) == 2 and all(o["text"] is None and o["image"].size == (64, 64) for o in out)
    pixel_close(out[0]["image"], g["t2i_image"], "batched t2i, sample 0")
    assert np.abs(np.asarray(out[0]["image"]).astype(int) - np.asarray(out[1]["image"]).astype(int)).mean() > 1.0
    # image editing in a batch (VAE + ViT context, ragged image sizes)
    torch.manual_seed(12)
    ed = inf(image=[pil, pil2], text=["9 10", "9 10"], image_shapes=(64, 48), num_timesteps=3, cfg_text_scale=4.0,
             cfg_img_scale=2.0, cfg_interval=(0.0, 1.0), timestep_shift=3.0, cfg_renorm_type="text_channel")
    # (the padded VAE-encode batch draws its noise in a different order than two single calls: only shapes are checked)
    assert ed[0]["image"].size == (48, 64) and ed[1]["image"].size == (48, 64)
    assert np.asarray(ed[0]["image"]).std() > 0
    with pytest.raises(ValueError, match="one item structure"):
        inf.batch_interleave_inference([[pil, "1 2"], ["3 4"]], understanding_output=True)
    with pytest.raises(ValueError, match="one entry per prompt"):
        inf(image=[pil], text=["1", "2"])


def test_entry_point_classes(stack, tmp_path):
    from unimedvl_amd.interactive_image_generator import ImageGenerator
    from unimedvl_amd.interactive_vqa_inferencer import DEFAULT_CONFIG, VQAInferencer
    model, vae, tok = stack
    assert {"model_path", "target_gpu_device", "temperature", "max_new_tokens", "do_sample", "seed"} <= set(DEFAULT_CONFIG)
    v = VQAInferencer({"max_new_tokens": 5, "do_sample": False})
    with pytest.raises(RuntimeError):
        v.infer_single("x.png", "1 2")
    v.load_model(model=model, tokenizer=tok, new_token_ids=NEW_TOKEN_IDS)
    v.image_transform = __import__("unimedvl_amd.transforms", fromlist=["ImageTransform"]).ImageTransform(56, 28, 14)
    g = load_golden("inferencer")
    p = tmp_path / "img.png"
    Image.fromarray(g["pil_image"].numpy()).save(p)
    res = v.infer_single(str(p), "5 6 7 8")
    assert set(res) == {"answer", "input_image", "time", "image_path", "prompt", "timestamp"}
    assert isinstance(res["answer"], str)
    with pytest.raises(FileNotFoundError):
        v.infer_single(str(tmp_path / "missing.png"), "1")
    gen = ImageGenerator({"vae_transform_size": (64, 32, 16), "vit_transform_size": (56, 28, 14)})
    gen.load_model(model=model, vae_model=vae, tokenizer=tok, new_token_ids=NEW_TOKEN_IDS)
    out = gen.inferencer(text="40 41", image_shapes=(32, 32), num_timesteps=3)
    assert out["image"].size == (32, 32)


def test_load_from_checkpoint_directory(tmp_path):
    """The whole loading path the reference's scripts take (interactive_vqa_inferencer.py:191-268): a checkpoint DIRECTORY
    with llm_config.json / vit_config.json / ema.safetensors / vocab.json / merges.txt / tokenizer_config.json goes in,
    VQAInferencer(config).load_model().infer_single(...) comes out - and gives the answer of an engine built in memory from
    the same tensors.  (Synthetic tiny checkpoint; the tokenizer files are the fixture of tests/golden/tokenizer.)"""
    import json
    import shutil
    from safetensors.torch import save_file
    from conftest import GOLDEN
    from oracle.weights import TINY, make_weights
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.data_utils import add_special_tokens
    from unimedvl_amd.interactive_vqa_inferencer import VQAInferencer
    from unimedvl_amd.tokenizer import Qwen2Tokenizer
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    c = dict(TINY, vocab=704, vit_side=70, max_latent=64)
    sd, _ = make_weights(c, seed=77)
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    json.dump(dict(hidden_size=c["hidden"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                   num_key_value_heads=c["kv_heads"], intermediate_size=c["inter"], vocab_size=c["vocab"], rope_theta=c["rope_theta"],
                   rms_norm_eps=c["rms_eps"], max_position_embeddings=32768), open(ckpt / "llm_config.json", "w"))
    json.dump(dict(hidden_size=c["vit_hidden"], num_hidden_layers=c["vit_layers"] + 1, num_attention_heads=c["vit_heads"],
                   intermediate_size=c["vit_inter"], patch_size=c["patch"]), open(ckpt / "vit_config.json", "w"))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ckpt / "ema.safetensors"))
    for f in ("vocab.json", "merges.txt", "tokenizer_config.json"):
        shutil.copy(os.path.join(GOLDEN, "tokenizer", f), ckpt / f)

    rng = np.random.default_rng(3)
    pil = Image.fromarray(rng.integers(0, 255, (300, 420, 3), dtype=np.uint8))
    v = VQAInferencer({"model_path": str(ckpt), "max_new_tokens": 6, "do_sample": False})
    v.load_model()
    res = v.infer_single(pil, "What abnormality is visible?")
    assert isinstance(res["answer"], str) and res["image_path"] is None

    tok, nt, _ = add_special_tokens(Qwen2Tokenizer.from_pretrained(str(ckpt)))
    assert nt == v.new_token_ids and max(nt.values()) < c["vocab"]
    cfg = UniMedVLConfig.from_dict(c)
    ref = Bagel(cfg, lambda n: sd[n], device="cuda", visual_gen=False)
    want = ref.chat(tok, nt, v.image_transform, [pil], "What abnormality is visible?", max_length=6)
    assert res["answer"] == want

    # ---- the on-disk fast path (packstore.py; the reference's one-time ema_bf16.safetensors conversion,
    # interactive_vqa_inferencer.py:93-161): the first load wrote the device-ready images, the second reads only them
    packed = ckpt / "ema_packed_w-bf16_a-bf16_und.safetensors"
    assert v.load_stats["packed_cache"] == "written" and v.load_stats["from_packed"] == 0 and packed.exists()
    v2 = VQAInferencer({"model_path": str(ckpt), "max_new_tokens": 6, "do_sample": False})
    v2.load_model()
    assert v2.load_stats["packed_cache"] == "hit" and v2.load_stats["built"] == 0 and v2.load_stats["from_packed"] > 10
    assert v2.infer_single(pil, "What abnormality is visible?")["answer"] == want
    w1, w2 = v.model.language_model.w, v2.model.language_model.w
    assert torch.equal(w1.und[1].gate_up.wp, w2.und[1].gate_up.wp) and torch.equal(w1.lm_head.wp, w2.lm_head.wp)
    # fp8 weights get their own file (e4m3 images + scales stored, nothing re-quantised on the second load)
    v8 = VQAInferencer({"model_path": str(ckpt), "max_new_tokens": 6, "do_sample": False, "llm_weight_dtype": "fp8"})
    v8.load_model()
    assert v8.load_stats["packed_cache"] == "written" and (ckpt / "ema_packed_w-fp8_a-bf16_und.safetensors").exists()
    a8 = v8.infer_single(pil, "What abnormality is visible?")["answer"]
    v8b = VQAInferencer({"model_path": str(ckpt), "max_new_tokens": 6, "do_sample": False, "llm_weight_dtype": "fp8"})
    v8b.load_model()
    assert v8b.load_stats["packed_cache"] == "hit" and v8b.load_stats["built"] == 0
    assert torch.equal(v8.model.language_model.w.und[0].qkv.w8, v8b.model.language_model.w.und[0].qkv.w8)
    assert v8b.infer_single(pil, "What abnormality is visible?")["answer"] == a8
    # "packed_cache": False never touches the file; a changed source checkpoint invalidates it
    v3 = VQAInferencer({"model_path": str(ckpt), "packed_cache": False})
    v3.load_model()
    assert v3.load_stats["packed_cache"] == "disabled" and v3.load_stats["from_packed"] == 0
    # a shape mismatch in the file is reported by tensor name, not as a kernel fault later
    bad = dict(sd)
    bad["language_model.model.norm.weight"] = torch.zeros(c["hidden"] + 8, dtype=torch.bfloat16)
    save_file({k: t.contiguous() for k, t in bad.items()}, str(ckpt / "ema.safetensors"))
    with pytest.raises(ValueError, match="language_model.model.norm.weight"):
        VQAInferencer({"model_path": str(ckpt)}).load_model()      # (the packed file of the OLD ema.safetensors is not trusted)
