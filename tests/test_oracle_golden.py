"""The CPU oracle (oracle/unimedvl_cpu.py) against golden vectors produced by the
imported reference (oracle/gen_golden.py).  Runs without a GPU."""
import pytest
import torch

from conftest import load_golden, NEW_TOKEN_IDS
from oracle.unimedvl_cpu import OracleBagel, KVCache

BOS, EOS = NEW_TOKEN_IDS["bos_token_id"], NEW_TOKEN_IDS["eos_token_id"]


def wrap(ids):
    return [BOS] + [int(i) for i in ids] + [EOS]


@pytest.fixture(scope="module")
def oracle(tiny_weights):
    cfg, sd, vae_sd, _ = tiny_weights
    return OracleBagel(cfg, sd, vae_sd, attn_impl="sdpa")


def test_weights_digest(tiny_weights):
    g = load_golden("vit")
    assert g["weights_sha"] == tiny_weights[3], "RNG drift: synthetic weights differ from the ones the goldens were made with"


def test_vit_bit_exact(oracle):
    g = load_golden("vit")
    img = g["image"]
    px = oracle.patchify(img, oracle.vit_patch_size)
    pos = oracle.flattened_position_ids(img.shape[1], img.shape[2], 14, oracle.vit_max_num_patch_per_side)
    out = oracle.vit_forward(px, pos, [px.shape[0]])
    assert torch.equal(out, g["vit_out"])
    conn = oracle.connector(out) + oracle.sd["vit_pos_embed.pos_embed"][pos]
    assert torch.equal(conn, g["connector_out"])


def test_vqa_b1_bit_exact(oracle):
    g = load_golden("vqa_b1")
    L = oracle.c["layers"]
    cache = KVCache(L, 1)
    kvl, rope = oracle.update_vit(cache, [0], [0], [g["image"]], NEW_TOKEN_IDS)
    assert torch.equal(cache.k[0][0], g["k0_after_vit"])
    assert torch.equal(cache.v[L - 1][0], g["vL_after_vit"])
    kvl, rope = oracle.update_text(cache, kvl, rope, [wrap(g["prompt_ids"])])
    assert torch.equal(cache.k[0][0], g["k0_after_text"])
    assert torch.equal(cache.v[L - 1][0], g["vL_after_text"])
    assert kvl == g["kv_lens"].tolist() and rope == g["ropes"].tolist()
    ids, logits = oracle.generate_text(cache, rope, BOS, 8, return_logits=True)
    assert torch.equal(logits, g["logits"])
    assert torch.equal(ids, g["token_ids"])


def test_vqa_b2_ragged_bit_exact(oracle):
    g = load_golden("vqa_b2")
    cache = KVCache(oracle.c["layers"], 2)
    kvl, rope = oracle.update_vit(cache, [0, 0], [0, 0], [g["image0"], g["image1"]], NEW_TOKEN_IDS)
    kvl, rope = oracle.update_text(cache, kvl, rope, [wrap(g["prompt0"]), wrap(g["prompt1"])])
    assert kvl == g["kv_lens"].tolist() and rope == g["ropes"].tolist()
    ids, logits = oracle.generate_text(cache, rope, BOS, 6, return_logits=True)
    assert torch.equal(logits, g["logits"])
    assert torch.equal(ids, g["token_ids"])


def test_vae_bit_exact(oracle):
    g = load_golden("vae")
    assert torch.equal(oracle.vae_decode(g["z"]), g["decoded"])
    assert torch.equal(oracle.vae_encode(g["image"], g["enc_noise"]), g["encoded"])


def _t2i_contexts(oracle, g):
    L = oracle.c["layers"]
    gen = KVCache(L, 1)
    kvl, rope = oracle.update_text(gen, [0], [0], [wrap(g["prompt_ids"])])
    cfg_text = (KVCache(L, 1), [0])
    cfg_img = (gen.clone(), list(rope))
    return gen, rope, cfg_text, cfg_img


@pytest.mark.parametrize("rtype", ["global", "channel", "text_channel"])
def test_t2i_latents_bit_exact(oracle, rtype):
    g = load_golden("t2i")
    gen, rope, cfg_text, cfg_img = _t2i_contexts(oracle, g)
    H, W = g["image_shape"].tolist()
    lat = oracle.generate_image(gen, rope, [(H, W)], g["init_noise"], NEW_TOKEN_IDS, num_timesteps=6,
                                timestep_shift=3.0, cfg_interval=(0.4, 1.0), cfg_text_scale=4.0, cfg_text=cfg_text,
                                cfg_img_scale=1.5, cfg_img=cfg_img, cfg_renorm_type=rtype)
    assert torch.equal(lat[0], g["latent_" + rtype])


def test_t2i_nocfg_and_pixels(oracle):
    g = load_golden("t2i")
    gen, rope, _, _ = _t2i_contexts(oracle, g)
    H, W = g["image_shape"].tolist()
    lat = oracle.generate_image(gen, rope, [(H, W)], g["init_noise"], NEW_TOKEN_IDS, num_timesteps=4, timestep_shift=3.0)
    assert torch.equal(lat[0], g["latent_nocfg"])
    px = oracle.decode_image(g["latent_global"], (H, W))
    assert torch.equal(px, g["pixels_u8"])


def test_edit_prefill_bit_exact(oracle):
    g = load_golden("edit_prefill")
    L = oracle.c["layers"]
    cache = KVCache(L, 1)
    kvl, rope = oracle.update_vae(cache, [0], [0], [g["image"]], NEW_TOKEN_IDS, noise=g["enc_noise"])
    assert kvl == g["kv_lens"].tolist() and rope == g["ropes"].tolist()
    assert torch.equal(cache.k[0][0], g["k0"])
    assert torch.equal(cache.v[L - 1][0], g["vL"])


@pytest.fixture(scope="module")
def oracle_flash(tiny_weights):
    cfg, sd, vae_sd, _ = tiny_weights
    return OracleBagel(cfg, sd, vae_sd, attn_impl="flash")


def test_flash_branch_stays_within_its_quoted_spread_of_the_pinned_branch(oracle_flash):
    """The golden vectors pin attn_impl="sdpa" (what the imported reference ran on CPU).  The full-width / full-depth GPU tests compare
    the engine with attn_impl="flash" (P rounded to bf16 before PV: the model of the flash-attn kernel the reference runs on a GPU) - a
    branch no golden pins bit for bit.  This ties it to the pinned branch on the reference's own vectors: same
    greedy ids, logits within 0.02, guided latents within the 0.06-0.09 max / 0.014-0.018 mean that tests/test_engine_gpu.py quotes as its
    yardstick (measured here: 0.0625 / 0.0688 / 0.0859 for global / channel / text_channel renorm on latents of range 4.9-5.5)."""
    o = oracle_flash
    g = load_golden("vqa_b1")
    cache = KVCache(o.c["layers"], 1)
    kvl, rope = o.update_vit(cache, [0], [0], [g["image"]], NEW_TOKEN_IDS)
    kvl, rope = o.update_text(cache, kvl, rope, [wrap(g["prompt_ids"])])
    ids, logits = o.generate_text(cache, rope, BOS, 8, return_logits=True)
    assert torch.equal(ids, g["token_ids"])
    assert float((logits.float() - g["logits"].float()).abs().max()) <= 0.04
    g = load_golden("t2i")
    H, W = g["image_shape"].tolist()
    for rtype, bound in (("global", 0.07), ("channel", 0.08), ("text_channel", 0.09)):
        gen, rope, cfg_text, cfg_img = _t2i_contexts(o, g)
        lat = o.generate_image(gen, rope, [(H, W)], g["init_noise"], NEW_TOKEN_IDS, num_timesteps=6, timestep_shift=3.0, cfg_interval=(0.4, 1.0),
                               cfg_text_scale=4.0, cfg_text=cfg_text, cfg_img_scale=1.5, cfg_img=cfg_img, cfg_renorm_type=rtype)
        d = (lat[0].float() - g["latent_" + rtype].float()).abs()
        assert 0.0 < float(d.max()) <= bound and float(d.mean()) <= 0.02, (rtype, float(d.max()), float(d.mean()))
