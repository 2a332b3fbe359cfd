"""State-dict name -> shape table of the reference modules.

Used to validate checkpoints at load time and to synthesise random weights for
benches.  Names follow the module paths of codes/modeling/unimedvl/{bagel,qwen2_navit,
siglip_navit}.py and codes/modeling/autoencoder.py:122-257.
"""
from .config import UniMedVLConfig


def _c(cfg):
    return cfg.to_dict() if isinstance(cfg, UniMedVLConfig) else dict(cfg)


def llm_shapes(c):
    H, I, V = c["hidden"], c["inter"], c["vocab"]
    hd = H // c["heads"]
    kv = c["kv_heads"] * hd
    s = {"language_model.model.embed_tokens.weight": (V, H)}
    for i in range(c["layers"]):
        p = f"language_model.model.layers.{i}."
        for suf in ("", "_moe_gen"):
            s[p + f"self_attn.q_proj{suf}.weight"] = (H, H)
            s[p + f"self_attn.q_proj{suf}.bias"] = (H,)
            s[p + f"self_attn.k_proj{suf}.weight"] = (kv, H)
            s[p + f"self_attn.k_proj{suf}.bias"] = (kv,)
            s[p + f"self_attn.v_proj{suf}.weight"] = (kv, H)
            s[p + f"self_attn.v_proj{suf}.bias"] = (kv,)
            s[p + f"self_attn.o_proj{suf}.weight"] = (H, H)
            s[p + f"self_attn.q_norm{suf}.weight"] = (hd,)
            s[p + f"self_attn.k_norm{suf}.weight"] = (hd,)
            s[p + f"mlp{suf}.gate_proj.weight"] = (I, H)
            s[p + f"mlp{suf}.up_proj.weight"] = (I, H)
            s[p + f"mlp{suf}.down_proj.weight"] = (H, I)
            s[p + f"input_layernorm{suf}.weight"] = (H,)
            s[p + f"post_attention_layernorm{suf}.weight"] = (H,)
    s["language_model.model.norm.weight"] = (H,)
    s["language_model.model.norm_moe_gen.weight"] = (H,)
    s["language_model.lm_head.weight"] = (V, H)
    return s


def vit_shapes(c):
    h, i = c["vit_hidden"], c["vit_inter"]
    p = "vit_model.vision_model."
    s = {
        p + "embeddings.patch_embedding.weight": (h, 3 * c["patch"] ** 2),
        p + "embeddings.patch_embedding.bias": (h,),
        p + "embeddings.position_embedding.weight": (c["vit_side"] ** 2, h),
    }
    for l in range(c["vit_layers"]):
        q = p + f"encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[q + f"self_attn.{n}.weight"] = (h, h)
            s[q + f"self_attn.{n}.bias"] = (h,)
        s[q + "layer_norm1.weight"] = (h,)
        s[q + "layer_norm1.bias"] = (h,)
        s[q + "layer_norm2.weight"] = (h,)
        s[q + "layer_norm2.bias"] = (h,)
        s[q + "mlp.fc1.weight"] = (i, h)
        s[q + "mlp.fc1.bias"] = (i,)
        s[q + "mlp.fc2.weight"] = (h, i)
        s[q + "mlp.fc2.bias"] = (h,)
    s[p + "post_layernorm.weight"] = (h,)
    s[p + "post_layernorm.bias"] = (h,)
    return s


def glue_shapes(c):
    H, h = c["hidden"], c["vit_hidden"]
    pd = c["latent_patch"] ** 2 * c["z_channels"]
    return {
        "connector.fc1.weight": (H, h), "connector.fc1.bias": (H,),
        "connector.fc2.weight": (H, H), "connector.fc2.bias": (H,),
        "time_embedder.mlp.0.weight": (H, 256), "time_embedder.mlp.0.bias": (H,),
        "time_embedder.mlp.2.weight": (H, H), "time_embedder.mlp.2.bias": (H,),
        "vae2llm.weight": (H, pd), "vae2llm.bias": (H,),
        "llm2vae.weight": (pd, H), "llm2vae.bias": (pd,),
    }


def _resblock(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "nin_shortcut.weight"] = (cout, cin, 1, 1); s[p + "nin_shortcut.bias"] = (cout,)


def _attnblock(s, p, c):
    s[p + "norm.weight"] = (c,); s[p + "norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        s[p + n + ".weight"] = (c, c, 1, 1); s[p + n + ".bias"] = (c,)


def vae_shapes(c):
    """autoencoder.py:122-257 module tree."""
    ch, mult, nres, z = c["vae_ch"], tuple(c["vae_mult"]), c["vae_res"], c["z_channels"]
    s = {}
    # encoder
    s["encoder.conv_in.weight"] = (ch, 3, 3, 3); s["encoder.conv_in.bias"] = (ch,)
    in_mult = (1,) + mult
    block_in = ch
    for lvl in range(len(mult)):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        for b in range(nres):
            _resblock(s, f"encoder.down.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            s[f"encoder.down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"encoder.down.{lvl}.downsample.conv.bias"] = (block_in,)
    _resblock(s, "encoder.mid.block_1.", block_in, block_in)
    _attnblock(s, "encoder.mid.attn_1.", block_in)
    _resblock(s, "encoder.mid.block_2.", block_in, block_in)
    s["encoder.norm_out.weight"] = (block_in,); s["encoder.norm_out.bias"] = (block_in,)
    s["encoder.conv_out.weight"] = (2 * z, block_in, 3, 3); s["encoder.conv_out.bias"] = (2 * z,)
    # decoder
    block_in = ch * mult[-1]
    s["decoder.conv_in.weight"] = (block_in, z, 3, 3); s["decoder.conv_in.bias"] = (block_in,)
    _resblock(s, "decoder.mid.block_1.", block_in, block_in)
    _attnblock(s, "decoder.mid.attn_1.", block_in)
    _resblock(s, "decoder.mid.block_2.", block_in, block_in)
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        for b in range(nres + 1):
            _resblock(s, f"decoder.up.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            s[f"decoder.up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"decoder.up.{lvl}.upsample.conv.bias"] = (block_in,)
    s["decoder.norm_out.weight"] = (block_in,); s["decoder.norm_out.bias"] = (block_in,)
    s["decoder.conv_out.weight"] = (3, block_in, 3, 3); s["decoder.conv_out.bias"] = (3,)
    return s


def all_shapes(cfg):
    c = _c(cfg)
    s = {}
    s.update(llm_shapes(c))
    s.update(vit_shapes(c))
    s.update(glue_shapes(c))
    s["vit_pos_embed.pos_embed"] = (c["vit_side"] ** 2, c["hidden"])
    s["latent_pos_embed.pos_embed"] = (c["max_latent"] ** 2, c["hidden"])
    return s
