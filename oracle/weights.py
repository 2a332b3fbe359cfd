"""Deterministic synthetic weights for the parity tests.  TEST INFRASTRUCTURE.

The state-dict KEYS and SHAPES are those of the reference modules
(reference codes/interactive_image_generator.py:197-275
builds them; names follow module paths in
codes/modeling/unimedvl/{bagel,qwen2_navit,siglip_navit}.py and
codes/modeling/autoencoder.py).  Values come from a seeded torch.Generator,
one tensor at a time in a fixed key order, so the reference (in this
container), the oracle and the HIP path all see the same bits.  A sha256 of
the bytes is stored in every golden fixture to detect RNG drift.
"""
import hashlib
import math

import numpy as np
import torch

TINY = dict(
    hidden=256, layers=2, heads=2, kv_heads=1, inter=384, vocab=320,
    vit_hidden=144, vit_layers=2, vit_heads=2, vit_inter=208, patch=14, vit_side=8,
    max_latent=8, vae_ch=32, vae_mult=(1, 2, 4, 4), vae_res=1, z_channels=16,
    rope_theta=1e6, rms_eps=1e-6, ln_eps=1e-6, latent_patch=2,
    scale_factor=0.3611, shift_factor=0.1159,
)

FULL = dict(  # assumed BAGEL-7B-MoT dims, SURVEY.md section 0
    hidden=3584, layers=28, heads=28, kv_heads=4, inter=18944, vocab=152064,
    vit_hidden=1152, vit_layers=26, vit_heads=16, vit_inter=4304, patch=14, vit_side=70,
    max_latent=64, vae_ch=128, vae_mult=(1, 2, 4, 4), vae_res=2, z_channels=16,
    rope_theta=1e6, rms_eps=1e-6, ln_eps=1e-6, latent_patch=2,
    scale_factor=0.3611, shift_factor=0.1159,
)


def sincos_2d(embed_dim, grid_size):
    """Frozen 2-D sin-cos table, restating modeling_utils.py:23-65."""
    def one_d(dim, pos):
        omega = np.arange(dim // 2, dtype=np.float64)
        omega /= dim / 2.0
        omega = 1.0 / 10000 ** omega
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float()


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


def _fill(shapes, seed, std_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k in shapes:  # insertion order is the fixed key order
        shp = shapes[k]
        if len(shp) == 1:
            if k.endswith("bias"):
                t = 0.05 * torch.randn(shp, generator=g)
            else:  # norm gains
                t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = int(np.prod(shp[1:]))
            # keep activations O(1) so parity checks see signal (random-init
            # rule of the reference, modeling_qwen2.py:597-606, is std 0.02;
            # that makes tiny models degenerate, hence fan-in scaling here)
            t = torch.randn(shp, generator=g) * (std_scale / math.sqrt(fan_in))
            if "embed_tokens" in k or "position_embedding" in k:
                t = torch.randn(shp, generator=g) * 0.5
        sd[k] = t.to(torch.bfloat16)
    return sd


def make_weights(c, seed=1234):
    """Returns (model_sd, vae_sd), bf16, keys as the reference state dicts."""
    shapes = {}
    shapes.update(llm_shapes(c))
    shapes.update(vit_shapes(c))
    shapes.update(glue_shapes(c))
    sd = _fill(shapes, seed)
    sd["vit_pos_embed.pos_embed"] = sincos_2d(c["hidden"], c["vit_side"]).to(torch.bfloat16)
    sd["latent_pos_embed.pos_embed"] = sincos_2d(c["hidden"], c["max_latent"]).to(torch.bfloat16)
    vae_sd = _fill(vae_shapes(c), seed + 7)
    return sd, vae_sd


def digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().view(torch.int16).numpy().tobytes())
    return h.hexdigest()
