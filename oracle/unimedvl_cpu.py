"""CPU oracle: a restatement of the reference's forward path.  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product package (unimedvl_amd/) never does.

What it restates (all citations relative to /root/reference/codes/):
  * Qwen2-MoT packed LLM forward   modeling/unimedvl/qwen2_navit.py:525-626,
                                    843-902, 1115-1176
  * RMSNorm / RoPE / MLP            modeling/qwen2/modeling_qwen2.py:89-94,
                                    164-184, 188-220, 234-235
  * SigLIP NaViT ViT                modeling/unimedvl/siglip_navit.py:184-195,
                                    202-244, 255-259, 281-300, 345-371
  * connector / time / pos embed    modeling/unimedvl/modeling_utils.py:87-143
  * Bagel prepare_* / forward_cache_update_* / generate_text / generate_image
    / _forward_flow                 modeling/unimedvl/bagel.py:377-1317
  * FLUX VAE                        modeling/autoencoder.py:38-322
  * latent -> pixels                inferencer.py:234-256

Numerics = the reference run on CPU with bf16 weights under
``torch.autocast("cpu", dtype=torch.bfloat16)`` ("cpu_autocast" policy,
SURVEY.md section 8c): linear/conv/SDPA run in bf16, everything else in the dtype
of its inputs.  The KV cache is held per sample instead of through the
reference's packed index lists; the arithmetic is the same.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against vectors produced by the imported reference itself (oracle/gen_golden.py,
fixtures in tests/golden/).  The reference has no tests or golden vectors of
its own.
"""
import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def linear(x, w, b=None):
    """F.linear under cpu autocast: operands cast to bf16, bf16 result."""
    return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))


def rmsnorm(x, w, eps):
    """modeling_qwen2.py:89-94."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


def rope_cos_sin(position_ids, head_dim, theta, dtype):
    """modeling_qwen2.py:164-184 (default rope, attention_scaling 1)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = (inv_freq[None, :, None].float() @ position_ids[None, None, :].float()).transpose(1, 2)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos()[0].to(dtype), emb.sin()[0].to(dtype)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """modeling_qwen2.py:196-220 with unsqueeze_dim=1 on [T,H,D]."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def attention_segment(q, k, v, causal, impl="sdpa"):
    """One varlen segment of flash_attn_varlen_func. q [Lq,Hq,D], k/v [Lk,Hk,D] bf16.

    impl "sdpa": the stand-in used to run the reference here (oracle/ref_import.py).
    impl "flash": model of the flash-attn kernel itself: fp32 scores and softmax,
    P rounded to bf16 before PV, fp32 accumulate, bf16 out.
    Causal = bottom-right aligned (modeling_qwen2.py:369-372).
    """
    lq, lk = q.shape[0], k.shape[0]
    rep = q.shape[1] // k.shape[1]
    mask = None
    if causal:
        mask = torch.ones(lq, lk, dtype=torch.bool).tril(diagonal=lk - lq)
    if impl == "sdpa":
        qs = q.transpose(0, 1).unsqueeze(0)
        ks = k.transpose(0, 1).unsqueeze(0)
        vs = v.transpose(0, 1).unsqueeze(0)
        if rep > 1:
            ks = ks.repeat_interleave(rep, dim=1)
            vs = vs.repeat_interleave(rep, dim=1)
        o = F.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask)
        return o[0].transpose(0, 1)
    qf = q.float().transpose(0, 1)                                   # [Hq,Lq,D]
    kf = k.float().transpose(0, 1).repeat_interleave(rep, dim=0)
    vf = v.float().transpose(0, 1).repeat_interleave(rep, dim=0)
    s = qf @ kf.transpose(1, 2) / math.sqrt(q.shape[-1])
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    o = (p.to(BF16).float() @ vf) / l
    return o.transpose(0, 1).to(BF16)


def varlen_attention(q, k, v, q_lens, k_lens, causal, impl="sdpa"):
    out = torch.empty_like(q)
    q0 = k0 = 0
    for lq, lk in zip(q_lens, k_lens):
        out[q0:q0 + lq] = attention_segment(q[q0:q0 + lq], k[k0:k0 + lk], v[k0:k0 + lk], causal, impl)
        q0 += lq
        k0 += lk
    return out


class KVCache:
    """Per-layer, per-sample K/V lists; the reference's NaiveCache
    (qwen2_navit.py:207-221) holds the same tokens packed as [sum K, kvh, hd]."""

    def __init__(self, num_layers, num_samples=1):
        self.k = [[None] * num_samples for _ in range(num_layers)]
        self.v = [[None] * num_samples for _ in range(num_layers)]

    def clone(self):
        c = KVCache(len(self.k), len(self.k[0]))
        for l in range(len(self.k)):
            c.k[l] = [None if t is None else t.clone() for t in self.k[l]]
            c.v[l] = [None if t is None else t.clone() for t in self.v[l]]
        return c

    def lens(self):
        return [0 if t is None else t.shape[0] for t in self.k[0]]


# ----------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------
class OracleBagel:
    def __init__(self, cfg, sd, vae_sd=None, attn_impl="sdpa", act_fp8=False):
        self.c = dict(cfg)
        self.sd = sd
        # W8A8 mode of the fp8 extension (no reference counterpart, see oracle/fp8.py): every LLM forward that is not a
        # one-token decode step rounds the inputs of its linear layers per row through e4m3; `sd` must already hold
        # the dequantised weights (oracle.fp8.dequantised_weights)
        self.act_fp8 = act_fp8
        self._act8_now = False
        self.vae_sd = vae_sd
        self.attn_impl = attn_impl
        self.hidden = cfg["hidden"]
        self.head_dim = cfg["hidden"] // cfg["heads"]
        self.latent_downsample = 2 ** (len(cfg["vae_mult"]) - 1) * cfg["latent_patch"]
        self.latent_patch_size = cfg["latent_patch"]
        self.latent_channel = cfg["z_channels"]
        self.max_latent_size = cfg["max_latent"]
        self.vit_patch_size = cfg["patch"]
        self.vit_max_num_patch_per_side = cfg["vit_side"]

    # ---------------------------------------------------------------- LLM
    def _w(self, name):
        return self.sd[name]

    def _lin(self, x, w, b=None):
        """linear() of the LLM layers; in W8A8 mode on the per-row e4m3-rounded activations (umv_quantize_act_fp8)"""
        if self._act8_now and x.shape[0] > 0:
            from oracle.fp8 import quantize_act_rows
            x = quantize_act_rows(x.to(BF16))[2]
        return linear(x, w, b)

    def embed(self, ids):
        return self.sd["language_model.model.embed_tokens.weight"][ids]

    def _attn(self, l, x, query_lens, cos, sin, cache, update, is_causal, mode, text_idx, vae_idx):
        """qwen2_navit.py:525-626."""
        linear = self._lin   # noqa: F841  (shadows the module-level helper: W8A8 mode rounds the inputs)
        c = self.c
        nh, nkv, hd = c["heads"], c["kv_heads"], self.head_dim
        p = f"language_model.model.layers.{l}.self_attn."
        W = self._w
        eps = c["rms_eps"]
        if mode == "und":
            q = linear(x, W(p + "q_proj.weight"), W(p + "q_proj.bias")).view(-1, nh, hd)
            k = linear(x, W(p + "k_proj.weight"), W(p + "k_proj.bias")).view(-1, nkv, hd)
            v = linear(x, W(p + "v_proj.weight"), W(p + "v_proj.bias")).view(-1, nkv, hd)
            q = rmsnorm(q, W(p + "q_norm.weight"), eps)
            k = rmsnorm(k, W(p + "k_norm.weight"), eps)
        else:
            x = x.to(BF16)
            T = x.shape[0]
            q = x.new_zeros((T, nh * hd)); k = x.new_zeros((T, nkv * hd)); v = x.new_zeros((T, nkv * hd))
            xt, xv = x[text_idx], x[vae_idx]
            q[text_idx] = linear(xt, W(p + "q_proj.weight"), W(p + "q_proj.bias"))
            q[vae_idx] = linear(xv, W(p + "q_proj_moe_gen.weight"), W(p + "q_proj_moe_gen.bias"))
            k[text_idx] = linear(xt, W(p + "k_proj.weight"), W(p + "k_proj.bias"))
            k[vae_idx] = linear(xv, W(p + "k_proj_moe_gen.weight"), W(p + "k_proj_moe_gen.bias"))
            v[text_idx] = linear(xt, W(p + "v_proj.weight"), W(p + "v_proj.bias"))
            v[vae_idx] = linear(xv, W(p + "v_proj_moe_gen.weight"), W(p + "v_proj_moe_gen.bias"))
            q = q.view(-1, nh, hd).to(torch.float32)
            k = k.view(-1, nkv, hd).to(torch.float32)
            v = v.view(-1, nkv, hd)
            q[text_idx] = rmsnorm(q[text_idx], W(p + "q_norm.weight"), eps)
            q[vae_idx] = rmsnorm(q[vae_idx], W(p + "q_norm_moe_gen.weight"), eps)
            k[text_idx] = rmsnorm(k[text_idx], W(p + "k_norm.weight"), eps)
            k[vae_idx] = rmsnorm(k[vae_idx], W(p + "k_norm_moe_gen.weight"), eps)
        q, k = apply_rope(q, k, cos, sin)
        q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)

        # merge with the cache per sample: [past ; new]
        mk, mv, klens = [], [], []
        t0 = 0
        for s, lq in enumerate(query_lens):
            ks, vs = k[t0:t0 + lq], v[t0:t0 + lq]
            if cache is not None and cache.k[l][s] is not None:
                ks = torch.cat([cache.k[l][s], ks], 0)
                vs = torch.cat([cache.v[l][s], vs], 0)
            mk.append(ks); mv.append(vs); klens.append(ks.shape[0])
            t0 += lq
        o = varlen_attention(q, torch.cat(mk, 0), torch.cat(mv, 0), query_lens, klens, is_causal, self.attn_impl)
        o = o.reshape(-1, nh * hd)
        if mode == "und":
            o = linear(o, W(p + "o_proj.weight"))
        else:
            o[text_idx] = linear(o[text_idx], W(p + "o_proj.weight"))
            o[vae_idx] = linear(o[vae_idx], W(p + "o_proj_moe_gen.weight"))
        if update:
            for s in range(len(query_lens)):
                cache.k[l][s] = mk[s]
                cache.v[l][s] = mv[s]
        return o

    def _mlp(self, prefix, x):
        linear = self._lin
        W = self._w
        g = linear(x, W(prefix + "gate_proj.weight"))
        u = linear(x, W(prefix + "up_proj.weight"))
        return linear(F.silu(g) * u, W(prefix + "down_proj.weight"))

    def llm_forward(self, seq, query_lens, position_ids, cache, update=True, is_causal=True,
                    mode="und", text_idx=None, vae_idx=None):
        """qwen2_navit.py:1115-1176 + 843-902. Returns final-normed hidden states."""
        c = self.c
        eps = c["rms_eps"]
        W = self._w
        cos, sin = rope_cos_sin(position_ids, self.head_dim, c["rope_theta"], seq.dtype)
        query_lens = [int(x) for x in query_lens]
        self._act8_now = bool(self.act_fp8) and max(query_lens) > 1
        for l in range(c["layers"]):
            p = f"language_model.model.layers.{l}."
            residual = seq
            if mode == "und":
                x = rmsnorm(seq, W(p + "input_layernorm.weight"), eps)
            else:
                x = torch.zeros_like(seq)
                x[text_idx] = rmsnorm(seq[text_idx], W(p + "input_layernorm.weight"), eps)
                x[vae_idx] = rmsnorm(seq[vae_idx], W(p + "input_layernorm_moe_gen.weight"), eps)
            a = self._attn(l, x, query_lens, cos, sin, cache, update, is_causal, mode, text_idx, vae_idx)
            seq = residual + a
            residual = seq
            if mode == "und":
                x = rmsnorm(seq, W(p + "post_attention_layernorm.weight"), eps)
                m = self._mlp(p + "mlp.", x)
            else:
                xt = rmsnorm(seq[text_idx], W(p + "post_attention_layernorm.weight"), eps).to(BF16)
                xv = rmsnorm(seq[vae_idx], W(p + "post_attention_layernorm_moe_gen.weight"), eps).to(BF16)
                m = torch.zeros_like(seq).to(BF16)
                m[text_idx] = self._mlp(p + "mlp.", xt)
                m[vae_idx] = self._mlp(p + "mlp_moe_gen.", xv)
            seq = residual + m
        if mode == "und":
            seq = rmsnorm(seq, W("language_model.model.norm.weight"), eps)
        else:
            out = torch.zeros_like(seq)
            out[text_idx] = rmsnorm(seq[text_idx], W("language_model.model.norm.weight"), eps)
            out[vae_idx] = rmsnorm(seq[vae_idx], W("language_model.model.norm_moe_gen.weight"), eps)
            seq = out
        return seq

    def lm_head(self, h):
        return linear(h, self.sd["language_model.lm_head.weight"])

    # ---------------------------------------------------------------- ViT
    def vit_forward(self, pixels, pos_ids, seqlens):
        """siglip_navit.py:345-371. pixels [N, 3*p*p] fp32 -> [N, vit_hidden] bf16."""
        c = self.c
        W = self._w
        p = "vit_model.vision_model."
        eps = c["ln_eps"]
        nh = c["vit_heads"]
        hd = c["vit_hidden"] // nh
        h = linear(pixels, W(p + "embeddings.patch_embedding.weight"), W(p + "embeddings.patch_embedding.bias"))
        h = h + W(p + "embeddings.position_embedding.weight")[pos_ids]
        lens = [int(x) for x in seqlens]
        for l in range(c["vit_layers"]):
            q_ = p + f"encoder.layers.{l}."
            res = h
            x = F.layer_norm(h, (h.shape[-1],), W(q_ + "layer_norm1.weight"), W(q_ + "layer_norm1.bias"), eps)
            q = linear(x, W(q_ + "self_attn.q_proj.weight"), W(q_ + "self_attn.q_proj.bias")).view(-1, nh, hd)
            k = linear(x, W(q_ + "self_attn.k_proj.weight"), W(q_ + "self_attn.k_proj.bias")).view(-1, nh, hd)
            v = linear(x, W(q_ + "self_attn.v_proj.weight"), W(q_ + "self_attn.v_proj.bias")).view(-1, nh, hd)
            o = varlen_attention(q.to(BF16), k.to(BF16), v.to(BF16), lens, lens, False, self.attn_impl)
            o = linear(o.reshape(o.shape[0], -1), W(q_ + "self_attn.out_proj.weight"), W(q_ + "self_attn.out_proj.bias"))
            h = res + o
            res = h
            x = F.layer_norm(h, (h.shape[-1],), W(q_ + "layer_norm2.weight"), W(q_ + "layer_norm2.bias"), eps)
            x = linear(x, W(q_ + "mlp.fc1.weight"), W(q_ + "mlp.fc1.bias"))
            x = F.gelu(x, approximate="tanh")
            x = linear(x, W(q_ + "mlp.fc2.weight"), W(q_ + "mlp.fc2.bias"))
            h = res + x
        return F.layer_norm(h, (h.shape[-1],), W(p + "post_layernorm.weight"), W(p + "post_layernorm.bias"), eps)

    def connector(self, x):
        """modeling_utils.py:119-123."""
        W = self._w
        x = linear(x, W("connector.fc1.weight"), W("connector.fc1.bias"))
        x = F.gelu(x, approximate="tanh")
        return linear(x, W("connector.fc2.weight"), W("connector.fc2.bias"))

    def time_embed(self, t):
        """modeling_utils.py:87-109. t [N] fp32."""
        half = 128
        freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        W = self._w
        x = linear(emb, W("time_embedder.mlp.0.weight"), W("time_embedder.mlp.0.bias"))
        x = F.silu(x)
        return linear(x, W("time_embedder.mlp.2.weight"), W("time_embedder.mlp.2.bias"))

    # ---------------------------------------------------------------- host prep (bagel.py)
    @staticmethod
    def flattened_position_ids(img_h, img_w, patch, max_side):
        """data_utils.py:53-58."""
        ch = torch.arange(0, img_h // patch)
        cw = torch.arange(0, img_w // patch)
        return (ch[:, None] * max_side + cw).flatten()

    @staticmethod
    def patchify(image, p):
        """data_utils.py:43-50."""
        c, h, w = image.shape
        image = image.reshape(c, h // p, p, w // p, p)
        return torch.einsum("chpwq->hwpqc", image).reshape(-1, p ** 2 * c)

    def update_text(self, cache, kvlens, ropes, token_id_lists):
        """prepare_prompts + forward_cache_update_text (bagel.py:377-458).
        token_id_lists already wrapped with bos/eos by the caller."""
        ids, pos, lens = [], [], []
        for t, r in zip(token_id_lists, ropes):
            ids += t
            pos += list(range(r, r + len(t)))
            lens.append(len(t))
        seq = self.embed(torch.tensor(ids, dtype=torch.long))
        self.llm_forward(seq, lens, torch.tensor(pos, dtype=torch.long), cache, True, True, "und")
        return [k + n for k, n in zip(kvlens, lens)], [r + n for r, n in zip(ropes, lens)]

    def update_vit(self, cache, kvlens, ropes, image_tensors, new_token_ids):
        """prepare_vit_images + forward_cache_update_vit (bagel.py:460-615)."""
        toks, vpos, vlens, seqlens, pos = [], [], [], [], []
        text_ids, text_idx, vit_idx = [], [], []
        cur = 0
        for img, r in zip(image_tensors, ropes):
            text_ids.append(new_token_ids["start_of_image"]); text_idx.append(cur); cur += 1
            t = self.patchify(img, self.vit_patch_size)
            toks.append(t)
            vpos.append(self.flattened_position_ids(img.shape[1], img.shape[2], self.vit_patch_size,
                                                    self.vit_max_num_patch_per_side))
            n = t.shape[0]
            vlens.append(n)
            vit_idx += list(range(cur, cur + n)); cur += n
            text_ids.append(new_token_ids["end_of_image"]); text_idx.append(cur); cur += 1
            pos += [r] * (n + 2)
            seqlens.append(n + 2)
        emb = self.embed(torch.tensor(text_ids, dtype=torch.long))
        seq = emb.new_zeros((sum(seqlens), self.hidden))
        seq[torch.tensor(text_idx)] = emb
        vpos = torch.cat(vpos, 0)
        ve = self.vit_forward(torch.cat(toks, 0), vpos, vlens)
        ve = self.connector(ve) + self.sd["vit_pos_embed.pos_embed"][vpos]
        seq[torch.tensor(vit_idx)] = ve.to(seq.dtype)
        self.llm_forward(seq, seqlens, torch.tensor(pos, dtype=torch.long), cache, True, False, "und")
        return [k + n for k, n in zip(kvlens, seqlens)], [r + 1 for r in ropes]

    def update_vae(self, cache, kvlens, ropes, image_tensors, new_token_ids, noise=None, timestep=0):
        """prepare_vae_images + forward_cache_update_vae (bagel.py:617-806)."""
        p = self.latent_patch_size
        text_ids, text_idx, vae_idx, seqlens, pos, vpos, shapes = [], [], [], [], [], [], []
        cur = 0
        for img, r in zip(image_tensors, ropes):
            text_ids.append(new_token_ids["start_of_image"]); text_idx.append(cur); cur += 1
            H, W = img.shape[1:]
            h, w = H // self.latent_downsample, W // self.latent_downsample
            shapes.append((h, w))
            vpos.append(self.flattened_position_ids(H, W, self.latent_downsample, self.max_latent_size))
            vae_idx += list(range(cur, cur + h * w)); cur += h * w
            text_ids.append(new_token_ids["end_of_image"]); text_idx.append(cur); cur += 1
            pos += [r] * (h * w + 2)
            seqlens.append(h * w + 2)
        mh = max(i.shape[1] for i in image_tensors); mw = max(i.shape[2] for i in image_tensors)
        padded = torch.zeros((len(image_tensors), 3, mh, mw))
        for i, img in enumerate(image_tensors):
            padded[i, :, :img.shape[1], :img.shape[2]] = img
        emb = self.embed(torch.tensor(text_ids, dtype=torch.long))
        seq = emb.new_zeros((sum(seqlens), self.hidden))
        seq[torch.tensor(text_idx)] = emb
        lat = self.vae_encode(padded, noise)
        rows = []
        for z, (h, w) in zip(lat, shapes):
            z = z[:, :h * p, :w * p].reshape(self.latent_channel, h, p, w, p)
            rows.append(torch.einsum("chpwq->hwpqc", z).reshape(-1, p * p * self.latent_channel))
        packed = torch.cat(rows, 0)
        vpos = torch.cat(vpos, 0)
        te = self.time_embed(torch.tensor([timestep]))
        x = linear(packed, self.sd["vae2llm.weight"], self.sd["vae2llm.bias"]) + te + self.sd["latent_pos_embed.pos_embed"][vpos]
        seq[torch.tensor(vae_idx)] = x.to(seq.dtype)
        self.llm_forward(seq, seqlens, torch.tensor(pos, dtype=torch.long), cache, True, False, "gen",
                         torch.tensor(text_idx), torch.tensor(vae_idx))
        return [k + n for k, n in zip(kvlens, seqlens)], [r + 1 for r in ropes]

    # ---------------------------------------------------------------- text generation
    def generate_text(self, cache, ropes, start_token, max_length, end_token_id=None, return_logits=False):
        """prepare_start_tokens + generate_text greedy (bagel.py:1213-1317)."""
        B = len(ropes)
        cur = torch.full((B,), start_token, dtype=torch.long)
        pos = torch.tensor(ropes, dtype=torch.long)
        out, logits_all = [], []
        step = 0
        while step < max_length:
            out.append(cur)
            seq = self.embed(cur)
            h = self.llm_forward(seq, [1] * B, pos, cache, True, True, "und")
            logits = self.lm_head(h)
            logits_all.append(logits)
            cur = torch.argmax(logits, dim=-1)
            pos = pos + 1
            step += 1
            if end_token_id is not None and cur[0] == end_token_id:
                break
        ids = torch.stack(out, 0)
        return (ids, torch.stack(logits_all, 0)) if return_logits else ids

    # ---------------------------------------------------------------- image generation
    def _flow_inputs(self, image_sizes, ropes, new_token_ids):
        text_idx, vae_idx, seqlens, pos, vpos = [], [], [], [], []
        cur = 0
        for (H, W), r in zip(image_sizes, ropes):
            text_idx.append(cur); cur += 1
            h, w = H // self.latent_downsample, W // self.latent_downsample
            vpos.append(self.flattened_position_ids(H, W, self.latent_downsample, self.max_latent_size))
            vae_idx += list(range(cur, cur + h * w)); cur += h * w
            text_idx.append(cur); cur += 1
            pos += [r] * (h * w + 2)
            seqlens.append(h * w + 2)
        text_ids = [new_token_ids["start_of_image"], new_token_ids["end_of_image"]] * len(image_sizes)
        return (torch.tensor(text_ids), torch.tensor(text_idx), torch.tensor(vae_idx), seqlens,
                torch.tensor(pos, dtype=torch.long), torch.cat(vpos, 0))

    def forward_flow(self, x_t, t, fi, cache, cfg_text=None, cfg_img=None, cfg_text_scale=1.0,
                     cfg_img_scale=1.0, cfg_renorm_min=0.0, cfg_renorm_type="global"):
        """bagel.py:989-1211.  cfg_text / cfg_img = (cache, position_ids)."""
        text_ids, text_idx, vae_idx, seqlens, pos, vpos = fi
        emb = self.embed(text_ids)
        seq = emb.new_zeros((sum(seqlens), self.hidden))
        seq[text_idx] = emb
        te = self.time_embed(torch.full((x_t.shape[0],), float(t)))
        x = linear(x_t, self.sd["vae2llm.weight"], self.sd["vae2llm.bias"]) + te + self.sd["latent_pos_embed.pos_embed"][vpos]
        seq[vae_idx] = x.to(seq.dtype)
        W2, b2 = self.sd["llm2vae.weight"], self.sd["llm2vae.bias"]

        def one(c, p):
            h = self.llm_forward(seq, seqlens, p, c, False, False, "gen", text_idx, vae_idx)
            return linear(h, W2, b2)[vae_idx]
        v_t = one(cache, pos)
        if cfg_text_scale > 1.0:
            v_c = one(*cfg_text)
        if cfg_img_scale > 1.0:
            v_i = one(*cfg_img)
        if cfg_text_scale > 1.0:
            if cfg_renorm_type == "text_channel":
                v_text_ = v_c + cfg_text_scale * (v_t - v_c)
                n0 = torch.norm(v_t, dim=-1, keepdim=True)
                n1 = torch.norm(v_text_, dim=-1, keepdim=True)
                scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
                v_text = v_text_ * scale
                v_t = v_i + cfg_img_scale * (v_text - v_i) if cfg_img_scale > 1.0 else v_text
            else:
                v_text_ = v_c + cfg_text_scale * (v_t - v_c)
                v_ = v_i + cfg_img_scale * (v_text_ - v_i) if cfg_img_scale > 1.0 else v_text_
                if cfg_renorm_type == "global":
                    n0, n1 = torch.norm(v_t), torch.norm(v_)
                elif cfg_renorm_type == "channel":
                    n0 = torch.norm(v_t, dim=-1, keepdim=True)
                    n1 = torch.norm(v_, dim=-1, keepdim=True)
                else:
                    raise NotImplementedError(cfg_renorm_type)
                scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
                v_t = v_ * scale
        return v_t

    def generate_image(self, cache, ropes, image_sizes, init_noise, new_token_ids, num_timesteps=24,
                       timestep_shift=1.0, cfg_interval=(0.0, 1.0), cfg_text_scale=1.0, cfg_text=None,
                       cfg_img_scale=1.0, cfg_img=None, cfg_renorm_min=0.0, cfg_renorm_type="global",
                       trace=None):
        """bagel.py:901-986. cfg_text / cfg_img = (cache, ropes)."""
        fi = self._flow_inputs(image_sizes, ropes, new_token_ids)
        seqlens = fi[3]

        def cfg_pack(c):
            if c is None:
                return None
            cc, rr = c
            pos = []
            for n, r in zip(seqlens, rr):
                pos += [r] * n
            return (cc, torch.tensor(pos, dtype=torch.long))
        cfg_text, cfg_img = cfg_pack(cfg_text), cfg_pack(cfg_img)
        x_t = init_noise
        ts = torch.linspace(1, 0, num_timesteps)
        ts = timestep_shift * ts / (1 + (timestep_shift - 1) * ts)
        dts = ts[:-1] - ts[1:]
        ts = ts[:-1]
        for i, t in enumerate(ts):
            if t > cfg_interval[0] and t <= cfg_interval[1]:
                s_t, s_i = cfg_text_scale, cfg_img_scale
            else:
                s_t, s_i = 1.0, 1.0
            v_t = self.forward_flow(x_t, t, fi, cache, cfg_text, cfg_img, s_t, s_i, cfg_renorm_min, cfg_renorm_type)
            x_t = x_t - v_t * dts[i]
            if trace is not None:
                trace.append(x_t.clone())
        return x_t.split([n - 2 for n in seqlens])

    # ---------------------------------------------------------------- VAE
    def _conv(self, x, name, stride=1, padding=1):
        return F.conv2d(x.to(BF16), self.vae_sd[name + ".weight"], self.vae_sd[name + ".bias"],
                        stride=stride, padding=padding)

    def _gn(self, x, name):
        return F.group_norm(x, 32, self.vae_sd[name + ".weight"], self.vae_sd[name + ".bias"], 1e-6)

    def _resblock(self, x, p):
        cin = self.vae_sd[p + "conv1.weight"].shape[1]
        cout = self.vae_sd[p + "conv1.weight"].shape[0]
        h = self._gn(x, p + "norm1")
        h = h * torch.sigmoid(h)
        h = self._conv(h, p + "conv1")
        h = self._gn(h, p + "norm2")
        h = h * torch.sigmoid(h)
        h = self._conv(h, p + "conv2")
        if cin != cout:
            x = self._conv(x, p + "nin_shortcut", padding=0)
        return x + h

    def _attnblock(self, x, p):
        h = self._gn(x, p + "norm")
        q = self._conv(h, p + "q", padding=0)
        k = self._conv(h, p + "k", padding=0)
        v = self._conv(h, p + "v", padding=0)
        b, c, hh, ww = q.shape
        f = lambda t: t.reshape(b, c, hh * ww).transpose(1, 2).unsqueeze(1).contiguous()
        o = F.scaled_dot_product_attention(f(q), f(k), f(v))
        o = o.squeeze(1).transpose(1, 2).reshape(b, c, hh, ww)
        return x + self._conv(o, p + "proj_out", padding=0)

    def vae_encoder(self, x):
        """autoencoder.py:169-187."""
        nlev, nres = len(self.c["vae_mult"]), self.c["vae_res"]
        h = self._conv(x, "encoder.conv_in")
        for lvl in range(nlev):
            for b in range(nres):
                h = self._resblock(h, f"encoder.down.{lvl}.block.{b}.")
            if lvl != nlev - 1:
                h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
                h = self._conv(h, f"encoder.down.{lvl}.downsample.conv", stride=2, padding=0)
        h = self._resblock(h, "encoder.mid.block_1.")
        h = self._attnblock(h, "encoder.mid.attn_1.")
        h = self._resblock(h, "encoder.mid.block_2.")
        h = self._gn(h, "encoder.norm_out")
        h = h * torch.sigmoid(h)
        return self._conv(h, "encoder.conv_out")

    def vae_encode(self, x, noise=None):
        """autoencoder.py:300-303 (+ DiagonalGaussian :266-272); noise injected."""
        z = self.vae_encoder(x)
        mean, logvar = torch.chunk(z, 2, dim=1)
        std = torch.exp(0.5 * logvar)
        if noise is None:
            noise = torch.randn_like(mean)
        z = mean + std * noise.to(mean.dtype)
        return self.c["scale_factor"] * (z - self.c["shift_factor"])

    def vae_decode(self, z):
        """autoencoder.py:305-307, 240-257."""
        nlev, nres = len(self.c["vae_mult"]), self.c["vae_res"]
        z = z.to(BF16)
        z = z / self.c["scale_factor"] + self.c["shift_factor"]
        h = self._conv(z, "decoder.conv_in")
        h = self._resblock(h, "decoder.mid.block_1.")
        h = self._attnblock(h, "decoder.mid.attn_1.")
        h = self._resblock(h, "decoder.mid.block_2.")
        for lvl in reversed(range(nlev)):
            for b in range(nres + 1):
                h = self._resblock(h, f"decoder.up.{lvl}.block.{b}.")
            if lvl != 0:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self._conv(h, f"decoder.up.{lvl}.upsample.conv")
        h = self._gn(h, "decoder.norm_out")
        h = h * torch.sigmoid(h)
        return self._conv(h, "decoder.conv_out")

    def decode_image(self, latent, image_shape):
        """inferencer.py:234-256 -> uint8 [H,W,3]."""
        H, W = image_shape
        h, w = H // self.latent_downsample, W // self.latent_downsample
        p, c = self.latent_patch_size, self.latent_channel
        latent = latent.reshape(1, h, w, p, p, c)
        latent = torch.einsum("nhwpqc->nchpwq", latent).reshape(1, c, h * p, w * p)
        image = self.vae_decode(latent.to(BF16))
        image = (image * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255
        return image.to(torch.uint8)
