"""Device-side weight store.

Tensors are fetched by their reference state-dict names (SURVEY.md section 3.4 /
codes/interactive_image_generator.py:197-275) through a ``get(name) -> Tensor``
callable, moved to the GPU and re-tiled for the MFMA GEMM kernels
(umv_pack_weight_bf16).  q/k/v projections are fused into one [nq+2nkv)*hd, H]
weight and gate/up into one interleaved SwiGLU weight, per expert.
"""
import math

import torch

from . import ops
from .config import UniMedVLConfig

BF16 = torch.bfloat16


class LayerWeights:
    __slots__ = ("qkv", "o", "gate_up", "down", "in_norm", "post_norm", "q_norm", "k_norm")


def _dev(t, device):
    return t.to(device=device, dtype=BF16).contiguous()


class _NoStore:
    """stand-in when the getter carries no packstore.PackStore: build everything"""
    @staticmethod
    def linear(key, build):
        return build()

    @staticmethod
    def tensor(name, build):
        return build()


def _store(get):
    return getattr(get, "pack_store", None) or _NoStore


def _tensor(get, device, name):
    """bf16 device tensor `name`: from the packed fast-path file when the getter has one (packstore.py), else from the checkpoint"""
    return _store(get).tensor(name, lambda: _dev(get(name), device))


def _linear(get, device, wname, bname=None, fp8=False):
    def build():
        w = _dev(get(wname), device)
        b = _dev(get(bname), device) if bname else None
        return ops.PackedLinear.from_weight_fp8(w, b) if fp8 else ops.PackedLinear.from_weight(w, b)
    return _store(get).linear(wname, build)


class LLMWeights:
    """Qwen2-MoT: `und[l]` and `gen[l]` LayerWeights, embeddings, final norms, lm_head."""

    def __init__(self, cfg: UniMedVLConfig, get, device, load_gen=True):
        p = "language_model.model."
        if cfg.llm_weight_dtype not in ("bf16", "fp8"):
            raise ValueError(f"llm_weight_dtype must be 'bf16' or 'fp8', got {cfg.llm_weight_dtype!r}")
        fp8 = self.fp8 = cfg.llm_weight_dtype == "fp8"
        if cfg.llm_act_dtype not in ("bf16", "fp8") or (cfg.llm_act_dtype == "fp8" and not fp8):
            raise ValueError("llm_act_dtype must be 'bf16', or 'fp8' together with llm_weight_dtype='fp8'")
        self.act8 = cfg.llm_act_dtype == "fp8"
        self.embed = _tensor(get, device, p + "embed_tokens.weight")
        self.und, self.gen = [], []
        for l in range(cfg.layers):
            self.und.append(self._layer(get, device, p + f"layers.{l}.", "", fp8))
            self.gen.append(self._layer(get, device, p + f"layers.{l}.", "_moe_gen", fp8) if load_gen else None)
            if self.act8:   # W8A8: the fp8-MFMA image replaces the bf16 image of the dequantised weights
                for lw in (self.und[-1], self.gen[-1]):
                    if lw is not None:
                        for lin in (lw.qkv, lw.o, lw.gate_up, lw.down):
                            lin.enable_fp8_mfma()
        self.norm = _tensor(get, device, p + "norm.weight")
        self.norm_gen = _tensor(get, device, p + "norm_moe_gen.weight") if load_gen else None
        self.lm_head = _linear(get, device, "language_model.lm_head.weight", fp8=fp8)
        # rotary tables exactly as Qwen2RotaryEmbedding returns them (modeling_qwen2.py:164-184):
        # fp32 outer product, cos/sin, cast to bf16; built on the CPU so the bits match torch's.
        hd = cfg.head_dim
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.int64).float() / hd))
        pos = torch.arange(cfg.max_position, dtype=torch.float32)
        freqs = (inv_freq[None, :, None].float() @ pos[None, None, :]).transpose(1, 2)[0]
        emb = torch.cat((freqs, freqs), dim=-1)
        self.cos = emb.cos().to(BF16).to(device)
        self.sin = emb.sin().to(BF16).to(device)

    @staticmethod
    def _layer(get, device, p, suf, fp8=False):
        lw = LayerWeights()
        a = p + "self_attn."
        st = _store(get)

        def build_qkv():
            w = torch.cat([_dev(get(a + f"{n}_proj{suf}.weight"), device) for n in "qkv"], 0)
            b = torch.cat([_dev(get(a + f"{n}_proj{suf}.bias"), device) for n in "qkv"], 0)
            return ops.PackedLinear.from_weight_fp8(w, b) if fp8 else ops.PackedLinear.from_weight(w, b)

        def build_gate_up():
            g = _dev(get(p + f"mlp{suf}.gate_proj.weight"), device)
            u = _dev(get(p + f"mlp{suf}.up_proj.weight"), device)
            return ops.PackedLinear.from_gate_up_fp8(g, u) if fp8 else ops.PackedLinear.from_gate_up(g, u)
        lw.qkv = st.linear(a + f"qkv_proj{suf}", build_qkv)
        lw.o = _linear(get, device, a + f"o_proj{suf}.weight", fp8=fp8)
        lw.gate_up = st.linear(p + f"mlp{suf}.gate_up_proj", build_gate_up)
        lw.down = _linear(get, device, p + f"mlp{suf}.down_proj.weight", fp8=fp8)
        lw.in_norm = _tensor(get, device, p + f"input_layernorm{suf}.weight")
        lw.post_norm = _tensor(get, device, p + f"post_attention_layernorm{suf}.weight")
        lw.q_norm = _tensor(get, device, a + f"q_norm{suf}.weight")
        lw.k_norm = _tensor(get, device, a + f"k_norm{suf}.weight")
        return lw

    def decode_weight_bytes(self):
        """bytes one decode step streams: the e4m3 images when llm_weight_dtype == "fp8", else the bf16 ones"""
        nb = (lambda lin: lin.w8.numel()) if self.fp8 else (lambda lin: lin.nbytes())
        n = nb(self.lm_head)
        for lw in self.und:
            n += nb(lw.qkv) + nb(lw.o) + nb(lw.gate_up) + nb(lw.down)
        return n


class ViTLayer:
    __slots__ = ("qkv", "out", "fc1", "fc2", "ln1_w", "ln1_b", "ln2_w", "ln2_b")


class ViTWeights:
    def __init__(self, cfg: UniMedVLConfig, get, device):
        p = "vit_model.vision_model."
        self.k_in = 3 * cfg.patch ** 2
        self.k_pad = (self.k_in + 31) // 32 * 32
        self.patch = _linear(get, device, p + "embeddings.patch_embedding.weight", p + "embeddings.patch_embedding.bias")
        self.patch.K = self.k_pad   # x is zero padded to a 32 multiple; the packed image already is
        self.pos = _tensor(get, device, p + "embeddings.position_embedding.weight")
        self.layers = []
        for l in range(cfg.vit_layers):
            q = p + f"encoder.layers.{l}."
            lw = ViTLayer()
            def build_qkv(q=q):
                w = torch.cat([_dev(get(q + f"self_attn.{n}_proj.weight"), device) for n in "qkv"], 0)
                b = torch.cat([_dev(get(q + f"self_attn.{n}_proj.bias"), device) for n in "qkv"], 0)
                return ops.PackedLinear.from_weight(w, b)
            lw.qkv = _store(get).linear(q + "self_attn.qkv_proj", build_qkv)
            lw.out = _linear(get, device, q + "self_attn.out_proj.weight", q + "self_attn.out_proj.bias")
            lw.fc1 = _linear(get, device, q + "mlp.fc1.weight", q + "mlp.fc1.bias")
            lw.fc2 = _linear(get, device, q + "mlp.fc2.weight", q + "mlp.fc2.bias")
            lw.ln1_w, lw.ln1_b = _tensor(get, device, q + "layer_norm1.weight"), _tensor(get, device, q + "layer_norm1.bias")
            lw.ln2_w, lw.ln2_b = _tensor(get, device, q + "layer_norm2.weight"), _tensor(get, device, q + "layer_norm2.bias")
            self.layers.append(lw)
        self.post_w = _tensor(get, device, p + "post_layernorm.weight")
        self.post_b = _tensor(get, device, p + "post_layernorm.bias")


class GlueWeights:
    """connector, vit/latent position tables, time embedder, vae2llm, llm2vae (bagel.py:114-143)."""

    def __init__(self, cfg: UniMedVLConfig, get, device, visual_gen=True, visual_und=True):
        if visual_und:
            self.conn1 = _linear(get, device, "connector.fc1.weight", "connector.fc1.bias")
            self.conn2 = _linear(get, device, "connector.fc2.weight", "connector.fc2.bias")
            self.vit_pos = _tensor(get, device, "vit_pos_embed.pos_embed")
        if visual_gen:
            self.latent_pos = _tensor(get, device, "latent_pos_embed.pos_embed")
            self.time0 = _linear(get, device, "time_embedder.mlp.0.weight", "time_embedder.mlp.0.bias")
            self.time2 = _linear(get, device, "time_embedder.mlp.2.weight", "time_embedder.mlp.2.bias")
            self.vae2llm = _linear(get, device, "vae2llm.weight", "vae2llm.bias")
            self.llm2vae = _linear(get, device, "llm2vae.weight", "llm2vae.bias")


def sincos_2d_table(embed_dim, grid_size):
    """Frozen PositionEmbedding table (modeling_utils.py:23-65,126-143); used when a
    checkpoint omits the buffer and for random-weight benches."""
    import numpy as np

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


def random_getter(cfg: UniMedVLConfig, device, seed=1234, std=0.02):
    """get(name) for synthetic benches: N(0, std^2) bf16 generated ON the device, tensor by
    tensor (the reference init rule, modeling_qwen2.py:597-606); norm gains 1, biases small.
    llm2vae is random, not the reference's zero init (bagel.py:156-159), so velocities are non-zero."""
    from . import shapes
    table = shapes.all_shapes(cfg)
    gen = torch.Generator(device=device).manual_seed(seed)

    def get(name):
        if name == "vit_pos_embed.pos_embed":
            return sincos_2d_table(cfg.hidden, cfg.vit_side).to(BF16)
        if name == "latent_pos_embed.pos_embed":
            return sincos_2d_table(cfg.hidden, cfg.max_latent).to(BF16)
        shp = table[name]
        if len(shp) == 1:
            if name.endswith("bias"):
                return (torch.randn(shp, device=device, generator=gen) * std).to(BF16)
            return torch.ones(shp, device=device, dtype=BF16)
        return (torch.randn(shp, device=device, generator=gen) * std).to(BF16)
    return get
