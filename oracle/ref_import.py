"""Import the reference implementation (/root/reference/codes) in THIS container.

TEST INFRASTRUCTURE ONLY.  Never imported by the product package.  The
reference is Python, so it cannot travel to the GPU box; it is used here to
(i) validate the CPU restatement in ``oracle/unimedvl_cpu.py`` and (ii)
generate the golden vectors committed under ``tests/golden/`` (see
``oracle/gen_golden.py``).

Shims:
  1. ``flash_attn`` is not installed: a stub module exporting
     ``flash_attn_varlen_func`` implemented per segment with
     ``F.scaled_dot_product_attention`` (GQA by repeat_interleave, causal =
     bottom-right aligned, the flash-attn >= 2.1 semantics the reference
     relies on, modeling_qwen2.py:369-372).
  2. transformers-5 drift: ``ROPE_INIT_FUNCTIONS['default']`` was removed; we
     register the published default rule inv_freq = 1/theta^(arange(0,d,2)/d).
  3. ``torchvision`` / ``cv2`` are absent; ``install_vision_stubs`` provides the four
     primitives the reference's data/transforms.py:15-115 calls (F.resize, ToTensor,
     Normalize, InterpolationMode) so that the reference's OWN size arithmetic
     (MaxLongEdgeMinShortEdgeResize.forward) and ImageTransform can be imported
     and run to make fixtures (import_reference_transforms).
"""
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn.functional as F

REF_ROOT = os.environ.get("UNIMEDVL_REFERENCE", "/root/reference/codes")


def _flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k,
                            max_seqlen_q=None, max_seqlen_k=None, causal=False,
                            **_unused):
    """q [Tq,Hq,D], k/v [Tk,Hk,D]; varlen segments; returns [Tq,Hq,D]."""
    out = torch.empty_like(q)
    nseg = cu_seqlens_q.numel() - 1
    rep = q.shape[1] // k.shape[1]
    for s in range(nseg):
        q0, q1 = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        k0, k1 = int(cu_seqlens_k[s]), int(cu_seqlens_k[s + 1])
        qs = q[q0:q1].transpose(0, 1).unsqueeze(0)  # [1,H,Lq,D]
        ks = k[k0:k1].transpose(0, 1).unsqueeze(0)
        vs = v[k0:k1].transpose(0, 1).unsqueeze(0)
        if rep > 1:
            ks = ks.repeat_interleave(rep, dim=1)
            vs = vs.repeat_interleave(rep, dim=1)
        lq, lk = q1 - q0, k1 - k0
        mask = None
        if causal:
            mask = torch.ones(lq, lk, dtype=torch.bool).tril(diagonal=lk - lq)
        o = F.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask)
        out[q0:q1] = o[0].transpose(0, 1)
    return out


def install_shims():
    if "flash_attn" not in sys.modules:
        m = types.ModuleType("flash_attn")
        m.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)
        m.__version__ = "2.5.8"
        m.flash_attn_varlen_func = _flash_attn_varlen_func
        m.flash_attn_func = None
        sys.modules["flash_attn"] = m
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    if "default" not in ROPE_INIT_FUNCTIONS:
        def _default_rope(config, device=None, seq_len=None, **kw):
            base = config.rope_theta
            dim = getattr(config, "head_dim", None) or config.hidden_size // config.num_attention_heads
            inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float().to(device) / dim))
            return inv, 1.0
        ROPE_INIT_FUNCTIONS["default"] = _default_rope


def install_vision_stubs():
    """cv2 / torchvision stand-ins, just enough for `import data.transforms` (SURVEY.md section 8c shim 3).

    torchvision semantics restated (torchvision 0.20, the version paired with the reference's torch 2.5.1):
      * F.resize(PIL, (h, w), BICUBIC, antialias) == PIL's Image.resize((w, h), Image.BICUBIC) (F_pil.resize; PIL always
        antialiases when down-scaling);
      * F.resize(tensor, (h, w), BICUBIC, antialias=True) == torch interpolate(mode="bicubic", align_corners=False,
        antialias=True) on a float copy, rounded and clamped back for uint8 inputs (F_t.resize);
      * ToTensor: HWC uint8 PIL -> CHW float32 / 255;  Normalize(mean, std, inplace=True): (x - mean) / std.
    """
    import numpy as np
    from PIL import Image
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.__spec__ = importlib.machinery.ModuleSpec("cv2", None)
        sys.modules["cv2"] = cv2
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")

    class InterpolationMode:
        NEAREST, BILINEAR, BICUBIC = "nearest", "bilinear", "bicubic"

    def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
        h, w = size
        if isinstance(img, torch.Tensor):
            if interpolation != InterpolationMode.BICUBIC:
                raise NotImplementedError(interpolation)
            x = img if img.dim() == 4 else img.unsqueeze(0)
            out = F.interpolate(x.to(torch.float32), size=(h, w), mode="bicubic", align_corners=False, antialias=bool(antialias))
            if img.dtype == torch.uint8:
                out = out.round().clamp(0, 255).to(torch.uint8)
            else:
                out = out.to(img.dtype)
            return out if img.dim() == 4 else out[0]
        pil_mode = {InterpolationMode.NEAREST: Image.NEAREST, InterpolationMode.BILINEAR: Image.BILINEAR,
                    InterpolationMode.BICUBIC: Image.BICUBIC}[interpolation]
        return img.resize((w, h), pil_mode)

    class ToTensor:
        def __call__(self, pic):
            if isinstance(pic, torch.Tensor):
                raise TypeError("pic should be PIL Image or ndarray")
            arr = np.asarray(pic)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            t = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous()
            return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t

    class Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std, self.inplace = mean, std, inplace

        def __call__(self, t):
            if not self.inplace:
                t = t.clone()
            mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return t.sub_(mean).div_(std)

    fn.resize = resize
    fn.InterpolationMode = InterpolationMode
    tr.functional, tr.InterpolationMode, tr.ToTensor, tr.Normalize = fn, InterpolationMode, ToTensor, Normalize
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn})


def import_reference_transforms():
    """The reference's data/transforms.py (ImageTransform, MaxLongEdgeMinShortEdgeResize) over the stubs above."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    sys.dont_write_bytecode = True
    install_vision_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import data.transforms as ref_tf
    return ref_tf


def import_reference():
    """Returns a namespace with the reference classes used on the hot path."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    sys.dont_write_bytecode = True
    install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    from modeling.unimedvl.bagel import Bagel, BagelConfig
    from modeling.unimedvl.qwen2_navit import Qwen2Config, Qwen2ForCausalLM, NaiveCache
    from modeling.unimedvl.siglip_navit import SiglipVisionConfig, SiglipVisionModel
    from modeling.autoencoder import AutoEncoder, AutoEncoderParams
    ns.Bagel, ns.BagelConfig = Bagel, BagelConfig
    ns.Qwen2Config, ns.Qwen2ForCausalLM, ns.NaiveCache = Qwen2Config, Qwen2ForCausalLM, NaiveCache
    ns.SiglipVisionConfig, ns.SiglipVisionModel = SiglipVisionConfig, SiglipVisionModel
    ns.AutoEncoder, ns.AutoEncoderParams = AutoEncoder, AutoEncoderParams
    return ns


def build_tiny_reference(cfg, seed=0):
    """Build a random-weight reference Bagel (+VAE) from a dict of tiny dims.

    cfg keys: hidden, layers, heads, kv_heads, inter, vocab, vit_hidden,
    vit_layers, vit_heads, vit_inter, patch, vit_side, max_latent, vae_ch,
    vae_mult.
    """
    ns = import_reference()
    torch.manual_seed(seed)
    llm_cfg = ns.Qwen2Config(
        vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
        num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
        num_key_value_heads=cfg["kv_heads"], max_position_embeddings=4096,
        rms_norm_eps=1e-6, rope_theta=1e6, qk_norm=True, tie_word_embeddings=False,
        layer_module="Qwen2MoTDecoderLayer", pad_token_id=None,
    )
    vit_cfg = ns.SiglipVisionConfig(
        hidden_size=cfg["vit_hidden"], intermediate_size=cfg["vit_inter"],
        num_hidden_layers=cfg["vit_layers"], num_attention_heads=cfg["vit_heads"],
        num_channels=3, image_size=cfg["patch"] * cfg["vit_side"], patch_size=cfg["patch"],
        hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, rope=False,
    )
    vae_params = ns.AutoEncoderParams(
        resolution=256, in_channels=3, downsample=2 ** (len(cfg["vae_mult"]) - 1), ch=cfg["vae_ch"],
        out_ch=3, ch_mult=list(cfg["vae_mult"]), num_res_blocks=cfg.get("vae_res", 2), z_channels=16,
        scale_factor=0.3611, shift_factor=0.1159,
    )
    vae = ns.AutoEncoder(vae_params)
    lm = ns.Qwen2ForCausalLM(llm_cfg)
    vit = ns.SiglipVisionModel(vit_cfg)
    bcfg = ns.BagelConfig(
        visual_gen=True, visual_und=True, llm_config=llm_cfg, vit_config=vit_cfg,
        vae_config=vae_params, vit_max_num_patch_per_side=cfg["vit_side"],
        connector_act="gelu_pytorch_tanh", latent_patch_size=2, max_latent_size=cfg["max_latent"],
    )
    model = ns.Bagel(lm, vit, bcfg)
    model.vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_cfg)
    # reference zero-inits llm2vae (bagel.py:156-159); give it signal
    torch.nn.init.normal_(model.llm2vae.weight, std=0.05)
    torch.nn.init.normal_(model.llm2vae.bias, std=0.05)
    # make norms / biases non-trivial so parity tests exercise them
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in list(model.named_parameters()) + list(vae.named_parameters()):
        if p.ndim == 1 and ("norm" in n or "layernorm" in n) and n.endswith("weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif p.ndim == 1 and n.endswith("bias"):
            p.data = 0.05 * torch.randn(p.shape, generator=g)
    model = model.to(torch.bfloat16).eval()
    vae = vae.to(torch.bfloat16).eval()
    return ns, model, vae
