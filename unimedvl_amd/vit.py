"""Packed variable-resolution SigLIP tower on the HIP kernels.

Mirrors ``SiglipVisionModel.forward(packed_pixel_values, packed_flattened_position_ids,
cu_seqlens, max_seqlen)`` (codes/modeling/unimedvl/siglip_navit.py:389-402, 345-371):
linear patch embed + learned absolute positions, pre-LN blocks with one full-attention
window per image (cu_seqlens), post-LN.  head_dim 72 runs on the same varlen attention
kernel as the LLM (zero-padded to 96 / 80 inside MFMA fragments).
"""
import torch

from . import ops
from .config import UniMedVLConfig
from .data_utils import PackedVitImages
from .weights import ViTWeights

BF16 = torch.bfloat16


class SiglipVisionModel:
    def __init__(self, cfg: UniMedVLConfig, weights: ViTWeights, device):
        self.cfg, self.w, self.device = cfg, weights, device

    def __call__(self, packed_pixel_values=None, packed_flattened_position_ids=None, cu_seqlens=None, max_seqlen=None, plan=None):
        return self.forward(packed_pixel_values, packed_flattened_position_ids, cu_seqlens, max_seqlen, plan=plan)

    @ops.on_device
    def make_plan(self, packed_pixel_values, packed_flattened_position_ids, cu_seqlens, max_seqlen, into=None):
        """Host -> device part of a forward (pixels, position ids, segment bookkeeping) as persistent tensors; `into`
        refreshes an existing plan in place for a same-shaped batch (the addresses a captured HIP graph reads)."""
        dev = self.device
        cu_host = [int(v) for v in cu_seqlens.tolist()]
        lens = [cu_host[i + 1] - cu_host[i] for i in range(len(cu_host) - 1)]
        # PackedVitImages (prepare_vit_images with device_patchify): upload the [3, H, W] images, the tokens are made by
        # umv_patchify_f32_bf16 inside forward(); a tensor: the reference's packed patch tokens
        images = packed_pixel_values.images if isinstance(packed_pixel_values, PackedVitImages) else None
        if images is not None and packed_pixel_values.token_counts() != lens:
            raise ValueError(f"images of {packed_pixel_values.token_counts()} patches against vit_token_seqlens {lens}")
        if into is not None:
            if into["lens"] != lens or (images is None) != (into["imgs"] is None):
                raise ValueError("a ViT plan can only be refreshed for the image sizes (and input kind) it was made for")
            if images is None:
                into["px"].copy_(packed_pixel_values.to(torch.float32), non_blocking=True)
            else:
                for dst, im in zip(into["imgs"], images):
                    if dst.shape != im.shape:
                        raise ValueError("a ViT plan can only be refreshed for the image sizes it was made for")
                    dst.copy_(im.to(torch.float32), non_blocking=True)
            into["pos_ids"].copy_(packed_flattened_position_ids.to(torch.int64), non_blocking=True)
            return into
        seg, slot = [], []
        for i, n in enumerate(lens):
            seg += [i] * n
            slot += list(range(n))
        return dict(px=None if images is not None else packed_pixel_values.to(device=dev, dtype=torch.float32).contiguous(),
                    imgs=None if images is None else [im.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous() for im in images],
                    pos_ids=packed_flattened_position_ids.to(device=dev, dtype=torch.int64),
                    cu_q=torch.tensor(cu_host, dtype=torch.int32).to(dev), kv_len=torch.tensor(lens, dtype=torch.int32).to(dev),
                    meta=torch.tensor([seg, slot], dtype=torch.int32).to(dev), lens=lens, max_seqlen=int(max_seqlen))

    @ops.on_device
    def forward(self, packed_pixel_values=None, packed_flattened_position_ids=None, cu_seqlens=None, max_seqlen=None, plan=None):
        cfg, w, dev = self.cfg, self.w, self.device
        nh, hd, h_dim = cfg.vit_heads, cfg.vit_head_dim, cfg.vit_hidden
        if plan is None:
            plan = self.make_plan(packed_pixel_values, packed_flattened_position_ids, cu_seqlens, max_seqlen)
        px, pos_ids, cu_q, kv_len, meta, lens, max_seqlen = (plan[k] for k in ("px", "pos_ids", "cu_q", "kv_len", "meta", "lens", "max_seqlen"))
        N = sum(lens)
        nimg = len(lens)

        if plan.get("imgs") is not None:                     # patchify (data_utils.py:43-50) + the cast below, on the device
            xb = torch.empty((N, w.k_pad), dtype=BF16, device=dev)
            off = 0
            for im, n in zip(plan["imgs"], lens):
                ops.patchify(im, xb[off:off + n], cfg.patch)
                off += n
        else:
            xb = ops.cast_pad(px, w.k_pad)                   # autocast's fp32->bf16 cast of the pixels
        h = ops.gemm(xb, w.patch)                            # patch embedding (siglip_navit.py:190)
        ops.add_rows(h, h, table=w.pos, idx=pos_ids)         # + position_embedding(ids) (:192)
        # no cache in this tower: q and K are read by the attention kernel where the QKV GEMM wrote them (column slices of
        # `qkv`); only V needs its transposed slab
        slab = ops.KVSlab(nimg, nh, (max(lens) + 31) // 32 * 32, hd, dev, keys=False)
        x = torch.empty_like(h)
        qkv = torch.empty((N, 3 * h_dim), dtype=BF16, device=dev)
        o = torch.empty((N, h_dim), dtype=BF16, device=dev)
        a = torch.empty((N, cfg.vit_inter), dtype=BF16, device=dev)
        for lw in w.layers:
            ops.layernorm(h, lw.ln1_w, lw.ln1_b, cfg.ln_eps, out=x)
            ops.gemm(x, lw.qkv, out=qkv)
            ops.qkv_post(qkv, None, slab, meta[0], meta[1], None, nh, nh, hd)
            ops.attention(qkv[:, :h_dim], o, slab, cu_q, kv_len, nh, nh, hd, False, max_seqlen, max_seqlen, k_packed=qkv[:, h_dim:2 * h_dim])
            ops.gemm(o, lw.out, out=h, residual=h)
            ops.layernorm(h, lw.ln2_w, lw.ln2_b, cfg.ln_eps, out=x)
            ops.gemm(x, lw.fc1, out=a, act="gelu_tanh")
            ops.gemm(a, lw.fc2, out=h, residual=h)
        return ops.layernorm(h, w.post_w, w.post_b, cfg.ln_eps)
