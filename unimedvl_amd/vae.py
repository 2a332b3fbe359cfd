"""FLUX-style conv autoencoder on the HIP kernels (NHWC bf16 activations).

Mirrors ``AutoEncoder.encode / decode`` and the module tree of
codes/modeling/autoencoder.py:38-322 (ResnetBlock, AttnBlock, Down/Upsample, Encoder,
Decoder, DiagonalGaussian).  3x3 / 1x1 convolutions are MFMA implicit GEMMs over weights
re-tiled at load time, GroupNorm(32)+swish is one fused kernel pair, the nearest-2x
upsample and the asymmetric-pad stride-2 downsample live in the convolution's gather, the
mid-block attention reuses the varlen attention kernel (single head of C channels).
``encode`` samples with an injected or CPU-drawn noise tensor: the reference draws
``torch.randn_like`` on whatever device it runs on (autoencoder.py:270); a GPU RNG stream
cannot match it, so the noise is a host-side input here.
"""
import ctypes as C
import os

import torch

from . import _lib, ops
from .config import UniMedVLConfig

BF16 = torch.bfloat16


_stream = ops._stream


class _Conv:
    """conv weight [Cout,Cin,k,k] -> packed [Cout, k*k*Cin_p] (+bias), Cin padded to 8."""

    def __init__(self, w, b, device):
        w = w.to(device=device, dtype=BF16)
        cout, cin, k, _ = w.shape
        cin_p = (cin + 7) // 8 * 8
        if cin_p != cin:
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cin_p - cin))
        w2 = w.permute(0, 2, 3, 1).reshape(cout, k * k * cin_p).contiguous()
        self.lin = ops.PackedLinear.from_weight(w2, None)
        self.bias = None if b is None else b.to(device=device, dtype=BF16).contiguous()
        self.cout, self.cin, self.k = cout, cin_p, k


class AutoEncoder:
    def __init__(self, cfg: UniMedVLConfig, get, device="cuda"):
        """`get(name)` returns tensors keyed as the reference VAE state dict (ae.safetensors)."""
        self.cfg = cfg
        self.device = torch.device(device)
        self.scale_factor, self.shift_factor = cfg.scale_factor, cfg.shift_factor
        self.z = cfg.z_channels
        self._get = get
        self._convs, self._norms = {}, {}
        self.nlev, self.nres = len(cfg.vae_mult), cfg.vae_res
        lib = _lib.load()
        self._lib = lib
        self._ws = None
        self.attn_as_gemm = os.environ.get("UMV_VAE_ATTN_GEMM", "1") not in ("0", "")   # A/B only: 0 = the streaming attention kernel

    # ------------------------------------------------------------------ lazy parameter access
    def _conv_w(self, name):
        c = self._convs.get(name)
        if c is None:
            c = _Conv(self._get(name + ".weight"), self._get(name + ".bias"), self.device)
            self._convs[name] = c
        return c

    def _norm_w(self, name):
        n = self._norms.get(name)
        if n is None:
            n = (self._get(name + ".weight").to(device=self.device, dtype=BF16).contiguous(),
                 self._get(name + ".bias").to(device=self.device, dtype=BF16).contiguous())
            self._norms[name] = n
        return n

    def parameters(self):
        return iter(())

    # ------------------------------------------------------------------ ops on NHWC tensors [B,H,W,C]
    def conv(self, x, name, mode=0, residual=None):
        c = self._conv_w(name)
        B, H, W, Cin = x.shape
        assert Cin == c.cin, (name, Cin, c.cin)
        if mode == 0:
            Ho, Wo = H, W
        elif mode == 1:
            Ho, Wo = 2 * H, 2 * W
        else:
            Ho, Wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
        out = torch.empty((B, Ho, Wo, c.cout), dtype=BF16, device=x.device)
        _lib.check(self._lib.umv_conv2d_nhwc_bf16(
            x.data_ptr(), c.lin.wp.data_ptr(), None if c.bias is None else c.bias.data_ptr(),
            None if residual is None else residual.data_ptr(), out.data_ptr(), B, Cin, H, W, c.cout, c.k, mode,
            _stream()), "umv_conv2d_nhwc_bf16")
        return out

    def groupnorm(self, x, name, swish):
        g, b = self._norm_w(name)
        B, H, W, Cc = x.shape
        need = self._lib.umv_groupnorm_workspace_bytes(B, H * W)
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty(max(need // 4, 64), dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        _lib.check(self._lib.umv_groupnorm_nhwc_bf16(x.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                     self._ws.data_ptr(), B, H * W, Cc, 1e-6, int(swish), _stream()),
                   "umv_groupnorm_nhwc_bf16")
        return out

    def resblock(self, x, p):
        """ResnetBlock.forward (autoencoder.py:82-95)."""
        h = self.groupnorm(x, p + "norm1", True)
        h = self.conv(h, p + "conv1")
        h = self.groupnorm(h, p + "norm2", True)
        c1 = self._conv_w(p + "conv1")
        skip = x
        if c1.cin != c1.cout:
            skip = self.conv(x, p + "nin_shortcut")
        return self.conv(h, p + "conv2", residual=skip)        # x + h in the conv epilogue

    def attnblock(self, x, p):
        """AttnBlock.forward (autoencoder.py:50-65): single-head SDPA over H*W tokens of C channels."""
        B, H, W, Cc = x.shape
        h = self.groupnorm(x, p + "norm", False)
        key = p + "qkv"
        lin = self._convs.get(key)
        if lin is None:
            ws = [self._get(p + n + ".weight").to(device=self.device, dtype=BF16).reshape(Cc, Cc) for n in "qkv"]
            bs = [self._get(p + n + ".bias").to(device=self.device, dtype=BF16) for n in "qkv"]
            lin = ops.PackedLinear.from_weight(torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous())
            self._convs[key] = lin
        T = B * H * W
        qkv = ops.gemm(h.view(T, Cc), lin)
        n = H * W
        if n % 8 == 0 and 256 <= n <= 16384 and self.attn_as_gemm:
            # one head of Cc channels: S = Q K^T and O = P V are plain GEMMs at the tiled kernel's rate (per sample: K and V^T of
            # a sample are that GEMM's weight), fp32 scores, softmax rows and the final division in two small kernels:
            # 750 -> ~100 us at 448 x 448 against the streaming attention kernel at head dim 512.  Query rows go in chunks
            # that keep the fp32 score block under 256 MB (one chunk up to 8192 positions, 4 at a 1024 x 1024 image's 16384).
            lib = _lib.load()
            rc = n if n * n <= (64 << 20) else max(256, ((64 << 20) // n) // 256 * 256)
            o = torch.empty((T, Cc), dtype=BF16, device=x.device)
            S = torch.empty((rc, n), dtype=torch.float32, device=x.device)
            P = torch.empty((rc, n), dtype=BF16, device=x.device)
            l = torch.empty((rc,), dtype=torch.float32, device=x.device)
            Of = torch.empty((rc, Cc), dtype=torch.float32, device=x.device)
            for b in range(B):
                lin_k = ops.PackedLinear.from_weight(qkv[b * n:(b + 1) * n, Cc:2 * Cc].contiguous())        # [N = n keys, K = Cc]
                lin_v = ops.PackedLinear.from_weight(qkv[b * n:(b + 1) * n, 2 * Cc:].t().contiguous())      # V^T: [N = Cc, K = n keys]
                for r0 in range(0, n, rc):
                    m = min(rc, n - r0)
                    rows = slice(b * n + r0, b * n + r0 + m)
                    ops.gemm(qkv[rows, :Cc], lin_k, out=S, out_f32=True)
                    _lib.check(lib.umv_softmax_rows_f32(S.data_ptr(), n, P.data_ptr(), n, l.data_ptr(), m, n, float(Cc) ** -0.5, _stream()),
                               "umv_softmax_rows_f32")
                    ops.gemm(P[:m], lin_v, out=Of, out_f32=True)
                    _lib.check(lib.umv_rowscale_f32_bf16(Of.data_ptr(), Cc, l.data_ptr(), o[rows].data_ptr(), Cc, m, Cc, _stream()),
                               "umv_rowscale_f32_bf16")
            return self.conv(o.view(B, H, W, Cc), p + "proj_out", residual=x)
        slab = ops.KVSlab(B, 1, (n + 31) // 32 * 32, Cc, x.device)
        seg = torch.arange(B, dtype=torch.int32, device=x.device).repeat_interleave(n)
        slot = torch.arange(n, dtype=torch.int32, device=x.device).repeat(B)
        q = torch.empty((T, 1, Cc), dtype=BF16, device=x.device)
        ops.qkv_post(qkv, q, slab, seg, slot, None, 1, 1, Cc)
        cu = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=x.device)
        kvl = torch.full((B,), n, dtype=torch.int32, device=x.device)
        o = torch.empty((T, 1, Cc), dtype=BF16, device=x.device)
        ops.attention(q, o, slab, cu, kvl, 1, 1, Cc, False, n, n)
        return self.conv(o.view(B, H, W, Cc), p + "proj_out", residual=x)

    # ------------------------------------------------------------------ encoder / decoder
    @ops.on_device
    def encoder(self, x_nhwc):
        """Encoder.forward (autoencoder.py:169-187) -> moments NHWC [B,H/8,W/8,2z]."""
        h = self.conv(x_nhwc, "encoder.conv_in")
        for lvl in range(self.nlev):
            for b in range(self.nres):
                h = self.resblock(h, f"encoder.down.{lvl}.block.{b}.")
            if lvl != self.nlev - 1:
                h = self.conv(h, f"encoder.down.{lvl}.downsample.conv", mode=2)
        h = self.resblock(h, "encoder.mid.block_1.")
        h = self.attnblock(h, "encoder.mid.attn_1.")
        h = self.resblock(h, "encoder.mid.block_2.")
        h = self.groupnorm(h, "encoder.norm_out", True)
        return self.conv(h, "encoder.conv_out")

    @ops.on_device
    def decoder(self, z_nhwc):
        """Decoder.forward (autoencoder.py:240-257) -> NHWC [B,H,W,3]."""
        h = self.conv(z_nhwc, "decoder.conv_in")
        h = self.resblock(h, "decoder.mid.block_1.")
        h = self.attnblock(h, "decoder.mid.attn_1.")
        h = self.resblock(h, "decoder.mid.block_2.")
        for lvl in reversed(range(self.nlev)):
            for b in range(self.nres + 1):
                h = self.resblock(h, f"decoder.up.{lvl}.block.{b}.")
            if lvl != 0:
                h = self.conv(h, f"decoder.up.{lvl}.upsample.conv", mode=1)
        h = self.groupnorm(h, "decoder.norm_out", True)
        return self.conv(h, "decoder.conv_out")

    # ------------------------------------------------------------------ reference-shaped API
    def _to_nhwc(self, images):
        x = images.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, 8), dtype=BF16, device=self.device)
        _lib.check(self._lib.umv_nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), B, Cc, H, W, 8, _stream()),
                   "umv_nchw_f32_to_nhwc_bf16")
        return out

    @ops.on_device
    def encode_moments(self, images):
        return self.encoder(self._to_nhwc(images))

    def _noise(self, B, Hm, Wm, noise):
        if noise is None:   # same draw the reference makes on CPU: randn_like(mean) with mean bf16 [B,z,Hm,Wm]
            noise = torch.randn_like(torch.empty((B, self.z, Hm, Wm), dtype=BF16))
        return noise.to(device=self.device, dtype=BF16).contiguous()

    @ops.on_device
    def encode_packed(self, padded_images, latent_shapes, patch, noise=None):
        """vae.encode + the per-image crop / 2x2 patchify of bagel.py:757-776 -> [sum h*w, p*p*z] bf16."""
        mom = self.encode_moments(padded_images)
        B, Hm, Wm, _ = mom.shape
        nz = self._noise(B, Hm, Wm, noise)
        D = patch * patch * self.z
        total = sum(h * w for h, w in latent_shapes)
        out = torch.empty((total, D), dtype=BF16, device=self.device)
        off = 0
        for b, (h, w) in enumerate(latent_shapes):
            _lib.check(self._lib.umv_latent_sample_patchify(mom.data_ptr(), nz.data_ptr(), out[off:].data_ptr(), b, Hm, Wm,
                                                            self.z, h, w, patch, self.scale_factor, self.shift_factor,
                                                            _stream()), "umv_latent_sample_patchify")
            off += h * w
        return out

    @ops.on_device
    def encode(self, x, noise=None):
        """AutoEncoder.encode (autoencoder.py:300-303) -> [B,z,H/8,W/8] bf16 (NCHW, like the reference)."""
        mom = self.encode_moments(x)
        B, Hm, Wm, _ = mom.shape
        tok = self.encode_packed_from_moments(mom, noise)
        return tok

    @ops.on_device
    def encode_packed_from_moments(self, mom, noise):
        B, Hm, Wm, _ = mom.shape
        nz = self._noise(B, Hm, Wm, noise)
        outs = []
        for b in range(B):   # patch=1 "patchify" is the identity layout [Hm*Wm, z]
            t = torch.empty((Hm * Wm, self.z), dtype=BF16, device=self.device)
            _lib.check(self._lib.umv_latent_sample_patchify(mom.data_ptr(), nz.data_ptr(), t.data_ptr(), b, Hm, Wm, self.z,
                                                            Hm, Wm, 1, self.scale_factor, self.shift_factor, _stream()),
                       "umv_latent_sample_patchify")
            outs.append(t.view(Hm, Wm, self.z).permute(2, 0, 1))
        return torch.stack(outs, 0)

    @ops.on_device
    def decode_tokens(self, latent_tokens, image_shape, latent_downsample, patch):
        """latent tokens [h*w, p*p*z] -> decoder output NHWC bf16 [1,H,W,3]."""
        H, W = image_shape
        h, w = H // latent_downsample, W // latent_downsample
        tok = latent_tokens.to(device=self.device, dtype=torch.float32).contiguous()
        z = torch.empty((1, h * patch, w * patch, self.z), dtype=BF16, device=self.device)
        _lib.check(self._lib.umv_unpatchify_latent(tok.data_ptr(), z.data_ptr(), h, w, patch, self.z, self.scale_factor,
                                                   self.shift_factor, _stream()), "umv_unpatchify_latent")
        return self.decoder(z)

    @ops.on_device
    def decode_tokens_batch_to_uint8(self, latents, image_shape, latent_downsample, patch):
        """decode_tokens_to_uint8 for several latents of ONE image shape in one pass through the decoder: [B,H,W,3] uint8.
        Convolutions, GroupNorm and the mid-block attention are per sample, so every image equals its own
        decode_tokens_to_uint8 - bit for bit whenever the 1x1-convolution GEMMs take the same kernel either way (>= 65 rows
        per image, i.e. any real size; tests/test_vae_gpu.py).  The decoder's low-resolution levels are tiny grids for one
        image (a 32 x 32 x 512 convolution is 64 workgroups of a 72-step chain: 150 us each on 256 CUs); a batch fills them:
        four 256 x 256 images 27.3 -> 14.3 ms."""
        H, W = image_shape
        h, w = H // latent_downsample, W // latent_downsample
        B = len(latents)
        z = torch.empty((B, h * patch, w * patch, self.z), dtype=BF16, device=self.device)
        for b, lt in enumerate(latents):
            tok = lt.to(device=self.device, dtype=torch.float32).contiguous()
            _lib.check(self._lib.umv_unpatchify_latent(tok.data_ptr(), z[b].data_ptr(), h, w, patch, self.z, self.scale_factor,
                                                       self.shift_factor, _stream()), "umv_unpatchify_latent")
        img = self.decoder(z)
        _, Ho, Wo, Cs = img.shape
        out = torch.empty((B, Ho, Wo, 3), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.umv_pixels_to_u8(img.data_ptr(), out.data_ptr(), B * Ho * Wo, Cs, _stream()), "umv_pixels_to_u8")
        return out

    @ops.on_device
    def decode_tokens_to_uint8(self, latent_tokens, image_shape, latent_downsample, patch):
        img = self.decode_tokens(latent_tokens, image_shape, latent_downsample, patch)
        _, H, W, Cs = img.shape
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.umv_pixels_to_u8(img.data_ptr(), out.data_ptr(), H * W, Cs, _stream()), "umv_pixels_to_u8")
        return out

    @ops.on_device
    def decode(self, z):
        """AutoEncoder.decode (autoencoder.py:305-307): z [B,z,h,w] -> [B,3,H,W] bf16 (NCHW view)."""
        outs = []
        for b in range(z.shape[0]):
            zz = z[b].to(self.device)
            c, h, w = zz.shape
            tok = zz.permute(1, 2, 0).reshape(h * w, c).float().contiguous()   # patch=1 tokens
            img = self.decode_tokens(tok, (h, w), 1, 1)
            outs.append(img[0].permute(2, 0, 1))
        return torch.stack(outs, 0)
