"""VAE kernels and the image paths that use them, against reference goldens / the oracle.
Tolerances: VAE activations go through ~30 bf16 conv+GroupNorm stages; decoded images in
[-1,1]-ish units are compared at atol 0.06 (max) / 0.01 (mean); uint8 pixels within +-4 on
>= 99% of pixels (the reference's bf16 *255 quantises to 1-2 grey levels above 128)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def vae(tiny_weights):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.vae import AutoEncoder
    cfg, sd, vae_sd, _ = tiny_weights
    return AutoEncoder(UniMedVLConfig.from_dict(cfg), lambda n: vae_sd[n], device="cuda")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF16)


@pytest.mark.parametrize("mode,cin,cout,h,w", [(0, 16, 128, 8, 8), (0, 128, 64, 13, 9), (1, 64, 64, 6, 10),
                                               (2, 32, 32, 16, 12), (2, 32, 32, 15, 11), (0, 8, 32, 20, 20),
                                               # the input-stationary kernel (conv3x3_patch_kernel: Cin % 64 == 0, stride 1), forced
                                               # by UMV_CONV_PATCH=2 below: ragged tiles, several 64-channel slices, Cout < / > 128
                                               (0, 64, 128, 16, 16), (0, 128, 64, 13, 9), (0, 192, 160, 37, 21), (0, 256, 256, 32, 48),
                                               (1, 64, 64, 8, 8), (1, 128, 128, 9, 13), (1, 256, 128, 24, 24)])
def test_conv3x3(mode, cin, cout, h, w):
    from unimedvl_amd import _lib
    from unimedvl_amd.vae import _Conv, _stream
    lib = _lib.load()
    x = rnd((2, cin, h, w), 1)
    wt, b = rnd((cout, cin, 3, 3), 2, 1 / (3 * cin ** 0.5)), rnd((cout,), 3, 0.1)
    res = None
    if mode == 0:
        ref = F.conv2d(x.float(), wt.float(), b.float(), padding=1)
    elif mode == 1:
        ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), wt.float(), b.float(), padding=1)
    else:
        ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), wt.float(), b.float(), stride=2)
    ref = ref.to(BF16)
    if mode == 0:
        res = rnd(tuple(ref.shape), 4)
        ref = ref + res
    c = _Conv(wt, b, "cuda")
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    Ho, Wo = ref.shape[2], ref.shape[3]
    resn = None if res is None else res.permute(0, 2, 3, 1).contiguous().cuda()
    # mode | 32 = the gather kernel (conv_tiled_kernel), | 16 = the input-stationary kernel (conv3x3_patch_kernel), plain = the policy
    variants = [mode, mode | 32] + ([mode | 16] if (mode in (0, 1) and cin % 64 == 0) else [])
    for mv in variants:
        out = torch.full((2, Ho, Wo, cout), float("nan"), dtype=BF16, device="cuda")
        _lib.check(lib.umv_conv2d_nhwc_bf16(xn.data_ptr(), c.lin.wp.data_ptr(), c.bias.data_ptr(),
                                            None if resn is None else resn.data_ptr(), out.data_ptr(), 2, cin, h, w, cout, 3,
                                            mv, _stream()), "conv")
        got = out.cpu().permute(0, 3, 1, 2).float()
        err = (got - ref.float()).abs().max().item()
        assert err <= 2 ** -6 * max(1.0, ref.float().abs().max().item()), f"conv mode {mv}: max err {err}"


@pytest.mark.parametrize("C,hw,swish", [(32, 64, True), (128, 300, True), (512, 1024, False), (64, 257, True)])
def test_groupnorm(vae, C, hw, swish):
    from unimedvl_amd import _lib
    from unimedvl_amd.vae import _stream
    lib = _lib.load()
    x = rnd((2, C, hw, 1), 5, 2.0)
    g, b = rnd((C,), 6) + 1, rnd((C,), 7)
    ref = F.group_norm(x, 32, g, b, 1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty_like(xn)
    ws = torch.empty(lib.umv_groupnorm_workspace_bytes(2, hw) // 4 + 16, dtype=torch.float32, device="cuda")
    gd, bd = g.cuda(), b.cuda()   # keep the device copies alive across the asynchronous launch
    _lib.check(lib.umv_groupnorm_nhwc_bf16(xn.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr(),
                                           ws.data_ptr(), 2, hw, C, 1e-6, int(swish), _stream()), "gn")
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2).float()
    d = (got - ref.float()).abs()
    tol = 2 ** -6 * ref.float().abs().clamp_min(1.0)     # 2 bf16 ulp of the value
    assert bool((d <= tol).all()) and d.mean().item() < 3e-3, f"groupnorm C={C}: max {d.max().item()} mean {d.mean().item()}"


def test_groupnorm_swish_every_bf16_value():
    """the fused swish of umv_groupnorm_nhwc_bf16 uses v_exp_f32 / v_rcp_f32 (about 1 ulp each) where the reference
    computes torch.sigmoid on a bf16 tensor - check ALL bf16 values (round 4 found and fixed the one range that differed: y <= -87.5,
    where the clamped hardware exponent returned y * 2^-126 instead of the underflowing sigmoid's -0).  With gamma = 0 and beta = v the normalised value is exactly
    v for every pixel, so 32 launches of 2048 channels push every finite bf16 pattern through y -> bf16(y * bf16(sigmoid(y)))."""
    from unimedvl_amd import _lib
    from unimedvl_amd.vae import _stream
    lib = _lib.load()
    C, hw = 2048, 4
    bits = torch.arange(65536, dtype=torch.int32)
    allv = bits.to(torch.int16).view(BF16)
    x = torch.tensor([1.0, -1.0, 1.0, -1.0]).view(1, hw, 1).expand(1, hw, C).contiguous().to(BF16).cuda()
    ws = torch.empty(lib.umv_groupnorm_workspace_bytes(1, hw) // 4 + 16, dtype=torch.float32, device="cuda")
    gamma = torch.zeros(C, dtype=BF16, device="cuda")
    out = torch.empty_like(x)
    bad, worst, checked = 0, 0.0, 0
    for i in range(65536 // C):
        v = allv[i * C:(i + 1) * C]
        beta = v.cuda()
        _lib.check(lib.umv_groupnorm_nhwc_bf16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), 1, hw, C, 1e-6, 1,
                                               _stream()), "gn")
        torch.cuda.synchronize()
        got = out[0, 0].cpu()
        fin = torch.isfinite(v.float())
        ref = v * torch.sigmoid(v)                         # bf16 tensor ops: sigmoid rounded to bf16, product rounded to bf16
        ne = (got.view(torch.int16) != ref.view(torch.int16)) & fin & ~((got.float() == 0) & (ref.float() == 0))
        bad += int(ne.sum())
        checked += int(fin.sum())
        if ne.any():
            worst = max(worst, float(((got.float() - ref.float()).abs() / ref.float().abs().clamp_min(1e-30))[ne].max()))
    print(f"swish over all {checked} finite bf16 values: {bad} differ from torch (worst relative difference {worst:.4g})")
    assert bad == 0, f"{bad} of {checked} bf16 values differ from torch.sigmoid's swish (worst relative difference {worst:.4g})"


def test_attention_hd512():
    from unimedvl_amd import ops
    from oracle.unimedvl_cpu import attention_segment
    n, hd = 200, 512
    q, k, v = rnd((n, 1, hd), 8, 0.3), rnd((n, 1, hd), 9, 0.3), rnd((n, 1, hd), 10)
    slab = ops.KVSlab(1, 1, 224, hd, "cuda")
    kh, vh = slab.k.cpu(), slab.vt.cpu()
    kh[0, :, :n] = k.transpose(0, 1)
    vh[0, :, :, :n] = v.permute(1, 2, 0)
    slab.k.copy_(kh); slab.vt.copy_(vh)
    out = torch.zeros((n, 1, hd), dtype=BF16, device="cuda")
    ops.attention(q.cuda(), out, slab, torch.tensor([0, n], dtype=torch.int32).cuda(),
                  torch.tensor([n], dtype=torch.int32).cuda(), 1, 1, hd, False, n, n)
    ref = attention_segment(q, k, v, False, impl="flash")
    err = (out.cpu().float() - ref.float()).abs().max().item()
    assert err < 0.03, err


def test_vae_decode_encode_vs_reference(vae):
    g = load_golden("vae")
    dec = vae.decode(g["z"].to(BF16)).float().cpu()
    d = (dec - g["decoded"].float()).abs()
    assert d.max().item() < 0.06 and d.mean().item() < 0.01, f"decode: max {d.max().item()} mean {d.mean().item()}"
    enc = vae.encode(g["image"], noise=g["enc_noise"]).float().cpu()
    d = (enc - g["encoded"].float()).abs()
    assert d.max().item() < 0.06 and d.mean().item() < 0.01, f"encode: max {d.max().item()} mean {d.mean().item()}"


def test_t2i_pixels_vs_reference(vae, tiny_weights):
    g = load_golden("t2i")
    H, W = g["image_shape"].tolist()
    cfg = tiny_weights[0]
    down = 2 ** (len(cfg["vae_mult"]) - 1) * cfg["latent_patch"]
    px = vae.decode_tokens_to_uint8(g["latent_global"], (H, W), down, cfg["latent_patch"]).cpu()
    ref = g["pixels_u8"]
    assert px.shape == ref.shape
    diff = (px.int() - ref.int()).abs()
    assert (diff <= 4).float().mean().item() >= 0.99 and diff.max().item() <= 16, \
        f"pixels: {100 * (diff <= 4).float().mean().item():.2f}% within 4, max {diff.max().item()}"


def test_batched_decode_matches_single(vae, tiny_weights):
    """decode_tokens_batch_to_uint8 (the bench's T2I leg decodes its images of one shape in one pass) against
    decode_tokens_to_uint8 image by image: convolutions, GroupNorm(32) and the mid-block attention are per sample
    (autoencoder.py:240-257), so the only difference is which GEMM kernel the 1x1 convolutions / attention projections land on
    (B * H * W rows: at this toy size one image is <= 64 rows = the weight-streaming kernel, four are the tiled one - another
    fp32 summation order, amplified by ~30 bf16 stages and the truncation to uint8).  Here: every pixel within 6 grey levels (measured: max 5,
    94 % within one); at real sizes: bit for bit (next test)."""
    g = load_golden("t2i")
    H, W = g["image_shape"].tolist()
    cfg = tiny_weights[0]
    down = 2 ** (len(cfg["vae_mult"]) - 1) * cfg["latent_patch"]
    lat0 = g["latent_global"].float()
    gen = torch.Generator().manual_seed(11)
    lats = [lat0, lat0 + 0.3 * torch.randn(lat0.shape, generator=gen), -lat0, 0.5 * torch.randn(lat0.shape, generator=gen)]
    batch = vae.decode_tokens_batch_to_uint8(lats, (H, W), down, cfg["latent_patch"]).cpu()
    assert batch.shape[0] == 4 and batch.dtype == torch.uint8
    for b, lt in enumerate(lats):
        single = vae.decode_tokens_to_uint8(lt, (H, W), down, cfg["latent_patch"]).cpu()
        d = (batch[b].int() - single.int()).abs()
        assert d.max().item() <= 6 and (d <= 1).float().mean().item() >= 0.9, f"image {b}: max {d.max().item()}"
    assert not torch.equal(batch[0], batch[2])


def test_batched_decode_bit_identical_at_full_size():
    """the full-size decoder (128 / 256 / 512 channels, 16 latent channels) on four 256 x 256 images: one batched pass == four
    single passes, bit for bit (every GEMM-shaped op has >= 1024 rows either way: same kernels, same summation order)."""
    import math
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.shapes import vae_shapes
    from unimedvl_amd.vae import AutoEncoder
    cfg = UniMedVLConfig()
    dev = torch.device("cuda", 0)
    shapes = vae_shapes(cfg.to_dict())
    gen = torch.Generator(device=dev).manual_seed(4321)

    def vget(name):
        shp = shapes[name]
        if len(shp) == 1:
            return torch.ones(shp, device=dev, dtype=BF16) if name.endswith("weight") else torch.zeros(shp, device=dev, dtype=BF16)
        return (torch.randn(shp, device=dev, generator=gen) / math.sqrt(math.prod(shp[1:]))).to(BF16)
    full = AutoEncoder(cfg, vget, device=dev)
    hw, down, patch = 256, 16, 2
    lats = [torch.randn((hw // down) ** 2, patch * patch * full.z, device=dev, generator=gen) for _ in range(4)]
    batch = full.decode_tokens_batch_to_uint8(lats, (hw, hw), down, patch)
    assert batch.shape == (4, hw, hw, 3)
    for b, lt in enumerate(lats):
        assert torch.equal(batch[b], full.decode_tokens_to_uint8(lt, (hw, hw), down, patch)), f"image {b}"
    assert batch.float().std().item() > 1.0


def test_edit_prefill_vs_reference(vae, tiny_weights):
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    cfg, sd, _, _ = tiny_weights
    model = Bagel(UniMedVLConfig.from_dict(cfg), lambda n: sd[n], device="cuda")
    g = load_golden("edit_prefill")
    cache = NaiveCache(cfg["layers"])
    gi, kvl, rope = model.prepare_vae_images([0], [0], [g["image"]], lambda x: x, NEW_TOKEN_IDS)
    cache = model.forward_cache_update_vae(vae, cache, noise=g["enc_noise"], **gi)
    assert kvl == g["kv_lens"].tolist() and rope == g["ropes"].tolist()
    k0, vL = cache.packed_keys(0).float().cpu(), cache.packed_values(cfg["layers"] - 1).float().cpu()
    for got, ref, name in ((k0, g["k0"].float(), "k0"), (vL, g["vL"].float(), "vL")):
        err = (got - ref).abs().max().item()
        assert err <= 3e-2 * ref.abs().max().item(), f"{name}: max err {err} vs scale {ref.abs().max().item()}"


def test_attnblock_gemm_form_vs_streaming_kernel_and_fp32_model(tiny_weights):
    """AttnBlock.attention (autoencoder.py:50-62) at FULL width (one head of 512 channels, 28 x 28 = 784 and 56 x 56 = 3136
    positions, two samples): the two-GEMM form (umv_softmax_rows_f32 / umv_rowscale_f32_bf16 between the GEMMs) against an fp32
    model of the flash arithmetic and against the streaming attention kernel it replaces; per sample, so batched == single."""
    import math
    from unimedvl_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    C_ = 512
    for n in (784, 3136):
        B = 2
        qkv = (torch.randn(B * n, 3 * C_, generator=g) * 1.5).to(BF16).cuda()
        # fp32 model: scores from the bf16 operands, softmax in fp32, weights rounded to bf16 against the row maximum
        ref = torch.empty(B * n, C_, device="cuda")
        for b in range(B):
            q, k, v = (qkv[b * n:(b + 1) * n, i * C_:(i + 1) * C_].float() for i in range(3))
            s = q @ k.t() / math.sqrt(C_)
            p = torch.exp(s - s.max(-1, keepdim=True).values)
            ref[b * n:(b + 1) * n] = (p.to(BF16).float() @ v) / p.sum(-1, keepdim=True)
        # the two-GEMM form, exactly as vae.py::attnblock runs it
        o = torch.empty((B * n, C_), dtype=BF16, device="cuda")
        S = torch.empty((n, n), dtype=torch.float32, device="cuda")
        P = torch.empty((n, n), dtype=BF16, device="cuda")
        l = torch.empty((n,), dtype=torch.float32, device="cuda")
        Of = torch.empty((n, C_), dtype=torch.float32, device="cuda")
        for b in range(B):
            rows = slice(b * n, (b + 1) * n)
            ops.gemm(qkv[rows, :C_], ops.PackedLinear.from_weight(qkv[rows, C_:2 * C_].contiguous()), out=S, out_f32=True)
            _lib.check(lib.umv_softmax_rows_f32(S.data_ptr(), n, P.data_ptr(), n, l.data_ptr(), n, n, C_ ** -0.5, ops._stream()), "softmax")
            ops.gemm(P, ops.PackedLinear.from_weight(qkv[rows, 2 * C_:].t().contiguous()), out=Of, out_f32=True)
            _lib.check(lib.umv_rowscale_f32_bf16(Of.data_ptr(), C_, l.data_ptr(), o[rows].data_ptr(), C_, n, C_, ops._stream()), "rowscale")
        assert torch.isfinite(o.float()).all()
        d = (o.float() - ref).abs()
        assert d.max().item() <= 2e-3 + 2 ** -7 * ref.abs().max().item(), (n, d.max().item())
        assert (o == ref.to(BF16)).float().mean().item() > 0.85, "most elements must agree with the fp32 model to the last bf16 bit"
        # the streaming kernel (online softmax in 32-key blocks) agrees to a few bf16 ulps
        slab = ops.KVSlab(B, 1, (n + 31) // 32 * 32, C_, "cuda")
        seg = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(n)
        slot = torch.arange(n, dtype=torch.int32, device="cuda").repeat(B)
        qq = torch.empty((B * n, 1, C_), dtype=BF16, device="cuda")
        ops.qkv_post(qkv, qq, slab, seg, slot, None, 1, 1, C_)
        o2 = torch.empty((B * n, 1, C_), dtype=BF16, device="cuda")
        ops.attention(qq, o2, slab, torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device="cuda"),
                      torch.full((B,), n, dtype=torch.int32, device="cuda"), 1, 1, C_, False, n, n)
        assert (o.float() - o2.view(B * n, C_).float()).abs().max().item() <= 2e-3 + 2 ** -6 * ref.abs().max().item()


def test_softmax_rows_argument_errors():
    from unimedvl_amd import _lib, ops
    lib = _lib.load()
    S = torch.zeros((4, 10), dtype=torch.float32, device="cuda")
    P = torch.zeros((4, 10), dtype=BF16, device="cuda")
    l = torch.zeros(4, dtype=torch.float32, device="cuda")
    assert lib.umv_softmax_rows_f32(S.data_ptr(), 10, P.data_ptr(), 10, l.data_ptr(), 4, 9, 1.0, ops._stream()) != 0      # odd n
    assert lib.umv_softmax_rows_f32(S.data_ptr(), 10, P.data_ptr(), 10, l.data_ptr(), 4, 20000, 1.0, ops._stream()) != 0  # too wide
    assert lib.umv_softmax_rows_f32(S.data_ptr(), 10, P.data_ptr(), 10, l.data_ptr(), 4, 10, 1.0, ops._stream()) == 0
    torch.cuda.synchronize()
    assert torch.allclose(l.cpu(), torch.full((4,), 10.0)) and (P.float() == 1).all()
    # the wide variant (8192 < n <= 16384: a 1024 x 1024 image's latent) against torch
    g = torch.Generator().manual_seed(3)
    S2 = (torch.randn(8, 16384, generator=g) * 3).cuda()
    P2 = torch.zeros((8, 16384), dtype=BF16, device="cuda")
    l2 = torch.zeros(8, dtype=torch.float32, device="cuda")
    assert lib.umv_softmax_rows_f32(S2.data_ptr(), 16384, P2.data_ptr(), 16384, l2.data_ptr(), 8, 16384, 0.5, ops._stream()) == 0
    torch.cuda.synchronize()
    ref = torch.exp((S2 - S2.max(-1, keepdim=True).values) * 0.5)
    assert torch.allclose(l2, ref.sum(-1), rtol=1e-5) and (P2.float() - ref).abs().max().item() <= 2 ** -8
