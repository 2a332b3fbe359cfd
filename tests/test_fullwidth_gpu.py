"""FULL-WIDTH, reduced-depth parity against the CPU oracle (VERDICT r01 "next" #1).

BASELINE.json configs[1] / [2] / [3] at the real 14B WIDTHS (hidden 3584, 28/4 heads of 128, inter 18944, vocab 152064, ViT
1152/16x72/4304, VAE 128 ch) with 2 LLM layers and 2 ViT layers, so that the CPU oracle (oracle/unimedvl_cpu.py, pinned
bit-exact to the reference on the tiny goldens) finishes in seconds while the HIP engine runs exactly the kernel variants
the headline bench runs: gemm_tiled_kernel<2,4,8,4,1,4,1> (cfg 266) on 8208-row prefills, attn_prefill_kernel<128,2> /
<72,2>, the weight-streaming kernels at N = 37 888 / 152 064, split-K partial sums + residual_rmsnorm, split-KV decode
attention at context 1060 under the HIP graph, the packed 3-context flow pass and the full-size VAE decoder.

Same seeded weights on both sides (generated once on the device, copied to the host for the oracle); same inputs.
Tolerances are SURVEY.md section 8c's: K / V within 2e-2 of the tensor's range, logits atol 0.25 and cosine > 0.999, greedy
ids exact wherever the oracle's top-2 margin exceeds 0.25.  Decode is compared TEACHER-FORCED: the engine decodes freely
(greedy, HIP graph), the oracle is fed the engine's input token of every step, so one flipped near-tie does not end the
comparison.  Reference anchors: bagel.py:523-615 (image prefill), :412-458 (text), :1236-1317 (decode), :901-1211 (flow),
inferencer.py:234-256 (pixels)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


class IdTok:
    def __init__(self, table):
        self.table = table

    def encode(self, s):
        return self.table[int(s)]


def _synth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 1, h // 32 + 2, w // 32 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)[0]
    img = (img / img.abs().max()).clamp(-1, 1)
    return (img.repeat(3, 1, 1) + 0.05 * torch.randn(3, h, w, generator=g)).clamp(-1, 1).contiguous()


@pytest.fixture(scope="module")
def fw():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from oracle.unimedvl_cpu import OracleBagel
    from unimedvl_amd import shapes
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.vae import AutoEncoder
    from unimedvl_amd.weights import random_getter
    cfg = UniMedVLConfig(layers=2, vit_layers=2)        # every other dimension at its 14B value
    dev = torch.device("cuda", 0)
    get = random_getter(cfg, dev, seed=4242)
    sd = {name: get(name) for name in shapes.all_shapes(cfg)}           # one pass, fixed order
    # norm gains away from 1 and non-trivial q/k norms so that a swapped or skipped gain cannot hide
    g = torch.Generator(device=dev).manual_seed(17)
    for k, v in sd.items():
        if v.dim() == 1 and "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * torch.randn(v.shape, device=dev, generator=g)).to(BF16)
    vshapes = shapes.vae_shapes(cfg.to_dict())
    vae_sd = {}
    for name, shp in vshapes.items():
        if len(shp) == 1:
            vae_sd[name] = ((1.0 if name.endswith("weight") else 0.0) + 0.05 * torch.randn(shp, device=dev, generator=g)).to(BF16)
        else:
            fan_in = math.prod(shp[1:])
            vae_sd[name] = (torch.randn(shp, device=dev, generator=g) / math.sqrt(fan_in)).to(BF16)
    model = Bagel(cfg, lambda n: sd[n], device=dev, visual_gen=True, visual_und=True)
    vae = AutoEncoder(cfg, lambda n: vae_sd[n], device=dev)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    oracle = OracleBagel(cfg.to_dict(), {k: v.cpu() for k, v in sd.items()}, {k: v.cpu() for k, v in vae_sd.items()},
                         attn_impl="flash")
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    return model, vae, oracle, cfg, ntid


def _prompts(lens, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(1000, 150000, (n,), generator=g).tolist() for n in lens]


def _close(got, ref, rtol, what):
    got, ref = got.float().cpu(), ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().clamp_min(1e-6)
    d = (got - ref).abs()
    assert d.max() <= rtol * scale, f"{what}: max err {d.max():.4g} vs range {scale:.4g} ({(d.max() / scale):.4f})"
    assert d.mean() <= 0.1 * rtol * scale, f"{what}: mean err {d.mean():.4g} vs range {scale:.4g}"


def _check_kv(cache, ocache, layers, rtol, what):
    for l in layers:
        _close(cache.packed_keys(l), torch.cat(ocache.k[l], 0), rtol, f"{what} K layer {l}")
        _close(cache.packed_values(l), torch.cat(ocache.v[l], 0), rtol, f"{what} V layer {l}")


def _forced_decode_check(model, oracle, cache, ocache, kvl, rope, ntid, steps, what, atol=0.25, start_tokens=None):
    """engine: free greedy decode under the HIP graph; oracle: fed the engine's inputs step by step.  start_tokens: continue a
    previous session (its last predicted ids) instead of starting from bos."""
    from unimedvl_amd.decode import DecodeSession
    B = len(kvl)
    gi = model.prepare_start_tokens(kvl, rope, ntid)
    if start_tokens is not None:
        gi["packed_start_tokens"] = start_tokens.to(gi["packed_start_tokens"].dtype)
    sess = DecodeSession(model.language_model, cache, gi["packed_start_tokens"], gi["packed_query_position_ids"], steps + 1,
                         use_graph=True)
    assert sess.graph is not None and sess.sk != (1, 1, 1) and sess.nsplit > 1, "the test is meant to pin the shipped decode step"
    pos = torch.tensor(rope, dtype=torch.long)
    worst, flips, sure_checked = 0.0, 0, 0
    for s in range(steps):
        sess.step(1)
        lg = sess.logits.float().cpu()
        fed = sess.in_ids[s].cpu()
        if s == 0:
            assert torch.equal(fed, gi["packed_start_tokens"].cpu())
        h = oracle.llm_forward(oracle.embed(fed), [1] * B, pos, ocache, True, True, "und")
        ref = oracle.lm_head(h).float()
        pos = pos + 1
        d = (lg - ref).abs().max().item()
        worst = max(worst, d)
        assert d <= atol, f"{what}: logits differ by {d} at step {s}"
        cos = torch.nn.functional.cosine_similarity(lg, ref, dim=-1).min().item()
        assert cos > 0.999, f"{what}: logits cosine {cos} at step {s}"
        top2 = ref.topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > (0.25 if atol <= 0.25 else 2 * atol)
        pred, ref_pred = sess.pred_ids[s].cpu(), ref.argmax(-1)
        assert torch.equal(lg.argmax(-1), pred), f"{what}: device argmax disagrees with the logits it was taken from (step {s})"
        assert torch.equal(pred[sure], ref_pred[sure]), f"{what}: greedy id differs at step {s} despite a top-2 margin > 0.25"
        sure_checked += int(sure.sum())
        flips += int((pred != ref_pred).sum())
    sess.commit()
    print(f"{what}: {steps} teacher-forced steps x {B}: worst |logit diff| {worst:.4f}; {flips} near-tie flips; "
          f"{sure_checked} ids checked exactly")
    sess.worst_logit_diff, sess.sure_checked, sess.flips = worst, sure_checked, flips
    return sess


def test_configs1_vqa_b8_448_prefill_and_graph_decode(fw):
    """configs[1]: batch 8, 448x448 + 32-token question (context 1026 + 34 = 1060), greedy decode"""
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B = 8
    images = [_synth_image(448, 448, 100 + i) for i in range(B)]
    prompts = _prompts([32] * B, 5)
    cache = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ntid)
    assert int(gi["packed_seqlens"].sum()) == B * 1026
    cache = model.forward_cache_update_vit(cache, **gi)
    oc = KVCache(cfg.layers, B)
    okv, orope = oracle.update_vit(oc, [0] * B, [0] * B, images, ntid)
    assert okv == kvl and orope == rope
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "after image prefill")
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTok(prompts), ntid)
    cache = model.forward_cache_update_text(cache, **gi)
    okv, orope = oracle.update_text(oc, okv, orope, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl == [1060] * B and orope == rope
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "after text prefill")
    _forced_decode_check(model, oracle, cache, oc, kvl, rope, ntid, 24, "configs[1] B=8 ctx 1060")
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "after 24 decode steps")


def test_configs3_b32_report_decode(fw):
    """configs[3]: 32 samples per GPU, ViT encode + prefill + long decode.  The 32 samples use 4 distinct images (the
    oracle prefills each once and copies its KV - samples are independent, qwen2_navit.py:602-614 - the engine prefills
    all 32) and 32 distinct ragged prompts; 64 decode steps at M = 32 rows (skinny<2,4,...>, split-K, split-KV)."""
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B, U = 32, 4
    uniq = [_synth_image(448, 448, 200 + i) for i in range(U)]
    images = [uniq[i % U] for i in range(B)]
    g = torch.Generator().manual_seed(6)
    lens = torch.randint(96, 129, (B,), generator=g).tolist()
    prompts = _prompts(lens, 7)
    cache = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ntid)
    cache.reserve(B, 1026 + 130 + 80, cfg.kv_heads, cfg.head_dim, model.device)
    cache = model.forward_cache_update_vit(cache, **gi)
    ou = KVCache(cfg.layers, U)
    oracle.update_vit(ou, [0] * U, [0] * U, uniq, ntid)
    oc = KVCache(cfg.layers, B)
    for l in range(cfg.layers):
        oc.k[l] = [ou.k[l][i % U].clone() for i in range(B)]
        oc.v[l] = [ou.v[l][i % U].clone() for i in range(B)]
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "B=32 after image prefill")
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTok(prompts), ntid)
    cache = model.forward_cache_update_text(cache, **gi)
    okv, orope = oracle.update_text(oc, [1026] * B, [1] * B, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl and orope == rope
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "B=32 after text prefill")
    _forced_decode_check(model, oracle, cache, oc, kvl, rope, ntid, 64, "configs[3] B=32")


def test_configs2_t2i_256_guided_flow_and_pixels(fw):
    """configs[2]: text-to-image 256x256, batch 4, the reference's default guidance (cfg_text 4.0, cfg_img 1.5, interval
    (0.4, 1], global renorm, shift 3.0), 5 timesteps = 4 Euler steps (3 guided + 1 plain), then the full-size VAE decoder
    and the truncating uint8 conversion for two of the images."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B, hw, steps = 4, 256, 5
    prompts = _prompts([128] * B, 8)
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], IdTok(prompts), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    cfg_text, cfg_img = NaiveCache(cfg.layers), deepcopy(gen)
    og = KVCache(cfg.layers, B)
    okv, orope = oracle.update_text(og, [0] * B, [0] * B, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl and orope == rope
    _check_kv(gen, og, range(cfg.layers), 2e-2, "T2I prompt prefill")
    torch.manual_seed(11)
    gl = model.prepare_vae_latent(kvl, rope, [(hw, hw)] * B, ntid)
    gt = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(hw, hw)] * B)
    gim = model.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)] * B)
    noise = gl["packed_init_noises"].clone()
    trace = []
    lat = model.generate_image(
        past_key_values=gen, cfg_text_past_key_values=cfg_text, cfg_img_past_key_values=cfg_img, num_timesteps=steps,
        cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
        timestep_shift=3.0, callback=lambda i, x: trace.append(x.clone()), **gl,
        cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
        cfg_text_key_values_lens=gt["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_packed_position_ids=gim["cfg_packed_position_ids"], cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
        cfg_img_key_values_lens=gim["cfg_key_values_lens"], cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"])
    # the oracle follows the reference: 'global' renorm is one norm over the WHOLE packed batch (bagel.py:1197-1198),
    # so run it per sample to match the engine's documented per-sample batch semantics (== the reference at B = 1,
    # which is all its inferencer ever uses)
    n_tok = (hw // cfg.latent_downsample) ** 2
    worst = 0.0
    pix_stats, lat_devs = [], []
    for b in range(B):
        ob = KVCache(cfg.layers, 1)
        for l in range(cfg.layers):
            ob.k[l], ob.v[l] = [og.k[l][b]], [og.v[l][b]]
        otrace = []
        olat = oracle.generate_image(
            ob, [rope[b]], [(hw, hw)], noise[b * n_tok:(b + 1) * n_tok], ntid, num_timesteps=steps, timestep_shift=3.0,
            cfg_interval=(0.4, 1.0), cfg_text_scale=4.0, cfg_text=(KVCache(cfg.layers, 1), [0]), cfg_img_scale=1.5,
            cfg_img=(ob.clone(), [rope[b]]), cfg_renorm_min=0.0, cfg_renorm_type="global", trace=otrace)
        for i, (x, ox) in enumerate(zip(trace, otrace)):
            d = (x[b * n_tok:(b + 1) * n_tok].cpu() - ox).abs()
            worst = max(worst, d.max().item())
            lat_devs.append(d.flatten())
            # measured (MI355X): round 3, exact-running-maximum attention kernels (still there: UMV_ATTN_LAZY=0): max 0.0625, mean 0.0056,
            # p99 0.030; round 5, lazy softmax reference (shipped): max 0.0625, mean up to 0.0121 at the last step.  The oracle's flash model
            # rounds P = exp(s - max over ALL keys) to bf16 in one pass; a blockwise kernel rounds P against the reference point it has at
            # that block - the running maximum (exact kernels: the final one once it has settled, hence the closer match) or the lazy
            # reference (within 2^8 of it).  Both are equally far from exact fp32 attention (tests/test_kernel_branches_gpu.py::
            # test_attn_lazy_softmax: mean 9.2e-5 vs 8.7e-5); the reference's own flash-attn kernel is blockwise too (64 / 128-key blocks).
            # Bounds: max at 2x, mean at ~1.6x the lazy measurement (= 3.5x the exact one)
            assert d.max().item() <= 0.125 and d.mean().item() < 0.02, f"sample {b} step {i}: latent max {d.max().item()} mean {d.mean().item()}"
        if b < 2:       # full-size VAE decoder + truncating uint8 (inferencer.py:234-256) on the ORACLE's latent for both
            px = vae.decode_tokens_to_uint8(olat[0], (hw, hw), model.latent_downsample, model.latent_patch_size).cpu()
            ref = oracle.decode_image(olat[0], (hw, hw))
            assert px.shape == ref.shape == (hw, hw, 3)
            diff = (px.int() - ref.int()).abs()
            dist = {k: round(100 * (diff <= k).float().mean().item(), 3) for k in (0, 1, 2, 4, 8)}
            print(f"configs[2] pixels sample {b}: % of uint8 values within k grey levels {dist}, max {diff.max().item()}, "
                  f"mean {diff.float().mean().item():.3f}")
            pix_stats.append((dist, diff.max().item()))
            # +-2 grey levels on >= 99 % of the values (the decoder is ~30 bf16 stages deep and the conversion
            # truncates, so a 1-ulp difference upstream flips a grey level)
            # measured (MI355X, round 3): 53-54 % exact, 92.3-92.5 % within 1, 99.71-99.72 % within 2, max 4-5 levels
            assert dist[2] >= 99.0 and diff.max().item() <= 8, \
                f"pixels sample {b}: {dist}, max {diff.max().item()}"
    assert gen.lens == cfg_img.lens == kvl and cfg_text.seq_lens == 0, "flow passes must not commit KV"
    allv = torch.cat(lat_devs)
    q = torch.quantile(allv[torch.randperm(allv.numel())[:2_000_000]].float(), torch.tensor([0.5, 0.9, 0.99, 0.999]))
    print(f"configs[2] B=4 256x256: latent deviation over {steps - 1} Euler steps: max {worst:.4f} (bound 0.125), mean {allv.mean().item():.5f} "
          f"(bound 0.02 per step), p50 / p90 / p99 / p99.9 = {q[0]:.4f} / {q[1]:.4f} / {q[2]:.4f} / {q[3]:.4f}")


def _t2i_args(gl, gt, gim):
    return dict(
        cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
        cfg_text_key_values_lens=gt["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_packed_position_ids=gim["cfg_packed_position_ids"], cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
        cfg_img_key_values_lens=gim["cfg_key_values_lens"], cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"], **gl)


def test_configs2_t2i_256_fifty_timesteps(fw):
    """configs[2] AS WRITTEN: 50 diffusion timesteps = 49 Euler steps (bagel.py:937-986; the default of
    interactive_image_generator.py:63), 256 x 256, the reference's default guidance (cfg_text 4.0, cfg_img 1.5, interval (0.4, 1],
    global renorm, shift 3.0) - VERDICT r03 "missing" #2: CFG multiplies rounding noise by up to 6 per guided step and only 5
    timesteps had been compared.  One request (round 6; two until then - the packed-batch case is the 5-timestep test), 
    against its own oracle run: the latent after EVERY Euler step, then the uint8 pixels of the full-size VAE decoder on the
    ENGINE's final latent against the oracle's decoder on the ORACLE's final latent (the end-to-end image, inferencer.py:234-256)."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B, hw, steps = 1, 256, 50          # (one request: the oracle's 49 x 3 CPU passes are the test's cost; packed batches: the 5-timestep test above)
    prompts = _prompts([128] * B, 18)
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], IdTok(prompts), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    cfg_text, cfg_img = NaiveCache(cfg.layers), deepcopy(gen)
    og = KVCache(cfg.layers, B)
    okv, orope = oracle.update_text(og, [0] * B, [0] * B, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl and orope == rope
    torch.manual_seed(21)
    gl = model.prepare_vae_latent(kvl, rope, [(hw, hw)] * B, ntid)
    gt = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(hw, hw)] * B)
    gim = model.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)] * B)
    noise = gl["packed_init_noises"].clone()
    trace = []
    lat = model.generate_image(
        past_key_values=gen, cfg_text_past_key_values=cfg_text, cfg_img_past_key_values=cfg_img, num_timesteps=steps,
        cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
        timestep_shift=3.0, callback=lambda i, x: trace.append(x.clone()), **_t2i_args(gl, gt, gim))
    assert len(trace) == steps - 1
    n_tok = (hw // cfg.latent_downsample) ** 2
    for b in range(B):
        ob = KVCache(cfg.layers, 1)
        for l in range(cfg.layers):
            ob.k[l], ob.v[l] = [og.k[l][b]], [og.v[l][b]]
        otrace = []
        olat = oracle.generate_image(
            ob, [rope[b]], [(hw, hw)], noise[b * n_tok:(b + 1) * n_tok], ntid, num_timesteps=steps, timestep_shift=3.0,
            cfg_interval=(0.4, 1.0), cfg_text_scale=4.0, cfg_text=(KVCache(cfg.layers, 1), [0]), cfg_img_scale=1.5,
            cfg_img=(ob.clone(), [rope[b]]), cfg_renorm_min=0.0, cfg_renorm_type="global", trace=otrace)
        assert len(otrace) == steps - 1
        per_step = []
        for i, (x, ox) in enumerate(zip(trace, otrace)):
            d = (x[b * n_tok:(b + 1) * n_tok].cpu().float() - ox.float()).abs()
            per_step.append((d.max().item(), d.mean().item()))
        mx = [p[0] for p in per_step]
        mean = [p[1] for p in per_step]
        rng = otrace[-1].float().abs().max().item()
        print(f"configs[2] 50 timesteps, sample {b}: latent |diff| max per step: step 1 {mx[0]:.4f}, 10 {mx[9]:.4f}, 25 {mx[24]:.4f}, "
              f"40 {mx[39]:.4f}, 49 {mx[48]:.4f}; worst {max(mx):.4f}; mean at the last step {mean[-1]:.5f} (latent range {rng:.2f})")
        # measured (MI355X, round 4): the deviation grows steadily over the 49 steps - max 0.0006 after step 1, 0.003 after 10, 0.005 after
        # 25, 0.011 after 40, 0.0131 / 0.0137 after 49 (two samples), mean 0.0025 at the end, on a latent range of 5.8; bounds at ~2x
        assert max(mx) <= T2I50_LAT_MAX and mean[-1] <= T2I50_LAT_MEAN, f"sample {b}: latent max {max(mx)} mean(last) {mean[-1]}"
        mine = lat[b] if isinstance(lat, (list, tuple)) else lat[b * n_tok:(b + 1) * n_tok]
        px = vae.decode_tokens_to_uint8(mine, (hw, hw), model.latent_downsample, model.latent_patch_size).cpu()
        ref = oracle.decode_image(olat[0], (hw, hw))
        assert px.shape == ref.shape == (hw, hw, 3)
        diff = (px.int() - ref.int()).abs()
        dist = {k: round(100 * (diff <= k).float().mean().item(), 3) for k in (0, 1, 2, 4, 8, 16)}
        print(f"configs[2] 50 timesteps, sample {b}: END-TO-END uint8 pixels (engine latent -> engine VAE vs oracle latent -> oracle VAE): "
              f"% within k grey levels {dist}, max {diff.max().item()}, mean {diff.float().mean().item():.3f}")
        # measured: 48.5-48.9 % exact, 88.2-88.4 % within 1, 99.06-99.08 % within 2, 99.997 % within 4, max 5 levels (SURVEY 8c's +-2 on >= 99 %)
        assert dist[2] >= T2I50_PIX_WITHIN2 and dist[4] >= T2I50_PIX_WITHIN4 and diff.max().item() <= T2I50_PIX_MAX, \
            f"pixels sample {b}: {dist}, max {diff.max().item()}"
    assert gen.lens == cfg_img.lens == kvl and cfg_text.seq_lens == 0, "flow passes must not commit KV"


# bounds of the 50-timestep test (measured distribution in DESIGN.md section 3)
T2I50_LAT_MAX, T2I50_LAT_MEAN, T2I50_PIX_WITHIN2, T2I50_PIX_WITHIN4, T2I50_PIX_MAX = 0.03, 0.006, 98.5, 99.9, 10


def test_t2i_packed_batch_global_renorm_reference_semantics(fw):
    """cfg_renorm_batch_semantics="reference" (VERDICT r03 "missing" #4): the reference's generate_image takes ONE norm over the
    whole packed batch for cfg_renorm_type="global" (bagel.py:1197-1198), which couples the samples.  B = 2 packed, against the
    oracle run on the same PACKED batch (it restates that line as written); and the default per-sample semantics must differ
    from it (otherwise the switch tests nothing)."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B, hw, steps = 2, 256, 6
    prompts = _prompts([64, 128], 28)
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], IdTok(prompts), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    og = KVCache(cfg.layers, B)
    okv, orope = oracle.update_text(og, [0] * B, [0] * B, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl and orope == rope
    torch.manual_seed(31)
    gl = model.prepare_vae_latent(kvl, rope, [(hw, hw)] * B, ntid)
    gt = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(hw, hw)] * B)
    gim = model.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)] * B)
    noise = gl["packed_init_noises"].clone()
    # make the two samples' velocity norms differ a lot, so that one shared norm and two private ones cannot agree by accident
    n_tok = (hw // cfg.latent_downsample) ** 2
    noise[n_tok:] *= 3.0
    gl["packed_init_noises"] = noise.clone()
    out = {}
    for sem in ("reference", "per_sample"):
        trace = []
        model.generate_image(
            past_key_values=gen, cfg_text_past_key_values=NaiveCache(cfg.layers), cfg_img_past_key_values=deepcopy(gen),
            num_timesteps=steps, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.0, 1.0), cfg_renorm_min=0.0,
            cfg_renorm_type="global", timestep_shift=3.0, callback=lambda i, x: trace.append(x.clone()),
            cfg_renorm_batch_semantics=sem, **_t2i_args(dict(gl, packed_init_noises=noise.clone()), gt, gim))
        out[sem] = [x.float().cpu() for x in trace]
    otrace = []
    oracle.generate_image(
        og, rope, [(hw, hw)] * B, noise, ntid, num_timesteps=steps, timestep_shift=3.0, cfg_interval=(0.0, 1.0),
        cfg_text_scale=4.0, cfg_text=(KVCache(cfg.layers, B), [0] * B), cfg_img_scale=1.5, cfg_img=(og.clone(), list(rope)),
        cfg_renorm_min=0.0, cfg_renorm_type="global", trace=otrace)
    assert len(otrace) == len(out["reference"]) == steps - 1
    worst = max((x - ox.float()).abs().max().item() for x, ox in zip(out["reference"], otrace))
    mean = max((x - ox.float()).abs().mean().item() for x, ox in zip(out["reference"], otrace))
    gap = (out["per_sample"][-1] - otrace[-1].float()).abs().max().item()
    print(f"packed-batch global renorm (reference semantics), B = 2, {steps - 1} guided Euler steps: latent |diff| vs the oracle's packed "
          f"batch max {worst:.4f} mean {mean:.5f}; the per-sample default differs from the packed reference by {gap:.3f}")
    # measured (MI355X, round 4): max 0.0801, mean 0.0132 over the 5 guided steps (the second sample's noise is 3x, so are its latents);
    # the per-sample default is 0.473 away from the packed reference; bounds at 2x
    assert worst <= 0.16 and mean <= 0.027, f"reference batch semantics: latent max {worst} mean {mean}"
    assert gap > 4 * worst, "per-sample and packed-batch renorm should differ visibly on this input"


def test_configs1_long_free_running_decode_256(fw):
    """VERDICT r03 "missing" #3: a LONG free-running greedy decode at full width (bagel.py:1262-1314 runs to 512 tokens,
    interactive_vqa_inferencer.py:64).  B = 8, 448 x 448 + 32-token question (context 1060), 256 steps under the HIP graph in two
    sessions of 128: the second one makes the KV slabs GROW (capacity 1280 -> 2560: NaiveCache.ensure copies the committed
    keys into new slabs and the step is re-captured on the new addresses) and changes the key-split count of the decode
    attention (25 -> 28 splits).  The engine decodes freely; the oracle is fed the engine's tokens, so every one of the 2048
    greedy ids is checked (exactly, wherever the oracle's top-2 margin exceeds 0.25) and one near-tie cannot end the comparison."""
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    B = 8
    images = [_synth_image(448, 448, 400 + i) for i in range(B)]
    prompts = _prompts([32] * B, 41)
    cache = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ntid)
    cache = model.forward_cache_update_vit(cache, **gi)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTok(prompts), ntid)
    cache = model.forward_cache_update_text(cache, **gi)
    oc = KVCache(cfg.layers, B)
    okv, orope = oracle.update_vit(oc, [0] * B, [0] * B, images, ntid)
    okv, orope = oracle.update_text(oc, okv, orope, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl == [1060] * B and orope == rope
    cap0 = cache.cap
    s1 = _forced_decode_check(model, oracle, cache, oc, kvl, rope, ntid, 128, "long decode, steps 1-128")
    assert cache.cap == cap0 and cache.lens == [1060 + 128] * B
    last = s1.pred_ids[127].clone()
    kvl2, rope2 = [k + 128 for k in kvl], [r + 128 for r in rope]
    s1.graph = None
    s2 = _forced_decode_check(model, oracle, cache, oc, kvl2, rope2, ntid, 128, "long decode, steps 129-256 (after slab growth)",
                              start_tokens=last)
    assert cache.cap > cap0, f"the second session was meant to grow the slabs ({cap0} -> {cache.cap})"
    assert s2.nsplit != s1.nsplit, f"the second session was meant to change the key-split count ({s1.nsplit} -> {s2.nsplit})"
    assert cache.lens == [1060 + 256] * B
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "after 256 decode steps")
    print(f"long free-running decode: 256 steps x {B}, slab capacity {cap0} -> {cache.cap}, key splits {s1.nsplit} -> {s2.nsplit}; "
          f"worst |logit diff| {max(s1.worst_logit_diff, s2.worst_logit_diff):.4f}; {s1.sure_checked + s2.sure_checked} of {2 * 128 * B} ids "
          f"checked exactly (margin > 0.25), {s1.flips + s2.flips} near-tie flips")


@pytest.mark.parametrize("act8", [False, True])
def test_configs4_fp8_weights_fullwidth(fw, act8):
    """configs[4]: e4m3 LLM weights at full width.  PARITY UNPINNED BY THE REFERENCE (it has no fp8 path): the checker is
    the bf16 oracle on the DEQUANTISED weights (oracle/fp8.py: W' = q * 2^e is exact in bf16), in W8A8 mode with the linear
    inputs of every non-decode forward rounded per row through e4m3.  act8=False: weight-only (decode streams the e4m3
    image through gemm_skinny8, prefill runs the bf16 tiled kernel on W') at the bf16 tolerances; act8=True: prefill on the
    fp8 matrix instruction (gemm_tiled8) at the W8A8 bounds of tests/test_fp8_gpu.py (a rounding step of e4m3 is 2^-3
    relative, so one bf16 ulp upstream can move an activation by a whole step)."""
    from oracle import fp8
    from oracle.unimedvl_cpu import KVCache, OracleBagel
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    sd_cpu = oracle.sd
    cfg8 = UniMedVLConfig.from_dict(cfg.to_dict())
    cfg8.llm_weight_dtype = "fp8"
    cfg8.llm_act_dtype = "fp8" if act8 else "bf16"
    m8 = Bagel(cfg8, lambda n: sd_cpu[n], device=model.device, visual_gen=False, visual_und=True)
    w = m8.language_model.w
    assert w.fp8 and w.und[0].gate_up.w8 is not None and (w.und[0].qkv.w8m is not None) == act8
    if not hasattr(test_configs4_fp8_weights_fullwidth, "_deq"):     # 2 G weights through the CPU quantiser: once
        test_configs4_fp8_weights_fullwidth._deq = fp8.dequantised_weights(sd_cpu)
    o8 = OracleBagel(cfg.to_dict(), test_configs4_fp8_weights_fullwidth._deq, None, attn_impl="flash", act_fp8=act8)
    B = 4
    images = [_synth_image(448, 448, 300 + i) for i in range(B)]
    prompts = _prompts([32] * B, 9)
    cache = NaiveCache(cfg.layers)
    gi, kvl, rope = m8.prepare_vit_images([0] * B, [0] * B, images, lambda x: x, ntid)
    cache = m8.forward_cache_update_vit(cache, **gi)
    gi, kvl, rope = m8.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTok(prompts), ntid)
    cache = m8.forward_cache_update_text(cache, **gi)
    oc = KVCache(cfg.layers, B)
    okv, orope = o8.update_vit(oc, [0] * B, [0] * B, images, ntid)
    okv, orope = o8.update_text(oc, okv, orope, [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts])
    assert okv == kvl and orope == rope
    if not act8:
        _check_kv(cache, oc, range(cfg.layers), 2e-2, "fp8 weights, after prefill")
        _forced_decode_check(m8, o8, cache, oc, kvl, rope, ntid, 12, "configs[4] fp8 weights B=4")
    else:
        L = cfg.layers
        kref, kgot = torch.cat(oc.k[L - 1], 0).float(), cache.packed_keys(L - 1).float().cpu()
        rel = ((kgot - kref).norm() / kref.norm()).item()
        print(f"W8A8 full width: last-layer keys rel fro {rel:.4f}")
        assert rel <= 0.08, f"W8A8 last-layer keys: relative Frobenius error {rel}"       # measured 0.058 (tiny model: 0.04)
        _forced_decode_check(m8, o8, cache, oc, kvl, rope, ntid, 12, "configs[4] W8A8 B=4")   # measured worst 0.0625
    del m8
    torch.cuda.empty_cache()


def test_edit_path_448_vae_encode_and_gen_prefill(fw):
    """The image-editing input path at FULL size (VERDICT r02 "next" #3c): AutoEncoder.encode of a 448 x 448 image
    (autoencoder.py:300-303: 829 GFLOP of convolutions, 128 -> 512 channels, the mid-block attention over 3136 positions at
    hd 512), the sample z = mean + std * noise with injected noise, vae2llm + timestep-0 embedding + latent position table
    (bagel.py:757-790) and the gen-mode (MoT, non-causal) prefill of the 786-token span through both experts
    (bagel.py:697-806, qwen2_navit.py:552-562,891-898) - against the CPU oracle on the same weights, image and noise."""
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, vae, oracle, cfg, ntid = fw
    img = _synth_image(448, 448, 555)
    g = torch.Generator().manual_seed(556)
    noise = torch.randn(1, cfg.z_channels, 448 // 8, 448 // 8, generator=g).to(BF16)
    # ---- the VAE encoder alone (the latent the LLM is fed)
    lat = vae.encode(img[None], noise=noise).float().cpu()
    ref = oracle.vae_encode(img[None].to(BF16), noise).float()
    assert lat.shape == ref.shape == (1, cfg.z_channels, 56, 56)
    d = (lat - ref).abs()
    q = torch.quantile(d.flatten(), torch.tensor([0.5, 0.99]))
    print(f"edit path 448x448: VAE latent |diff| max {d.max().item():.4f} mean {d.mean().item():.5f} p50 {q[0]:.4f} p99 {q[1]:.4f} "
          f"(latent range {ref.abs().max().item():.2f})")
    # measured on MI355X (r03, input-stationary convolutions): max 0.219, mean 0.0022, p50 0.0010, p99 0.0156 of a latent range of
    # 6.28 - the tail is a handful of elements where std * noise amplifies a one-ulp bf16 difference in the log-variance.
    rng = ref.abs().max().item()
    assert d.max().item() <= 0.05 * rng and q[1].item() <= 0.005 * rng and d.mean().item() <= 0.001 * rng, \
        f"VAE encode: max {d.max().item()} p99 {q[1].item()} mean {d.mean().item()} vs range {rng}"
    # ---- the whole gen-mode prefill
    cache = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_vae_images([0], [0], [img], lambda x: x, ntid)
    cache = model.forward_cache_update_vae(vae, cache, noise=noise, **gi)
    oc = KVCache(cfg.layers, 1)
    okv, orope = oracle.update_vae(oc, [0], [0], [img], ntid, noise=noise)
    assert okv == kvl == [28 * 28 + 2] and orope == rope
    _check_kv(cache, oc, range(cfg.layers), 2e-2, "edit path, gen-mode prefill of the 786-token span")


def _edit_contexts(model, vae, oracle, cfg, ntid, imgs_vae, imgs_vit, prompts, noises):
    """The three contexts interleave_inference builds for [image, text] with understanding_output=False (inferencer.py:587-607):
    gen = VAE span + ViT span + text; cfg_text = gen before the text item (the image alone); cfg_img = the text alone.
    Built through the engine's prepare_* / forward_cache_update_* and through the oracle's update_*; returns both sides."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    B = len(prompts)
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_vae_images([0] * B, [0] * B, imgs_vae, lambda x: x, ntid)
    gen = model.forward_cache_update_vae(vae, gen, noise=noises, **gi)
    gi, kvl, rope = model.prepare_vit_images(kvl, rope, imgs_vit, lambda x: x, ntid)
    gen = model.forward_cache_update_vit(gen, **gi)
    cfg_text, kvl_t, rope_t = deepcopy(gen), list(kvl), list(rope)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTok(prompts), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    cfg_img = NaiveCache(cfg.layers)
    gi, kvl_i, rope_i = model.prepare_prompts([0] * B, [0] * B, [str(i) for i in range(B)], IdTok(prompts), ntid)
    cfg_img = model.forward_cache_update_text(cfg_img, **gi)
    # oracle
    ids = [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts]
    og = KVCache(cfg.layers, B)
    okv, orope = oracle.update_vae(og, [0] * B, [0] * B, imgs_vae, ntid, noise=noises)
    okv, orope = oracle.update_vit(og, okv, orope, imgs_vit, ntid)
    ot, okv_t, orope_t = og.clone(), list(okv), list(orope)
    okv, orope = oracle.update_text(og, okv, orope, ids)
    oi = KVCache(cfg.layers, B)
    okv_i, orope_i = oracle.update_text(oi, [0] * B, [0] * B, ids)
    assert (okv, orope, okv_t, orope_t, okv_i, orope_i) == (kvl, rope, kvl_t, rope_t, kvl_i, rope_i)
    return (gen, kvl, rope), (cfg_text, kvl_t, rope_t), (cfg_img, kvl_i, rope_i), (og, ot, oi)


def _edit_flow(model, oracle, cfg, ntid, ctx, shapes, steps, seed):
    """FlowSession with the generator script's edit settings (interactive_image_generator.py:303-306,365-371: cfg_text 4.0, cfg_img 2.0,
    interval [0, 1], text_channel renorm, shift 3.0) over three DISTINCT contexts, against the oracle's generate_image on the same
    noise; returns (engine latents, oracle latents, per-step (max, mean) deviations)."""
    from unimedvl_amd.bagel import FlowSession
    (gen, kvl, rope), (cfg_text, kvl_t, rope_t), (cfg_img, kvl_i, rope_i), (og, ot, oi) = ctx
    torch.manual_seed(seed)
    gl = model.prepare_vae_latent(kvl, rope, shapes, ntid)
    gt = model.prepare_vae_latent_cfg(kvl_t, rope_t, shapes)
    gim = model.prepare_vae_latent_cfg(kvl_i, rope_i, shapes)
    noise = gl["packed_init_noises"].clone()
    args = dict(gl)
    args.update(dict(past_key_values=gen, num_timesteps=steps, timestep_shift=3.0, cfg_renorm_min=0.0, cfg_renorm_type="text_channel",
                     cfg_interval=(0.0, 1.0), cfg_text_scale=4.0, cfg_img_scale=2.0,
                     cfg_text_past_key_values=cfg_text, cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
                     cfg_img_past_key_values=cfg_img, cfg_img_packed_position_ids=gim["cfg_packed_position_ids"]))
    sess = FlowSession(model, args)
    assert sess.nctx == 3 and sess.use_img and not sess.img_same, "the edit flow must run three DISTINCT contexts per guided step"
    trace = []
    while not sess.finished:
        sess.step(1)
        trace.append(sess.x_t.clone())
    assert len(trace) == steps - 1
    otrace = []
    olat = oracle.generate_image(og, rope, shapes, noise, ntid, num_timesteps=steps, timestep_shift=3.0, cfg_interval=(0.0, 1.0),
                                 cfg_text_scale=4.0, cfg_text=(ot, rope_t), cfg_img_scale=2.0, cfg_img=(oi, rope_i),
                                 cfg_renorm_min=0.0, cfg_renorm_type="text_channel", trace=otrace)
    per_step = []
    for x, ox in zip(trace, otrace):
        d = (x.cpu().float() - ox.float()).abs()
        per_step.append((d.max().item(), d.mean().item()))
    assert list(gen.lens) == kvl and list(cfg_text.lens) == kvl_t and list(cfg_img.lens) == kvl_i, "flow passes must not commit KV"
    _edit_flow.last = dict(noise=noise, trace=[t.cpu().float() for t in trace], otrace=[t.float() for t in otrace], ropes=(rope, rope_t, rope_i))
    return sess.latents(), olat, per_step, otrace[-1].float().abs().max().item()


def _oracle_edit_run(oracle, cfg, ntid, imgs_vae, imgs_vit, prompts, noises, shapes, init_noise, steps):
    """The oracle half of _edit_contexts + _edit_flow for another OracleBagel (the yardstick run with the other attention model)."""
    from oracle.unimedvl_cpu import KVCache
    B = len(prompts)
    ids = [[ntid["bos_token_id"]] + p + [ntid["eos_token_id"]] for p in prompts]
    og = KVCache(cfg.layers, B)
    okv, orope = oracle.update_vae(og, [0] * B, [0] * B, imgs_vae, ntid, noise=noises)
    okv, orope = oracle.update_vit(og, okv, orope, imgs_vit, ntid)
    ot, orope_t = og.clone(), list(orope)
    okv, orope = oracle.update_text(og, okv, orope, ids)
    oi = KVCache(cfg.layers, B)
    okv_i, orope_i = oracle.update_text(oi, [0] * B, [0] * B, ids)
    otrace = []
    olat = oracle.generate_image(og, orope, shapes, init_noise, ntid, num_timesteps=steps, timestep_shift=3.0, cfg_interval=(0.0, 1.0),
                                 cfg_text_scale=4.0, cfg_text=(ot, orope_t), cfg_img_scale=2.0, cfg_img=(oi, orope_i),
                                 cfg_renorm_min=0.0, cfg_renorm_type="text_channel", trace=otrace)
    return olat, [t.float() for t in otrace]


def _pix_dist(a, b):
    diff = (a.int() - b.int()).abs()
    return {k: round(100 * (diff <= k).float().mean().item(), 3) for k in (0, 1, 2, 4, 8, 16)}, int(diff.max()), float(diff.float().mean())


# bounds of the edit-pipeline tests (measured distribution printed by the tests with -s and quoted in DESIGN.md section 3)
# measured on MI355X (round 5, 512 x 512, 11 guided steps at cfg_text 4.0 / cfg_img 2.0): latent |diff| max per step 0.011 .. 0.162 (range 8.5),
# mean at the last step 0.026; uint8 pixels 35 % equal, 70 % within 1, 91.8 % within 2, 99.7 % within 4, 100 % within 8, max 9, mean 1.05.
# (The guidance combine v_text + 4 (v - v_text) multiplies the bf16 rounding differences of the three passes; the pure T2I flow at
# cfg 4.0 / 1.5 over part of the interval ends at 99 % within 2.)  Bounds = ~2x the measured tails.
EDIT_LAT_MAX, EDIT_LAT_MEAN, EDIT_PIX_WITHIN2, EDIT_PIX_WITHIN4, EDIT_PIX_MAX = 0.32, 0.05, 85.0, 99.0, 20


def test_edit_pipeline_512_three_contexts_text_channel(fw):
    """The reference's flagship edit flow at FULL width (interactive_image_generator.py:290-397 -> inferencer.py:587-607 ->
    bagel.py:1138-1186): a 448 x 448 input becomes a 512 x 512 VAE image (vae_transform min side 512, inferencer.py:42-70) and a
    448 x 448 ViT image; gen context = VAE span (1026) + ViT span (1026) + 32-token instruction, cfg_text = the image alone,
    cfg_img = the instruction alone - three DISTINCT contexts (asserted), so every guided step is three LLM passes over 1026
    query tokens combined by the `text_channel` renorm (bagel.py:1173-1186) with cfg_text 4.0 / cfg_img 2.0 over the whole interval.
    12 timesteps = 11 guided Euler steps; latent after every step and the final uint8 pixels (engine latent -> engine VAE against
    oracle latent -> oracle VAE at 512 x 512) against OracleBagel."""
    model, vae, oracle, cfg, ntid = fw
    img448 = _synth_image(448, 448, 701)
    img512 = torch.nn.functional.interpolate(img448[None], size=(512, 512), mode="bicubic", align_corners=False)[0].clamp(-1, 1).contiguous()
    g = torch.Generator().manual_seed(702)
    noise = torch.randn(1, cfg.z_channels, 512 // 8, 512 // 8, generator=g).to(BF16)
    prompts = _prompts([32], 703)
    ctx = _edit_contexts(model, vae, oracle, cfg, ntid, [img512], [img448], prompts, noise)
    (gen, kvl, rope), (cfg_text, kvl_t, _), (cfg_img, kvl_i, _), (og, ot, oi) = ctx
    assert kvl == [1026 + 1026 + 34] and kvl_t == [2052] and kvl_i == [34]
    _check_kv(gen, og, range(cfg.layers), 2e-2, "edit pipeline, gen context (VAE + ViT + text)")
    _check_kv(cfg_img, oi, range(cfg.layers), 2e-2, "edit pipeline, cfg_img context (text alone)")
    steps = 12
    lat, olat, per_step, rng = _edit_flow(model, oracle, cfg, ntid, ctx, [(512, 512)], steps, 704)
    mx = [p[0] for p in per_step]
    print(f"edit pipeline 512x512, 3 distinct contexts, text_channel: latent |diff| max per step {[round(v, 4) for v in mx]}; "
          f"mean at the last step {per_step[-1][1]:.5f} (latent range {rng:.2f})")
    assert max(mx) <= EDIT_LAT_MAX and per_step[-1][1] <= EDIT_LAT_MEAN, f"latent max {max(mx)} mean(last) {per_step[-1][1]}"
    px = vae.decode_tokens_to_uint8(lat[0], (512, 512), model.latent_downsample, model.latent_patch_size).cpu()
    ref = oracle.decode_image(olat[0], (512, 512))
    assert px.shape == ref.shape == (512, 512, 3)
    diff = (px.int() - ref.int()).abs()
    dist = {k: round(100 * (diff <= k).float().mean().item(), 3) for k in (0, 1, 2, 4, 8, 16)}
    print(f"edit pipeline 512x512: END-TO-END uint8 pixels: % within k grey levels {dist}, max {diff.max().item()}, mean {diff.float().mean().item():.3f}")
    assert dist[2] >= EDIT_PIX_WITHIN2 and dist[4] >= EDIT_PIX_WITHIN4 and diff.max().item() <= EDIT_PIX_MAX, f"pixels: {dist}, max {diff.max().item()}"
    # ---- an INDEPENDENT yardstick for those bounds: the same request through the oracle's OTHER attention model
    # ("sdpa": exact fp32 softmax, the branch the reference goldens pin; the fixture's oracle is "flash": P rounded to bf16).  Two equally
    # valid CPU formulations of the same math differ by the guidance-amplified rounding noise; the engine must not be further from the
    # flash oracle than the sdpa oracle is (x 1.25 for run-to-run spread of a 35-stage bf16 pipeline), latents and end-to-end pixels.
    from oracle.unimedvl_cpu import OracleBagel
    o2 = OracleBagel(cfg.to_dict(), oracle.sd, oracle.vae_sd, attn_impl="sdpa")
    last = _edit_flow.last
    olat2, otrace2 = _oracle_edit_run(o2, cfg, ntid, [img512], [img448], prompts, noise, [(512, 512)], last["noise"], steps)
    yard = [((a - b).abs().max().item(), (a - b).abs().mean().item()) for a, b in zip(otrace2, last["otrace"])]
    eng2 = [((a - b).abs().max().item(), (a - b).abs().mean().item()) for a, b in zip(last["trace"], otrace2)]
    print(f"edit pipeline yardstick: oracle sdpa vs flash latent |diff| max per step {[round(v[0], 4) for v in yard]}, mean(last) {yard[-1][1]:.5f}; "
          f"engine vs oracle(sdpa) max per step {[round(v[0], 4) for v in eng2]}, mean(last) {eng2[-1][1]:.5f}")
    ref2 = oracle.decode_image(olat2[0], (512, 512))
    ydist, ymax, ymean = _pix_dist(ref2, ref)
    e2dist, e2max, e2mean = _pix_dist(px, ref2)
    print(f"edit pipeline yardstick pixels: oracle sdpa vs flash {ydist} max {ymax} mean {ymean:.3f}; engine vs oracle(sdpa) {e2dist} max {e2max} mean {e2mean:.3f}")
    assert max(mx) <= 1.25 * max(v[0] for v in yard) and per_step[-1][1] <= 1.25 * yard[-1][1], \
        f"engine further from the flash oracle (max {max(mx):.4f}, mean {per_step[-1][1]:.5f}) than the sdpa oracle is (max {max(v[0] for v in yard):.4f}, mean {yard[-1][1]:.5f})"
    assert dist[2] >= ydist[2] - 2.0 and dist[4] >= ydist[4] - 1.0, f"pixels: engine vs flash {dist} against sdpa vs flash {ydist}"


def test_edit_pipeline_512_fifty_timesteps(fw):
    """The same request at the generator script's OWN schedule (interactive_image_generator.py:303-306,365-371: 50 timesteps = 49 guided Euler
    steps over three distinct contexts) - once, latents after every step and end-to-end pixels against the flash oracle.  What 12 timesteps
    cannot show is how the deviation behaves over the long, small-step schedule: it is printed per decile."""
    model, vae, oracle, cfg, ntid = fw
    img448 = _synth_image(448, 448, 701)
    img512 = torch.nn.functional.interpolate(img448[None], size=(512, 512), mode="bicubic", align_corners=False)[0].clamp(-1, 1).contiguous()
    g = torch.Generator().manual_seed(702)
    noise = torch.randn(1, cfg.z_channels, 512 // 8, 512 // 8, generator=g).to(BF16)
    prompts = _prompts([32], 703)
    ctx = _edit_contexts(model, vae, oracle, cfg, ntid, [img512], [img448], prompts, noise)
    steps = 50
    lat, olat, per_step, rng = _edit_flow(model, oracle, cfg, ntid, ctx, [(512, 512)], steps, 704)
    mx = [p[0] for p in per_step]
    print(f"edit pipeline 512x512, 50 timesteps: latent |diff| max at steps 0/5/../45/48: {[round(mx[i], 4) for i in list(range(0, 49, 5)) + [48]]}; "
          f"mean at the last step {per_step[-1][1]:.5f} (latent range {rng:.2f})")
    assert max(mx) <= EDIT_LAT_MAX and per_step[-1][1] <= EDIT_LAT_MEAN, f"latent max {max(mx)} mean(last) {per_step[-1][1]}"
    px = vae.decode_tokens_to_uint8(lat[0], (512, 512), model.latent_downsample, model.latent_patch_size).cpu()
    ref = oracle.decode_image(olat[0], (512, 512))
    dist, dmax, dmean = _pix_dist(px, ref)
    print(f"edit pipeline 512x512, 50 timesteps: END-TO-END uint8 pixels: % within k grey levels {dist}, max {dmax}, mean {dmean:.3f}")
    assert dist[2] >= EDIT_PIX_WITHIN2 and dist[4] >= EDIT_PIX_WITHIN4 and dmax <= EDIT_PIX_MAX, f"pixels: {dist}, max {dmax}"


def test_edit_pipeline_ragged_batch_of_two(fw):
    """The same flow as a RAGGED packed batch of two requests (the batch extension of the entry points): different input / output
    sizes (256 x 384 and 320 x 320 VAE images, 224 x 336 and 280 x 280 ViT images) and instruction lengths (20 / 32 tokens); the
    `text_channel` renorm is per token, so the packed oracle run is the per-request reference."""
    model, vae, oracle, cfg, ntid = fw
    sizes_vae, sizes_vit = [(256, 384), (320, 320)], [(224, 336), (280, 280)]
    imgs_vae = [_synth_image(h, w, 711 + i) for i, (h, w) in enumerate(sizes_vae)]
    imgs_vit = [torch.nn.functional.interpolate(im[None], size=s, mode="bilinear", align_corners=False)[0].contiguous()
                for im, s in zip(imgs_vae, sizes_vit)]
    g = torch.Generator().manual_seed(712)
    mh, mw = max(h for h, _ in sizes_vae), max(w for _, w in sizes_vae)
    noise = torch.randn(2, cfg.z_channels, mh // 8, mw // 8, generator=g).to(BF16)
    prompts = _prompts([20, 32], 713)
    ctx = _edit_contexts(model, vae, oracle, cfg, ntid, imgs_vae, imgs_vit, prompts, noise)
    (gen, kvl, rope), _, _, (og, ot, oi) = ctx
    _check_kv(gen, og, range(cfg.layers), 2e-2, "ragged edit batch, gen context")
    lat, olat, per_step, rng = _edit_flow(model, oracle, cfg, ntid, ctx, sizes_vae, 7, 714)
    mx = [p[0] for p in per_step]
    print(f"ragged edit batch (256x384 + 320x320): latent |diff| max per step {[round(v, 4) for v in mx]}; mean at the last step "
          f"{per_step[-1][1]:.5f} (latent range {rng:.2f})")
    assert max(mx) <= EDIT_LAT_MAX and per_step[-1][1] <= EDIT_LAT_MEAN
    for b, (h, w) in enumerate(sizes_vae):
        assert lat[b].shape == olat[b].shape == ((h // 16) * (w // 16), 64)


def test_configs4_mixed_fullwidth(fw):
    """configs[4] as written: e4m3 weights + e4m3 activations (W8A8), VQA requests and text-to-image requests IN ONE STEP
    STREAM (serving.MixedBatcher) at the full widths: 4 VQA slots (448 x 448 image + 32-token question) decode while a
    group of two 256 x 256 images goes through its guided flow passes on the fp8 matrix instruction and the full-size VAE.
    PARITY UNPINNED BY THE REFERENCE (no fp8 path there): the checker is OracleBagel(act_fp8=True) on the dequantised
    weights (oracle/fp8.py).  Checked: (1) the mixed run reproduces each request served alone bit for bit (greedy ids /
    final latents); (2) the W8A8 latents after every Euler step against the oracle, measured deviation printed next to the
    bound; (3) uint8 pixels through the full-size VAE."""
    from oracle import fp8
    from oracle.unimedvl_cpu import KVCache, OracleBagel
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.kvcache import NaiveCache
    from unimedvl_amd.serving import MixedBatcher
    model, vae, oracle, cfg, ntid = fw
    sd_cpu = oracle.sd
    cfg8 = UniMedVLConfig.from_dict(cfg.to_dict())
    cfg8.llm_weight_dtype, cfg8.llm_act_dtype = "fp8", "fp8"
    m8 = Bagel(cfg8, lambda n: sd_cpu[n], device=model.device, visual_gen=True, visual_und=True)
    if not hasattr(test_configs4_fp8_weights_fullwidth, "_deq"):
        test_configs4_fp8_weights_fullwidth._deq = fp8.dequantised_weights(sd_cpu)
    o8 = OracleBagel(cfg.to_dict(), test_configs4_fp8_weights_fullwidth._deq, oracle.vae_sd, attn_impl="flash", act_fp8=True)

    class Tok(IdTok):                       # ids in, ids out: the answers are compared as token strings
        def decode(self, ids):
            return "<|im_start|>" + " ".join(str(int(v)) for v in ids[1:]) + "<|im_end|>"
    nv, ni, hw, steps, new = 4, 2, 256, 5, 8
    images = [_synth_image(448, 448, 700 + i) for i in range(nv)]
    table = _prompts([32] * nv, 21) + _prompts([128] * ni, 22)
    tok = Tok(table)
    ident = lambda x: x   # noqa: E731
    n_tok = (hw // cfg.latent_downsample) ** 2
    g = torch.Generator().manual_seed(23)
    noises = [torch.randn(n_tok, cfg.latent_patch ** 2 * cfg.z_channels, generator=g) for _ in range(ni)]
    kw = dict(num_timesteps=steps, timestep_shift=3.0, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0,
              cfg_renorm_type="global")

    # ---- every request served ALONE on the same engine
    alone_ids = []
    for i in range(nv):
        c = NaiveCache(cfg.layers)
        gi, kvl, rope = m8.prepare_vit_images([0], [0], [images[i]], ident, ntid)
        c = m8.forward_cache_update_vit(c, **gi)
        gi, kvl, rope = m8.prepare_prompts(kvl, rope, [str(i)], tok, ntid)
        c = m8.forward_cache_update_text(c, **gi)
        gs = m8.prepare_start_tokens(kvl, rope, ntid)
        ids = m8.generate_text(past_key_values=c, max_length=new + 1, **gs)
        alone_ids.append([int(v) for v in ids[1:, 0]])
    alone_lat, traces = [], []
    for j in range(ni):
        gen = NaiveCache(cfg.layers)
        gi, kvl, rope = m8.prepare_prompts([0], [0], [str(nv + j)], tok, ntid)
        gen = m8.forward_cache_update_text(gen, **gi)
        gl = m8.prepare_vae_latent(kvl, rope, [(hw, hw)], ntid)
        gl["packed_init_noises"] = noises[j].clone()
        gt = m8.prepare_vae_latent_cfg([0], [0], [(hw, hw)])
        gim = m8.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)])
        tr = []
        lat = m8.generate_image(past_key_values=gen, cfg_text_past_key_values=NaiveCache(cfg.layers), cfg_img_past_key_values=gen.snapshot(),
                                callback=lambda i, x: tr.append(x.clone()), **kw, **gl,
                                cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_img_packed_position_ids=gim["cfg_packed_position_ids"])
        alone_lat.append(lat[0].clone())
        traces.append((tr, kvl, rope))

    # ---- the mixed run
    srv = MixedBatcher(m8, vae, tok, ntid, ident, slots=nv, t2i_batch=ni, flow_steps_per_round=1, max_context=1100, max_new_tokens=new,
                       check_every=2)
    rids = [srv.submit(images[i], str(i), max_new_tokens=new) for i in range(nv)]
    iids = [srv.submit_t2i(str(nv + j), (hw, hw), init_noise=noises[j], **kw) for j in range(ni)]
    got = srv.run()
    assert srv.stats["interleaved_rounds"] >= steps - 1 and srv.stats["decode_steps"] >= new and srv.stats["images"] == ni
    for rid, ids in zip(rids, alone_ids):
        want = " ".join(str(v) for v in ids)
        assert got[rid] == want[:len(got[rid])] and len(got[rid]) > 0, f"VQA request {rid}: {got[rid]!r} vs alone {want!r}"
    for iid, lat in zip(iids, alone_lat):
        assert torch.equal(srv.latents[iid], lat), f"T2I request {iid}: latent differs from the request served alone"

    # ---- W8A8 flow passes against the oracle (per image: the engine's batch semantics of 'global' renorm are per sample)
    worst, devs = 0.0, []
    for j in range(ni):
        tr, kvl, rope = traces[j]
        og = KVCache(cfg.layers, 1)
        okv, orope = o8.update_text(og, [0], [0], [[ntid["bos_token_id"]] + table[nv + j] + [ntid["eos_token_id"]]])
        assert okv == kvl and orope == rope
        otr = []
        olat = o8.generate_image(og, [rope[0]], [(hw, hw)], noises[j], ntid, num_timesteps=steps, timestep_shift=3.0, cfg_interval=(0.4, 1.0),
                                 cfg_text_scale=4.0, cfg_text=(KVCache(cfg.layers, 1), [0]), cfg_img_scale=1.5, cfg_img=(og.clone(), [rope[0]]),
                                 cfg_renorm_min=0.0, cfg_renorm_type="global", trace=otr)
        for x, ox in zip(tr, otr):
            d = (x.cpu() - ox).abs()
            worst = max(worst, d.max().item())
            devs.append(d.flatten())
        px = got[iids[j]]
        ref = o8.decode_image(olat[0], (hw, hw))
        diff = (px.int() - ref.int()).abs()
        dist = {k: round(100 * (diff <= k).float().mean().item(), 2) for k in (1, 2, 4, 8, 16)}
        print(f"configs[4] mixed, image {j}: uint8 pixels within k levels of the W8A8 oracle's {dist}, max {diff.max().item()}")
        # measured (MI355X, round 3): 27-30 % within 1 level, 61-65 % within 4, 86.5-89 % within 8, 99.4-99.6 % within 16, max 26-29
        assert dist[16] >= 98.5 and dist[8] >= 75.0, f"W8A8 pixels image {j}: {dist}"
    allv = torch.cat(devs)
    q = torch.quantile(allv[torch.randperm(allv.numel())[:2_000_000]].float(), torch.tensor([0.5, 0.9, 0.99]))
    scale = max(float(t.abs().max()) for t in traces[0][0])
    print(f"configs[4] mixed W8A8 T2I full width: latent deviation vs the oracle over {steps - 1} Euler steps: max {worst:.4f}, mean "
          f"{allv.mean().item():.5f}, p50 / p90 / p99 = {q[0]:.4f} / {q[1]:.4f} / {q[2]:.4f} (latent range {scale:.2f})")
    # rounding to e4m3 is a step function: a 1-ulp bf16 difference upstream can move an activation by a whole e4m3 step (2^-3
    # relative), and the guided steps amplify velocity differences ~6x.  Measured (MI355X, round 3): max 0.43, mean 0.039, p99 0.22
    # on latents of range 6.4; bounds = 2x that
    assert worst <= 0.9 and allv.mean().item() <= 0.08, f"W8A8 latents: max {worst} mean {allv.mean().item()}"
    del m8
    torch.cuda.empty_cache()


@pytest.mark.parametrize("paged", [False, True])
def test_continuous_batcher_fullwidth_against_the_oracle(fw, paged):
    """(paged = True: the same run on the block-table cache, kvcache.PagedCache - 256-token pages from one pool, nothing re-allocated, the
    pages of a finished request re-used - held to the oracle the same way.)
    serving.ContinuousBatcher checked against the ORACLE directly (not against Bagel.chat on the same HIP path): 7 ragged requests -
    448 x 448 / 224 x 336 / 224 x 224 images or none, 7..40-token prompts, budgets of 3..8 new tokens - on 3 slots at the 14B widths
    (2 layers), slots reserved for 256 tokens so that the first image request doubles the slabs (twice) while other slots hold
    live contexts, and four requests are admitted into slots freed in flight.  Per request the oracle rebuilds the context
    (update_vit / update_text, bagel.py:377-615) and replays the engine's tokens one step at a time (bagel.py:1262-1314, B = 1):
    every engine token must be the oracle's argmax, or lie within 0.25 of it in the oracle's own logits (teacher forcing where
    the reference's top-2 margin is inside the logit tolerance); an answer shorter than its budget must end where the oracle
    (within the same margin) predicts <|im_end|>."""
    import re
    from oracle.toy_tokenizer import ToyTokenizer
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.serving import ContinuousBatcher
    model, vae, oracle, cfg, ntid = fw
    tok = ToyTokenizer(ntid)
    names = {v: k for k, v in tok.names.items()}
    sizes = [(224, 224), None, (448, 448), (224, 336), None, (336, 224), (224, 224)]
    plens, budgets = [12, 40, 32, 7, 25, 18, 9], [6, 4, 8, 5, 7, 3, 6]
    prompts = _prompts(plens, 811)
    images = [None if sz is None else _synth_image(sz[0], sz[1], 820 + i) for i, sz in enumerate(sizes)]
    ident = lambda x: x   # noqa: E731
    srv = ContinuousBatcher(model, tok, ntid, ident, slots=3, max_context=16, max_new_tokens=8, check_every=3, use_graph=True,
                            **(dict(paged=True, pool_pages=16, context_limit=2048) if paged else {}))
    cap0 = srv.cache.cap
    rids = [srv.submit([] if im is None else [im], " ".join(str(t) for t in p), max_new_tokens=nb) for im, p, nb in zip(images, prompts, budgets)]
    got = srv.run()
    assert sorted(got) == sorted(rids) and srv.stats["prefills"] == 7
    if paged:       # 15 pages serve 7 requests of up to 1026 + 32 + 8 tokens through 3 slots because finished requests return theirs
        assert srv.stats["cache_grows"] == 0 and srv.cache.pages_in_use() <= 3, (srv.stats, srv.cache.pages_in_use())
    else:
        assert cap0 == 256 and srv.stats["cache_grows"] >= 2 and srv.cache.cap >= 1024, (cap0, srv.stats, srv.cache.cap)
    bos, eos = ntid["bos_token_id"], ntid["eos_token_id"]
    sure = flips = 0
    worst_margin = 0.0
    for rid, im, p, nb in zip(rids, images, prompts, budgets):
        ids = [names[w] if w in names else int(w) for w in re.findall(r"<\|[a-z_]+\|>|-?\d+", got[rid])]
        assert len(ids) <= nb, (rid, ids, nb)
        oc = KVCache(cfg.layers, 1)
        kv, rp = [0], [0]
        if im is not None:
            kv, rp = oracle.update_vit(oc, kv, rp, [im], ntid)
        kv, rp = oracle.update_text(oc, kv, rp, [[bos] + p + [eos]])
        pos = torch.tensor(rp, dtype=torch.long)
        fed = bos
        expect = ids + ([eos] if len(ids) < nb else [])      # a short answer must have ended on <|im_end|>
        for s, t in enumerate(expect):
            h = oracle.llm_forward(oracle.embed(torch.tensor([fed])), [1], pos, oc, True, True, "und")
            ref = oracle.lm_head(h).float()[0]
            pos = pos + 1
            top = int(ref.argmax())
            if top == t:
                sure += 1
            else:
                margin = float(ref[top] - ref[t])
                worst_margin = max(worst_margin, margin)
                assert margin <= 0.25, f"request {rid} step {s}: engine token {t}, oracle argmax {top}, margin {margin:.3f}"
                flips += 1
            fed = t
    print(f"continuous batcher vs oracle: {sure} tokens equal to the oracle's argmax, {flips} inside the 0.25 margin (largest {worst_margin:.3f}); "
          + (f"paged cache, {srv.cache.pool_pages - 1} pages of 256 tokens" if paged else
             f"slabs {cap0} -> {srv.cache.cap} tokens per slot in {srv.stats['cache_grows']} doublings"))
    assert sure >= 25 and flips <= 3
