"""FULL-DEPTH parity against the CPU oracle (VERDICT r02 "next" #3a): the real 14B understanding path - 26 SigLIP layers,
28 Qwen2-MoT layers at hidden 3584 / inter 18944 / vocab 152064 - for ONE request (a 448x448 image + a 32-token question,
context 1026 + 34 = 1060 tokens) and teacher-forced greedy decode steps under the HIP graph.

tests/test_fullwidth_gpu.py checks the shipped kernel variants at full width but 2 + 2 layers; tests/test_fullsize_gpu.py
checks 28 layers only against themselves.  Here all 54 layers run on both sides with the same seeded weights (generated on
the device, copied to the host for oracle/unimedvl_cpu.py, which is pinned bit for bit to the imported reference on the
goldens), so that what accumulates over the depth - one bf16 rounding per materialised tensor on both sides, different fp32
summation orders inside every GEMM / norm / softmax - is MEASURED against an independent restatement of the reference
(bagel.py:523-615 image prefill, :412-458 text prefill, :1236-1317 decode; qwen2_navit.py:843-902; siglip_navit.py:216-296).
The measured deviations are printed next to their bounds (`pytest -s`) and recorded in DESIGN.md section 3."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


class IdTok:
    def __init__(self, ids):
        self.ids = ids

    def encode(self, s):
        return self.ids


def _synth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 1, h // 32 + 2, w // 32 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)[0]
    img = (img / img.abs().max()).clamp(-1, 1)
    return (img.repeat(3, 1, 1) + 0.05 * torch.randn(3, h, w, generator=g)).clamp(-1, 1).contiguous()


@pytest.fixture(scope="module")
def fd():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from oracle.unimedvl_cpu import OracleBagel
    from unimedvl_amd import shapes
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.weights import random_getter
    cfg = UniMedVLConfig()                                   # every dimension AND both depths at their 14B values
    assert cfg.layers == 28 and cfg.vit_layers == 26 and cfg.hidden == 3584
    dev = torch.device("cuda", 0)
    get = random_getter(cfg, dev, seed=2828)
    skip = ("_moe_gen", "latent_pos_embed", "time_embedder", "vae2llm", "llm2vae")      # understanding path only
    names = [n for n in shapes.all_shapes(cfg) if not any(s in n for s in skip)]
    sd = {name: get(name) for name in names}
    g = torch.Generator(device=dev).manual_seed(29)
    for k, v in sd.items():       # norm gains away from 1 so that a swapped or skipped gain cannot hide
        if v.dim() == 1 and "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * torch.randn(v.shape, device=dev, generator=g)).to(BF16)
    model = Bagel(cfg, lambda n: sd[n], device=dev, visual_gen=False, visual_und=True)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    oracle = OracleBagel(cfg.to_dict(), {k: v.cpu() for k, v in sd.items()}, None, attn_impl="flash")
    del sd
    torch.cuda.empty_cache()
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    return model, oracle, cfg, ntid


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float()
    d = (got - ref).abs()
    return (d.max() / ref.abs().max().clamp_min(1e-6)).item(), ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


def test_full_depth_vqa_single_request(fd):
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.decode import DecodeSession
    from unimedvl_amd.kvcache import NaiveCache
    model, oracle, cfg, ntid = fd
    L = cfg.layers
    img = _synth_image(448, 448, 77)
    g = torch.Generator().manual_seed(78)
    prompt = torch.randint(1000, 150000, (32,), generator=g).tolist()

    cache = NaiveCache(L)
    gi, kvl, rope = model.prepare_vit_images([0], [0], [img], lambda x: x, ntid)
    cache = model.forward_cache_update_vit(cache, **gi)
    gi, kvl, rope = model.prepare_prompts(kvl, rope, ["q"], IdTok(prompt), ntid)
    cache = model.forward_cache_update_text(cache, **gi)

    oc = KVCache(L, 1)
    okv, orope = oracle.update_vit(oc, [0], [0], [img], ntid)
    okv, orope = oracle.update_text(oc, okv, orope, [[ntid["bos_token_id"]] + prompt + [ntid["eos_token_id"]]])
    assert okv == kvl == [1060] and orope == rope

    # ---- KV after the 1060-token prefill, layer by layer: the deviation a layer inherits from everything below it
    rows = []
    for l in range(L):
        mk, fk = _rel(cache.packed_keys(l), torch.cat(oc.k[l], 0))
        mv, fv = _rel(cache.packed_values(l), torch.cat(oc.v[l], 0))
        rows.append((l, mk, fk, mv, fv))
    print("full depth, KV after prefill (max err / range, relative Frobenius):")
    for l, mk, fk, mv, fv in rows:
        if l in (0, 1, 6, 13, 20, 27):
            print(f"  layer {l:2d}: K {mk:.4f} / {fk:.4f}   V {mv:.4f} / {fv:.4f}")
    # Measured on MI355X (round 3): max error 0.013 of range at layer 0, 0.031 at layer 13, 0.038 at layer 27; relative
    # Frobenius error 0.009 -> 0.024 -> 0.033: the deviation grows roughly with the square root of the depth.  Bounds: SURVEY 8c's
    # 2e-2 of range for the first two layers, 2x the measured worst case (0.08 / 0.065) below them.
    for l, mk, fk, mv, fv in rows:
        lim_max = 0.02 if l < 2 else 0.08
        assert mk <= lim_max and mv <= lim_max, f"layer {l}: K / V max error {mk:.4f} / {mv:.4f} of range (bound {lim_max})"
        assert fk <= 0.065 and fv <= 0.065, f"layer {l}: K / V relative Frobenius error {fk:.4f} / {fv:.4f} (bound 0.065)"

    # ---- teacher-forced greedy decode under the HIP graph: the oracle is fed the engine's input token of every step
    steps = 8
    gs = model.prepare_start_tokens(kvl, rope, ntid)
    sess = DecodeSession(model.language_model, cache, gs["packed_start_tokens"], gs["packed_query_position_ids"], steps + 1, use_graph=True)
    assert sess.graph is not None
    pos = torch.tensor(rope, dtype=torch.long)
    worst, worst_cos, exact = 0.0, 1.0, 0
    for s in range(steps):
        sess.step(1)
        lg = sess.logits.float().cpu()
        fed = sess.in_ids[s].cpu()
        h = oracle.llm_forward(oracle.embed(fed), [1], pos, oc, True, True, "und")
        ref = oracle.lm_head(h).float()
        pos = pos + 1
        d = (lg - ref).abs().max().item()
        cos = torch.nn.functional.cosine_similarity(lg, ref, dim=-1).min().item()
        worst, worst_cos = max(worst, d), min(worst_cos, cos)
        top2 = ref.topk(2, dim=-1).values
        margin = float(top2[0, 0] - top2[0, 1])
        pred, ref_pred = int(sess.pred_ids[s, 0]), int(ref.argmax(-1)[0])
        assert int(lg.argmax(-1)[0]) == pred
        print(f"  step {s}: |logit diff| max {d:.4f} (logit range {ref.abs().max().item():.2f}), cosine {cos:.6f}, oracle top-2 margin "
              f"{margin:.3f}, ids {pred} / {ref_pred}")
        # measured on MI355X (round 3): 0.148 - 0.164 on logits of range 5.3 - 5.7, cosine 0.99957 - 0.99962.  The bound is the STATED
        # tolerance (logits atol 0.25), not a multiple of the engine's own measurement
        assert d <= 0.25, f"step {s}: logits differ by {d} (bound 0.25)"
        assert cos > 0.999, f"step {s}: logits cosine {cos}"
        if margin > 0.25:
            assert pred == ref_pred, f"step {s}: greedy id {pred} vs oracle {ref_pred} despite a top-2 margin of {margin:.3f}"
            exact += 1
        # the engine's choice is always within the deviation bound of the oracle's best logit
        assert ref[0, pred] >= ref[0, ref_pred] - 2 * d - 1e-3
    print(f"full depth B=1 ctx 1060: {steps} teacher-forced steps, worst |logit diff| {worst:.4f}, worst cosine {worst_cos:.6f}, "
          f"{exact} ids with a decisive margin checked exactly")


@pytest.fixture(scope="module")
def fdg():
    """the full-depth model with BOTH experts (und + gen) and the image head - 29 GB of bf16 weights on each side"""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from oracle.unimedvl_cpu import OracleBagel
    from unimedvl_amd import shapes
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.weights import random_getter
    cfg = UniMedVLConfig()
    assert cfg.layers == 28 and cfg.hidden == 3584
    dev = torch.device("cuda", 0)
    get = random_getter(cfg, dev, seed=5656)
    names = [n for n in shapes.all_shapes(cfg) if not n.startswith("vit_model.") and "connector" not in n and "vit_pos_embed" not in n]
    sd = {name: get(name) for name in names}
    g = torch.Generator(device=dev).manual_seed(57)
    for k, v in sd.items():
        if v.dim() == 1 and "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * torch.randn(v.shape, device=dev, generator=g)).to(BF16)
    model = Bagel(cfg, lambda n: sd[n], device=dev, visual_gen=True, visual_und=False)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    oracle = OracleBagel(cfg.to_dict(), {k: v.cpu() for k, v in sd.items()}, None, attn_impl="flash")
    del sd
    torch.cuda.empty_cache()
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    return model, oracle, cfg, ntid


def test_full_depth_t2i_guided_flow_single_request(fdg):
    """The GENERATION path at full depth: a 32-token prompt through the 28 `und` layers, then Bagel.generate_image
    (bagel.py:901-1207) for one 128 x 128 image - 64 latent tokens through the 28 MoT layers of the `gen` expert, the two
    marker tokens through the `und` expert, timestep / position embedders, llm2vae, dual classifier-free guidance with global
    renorm - 4 timesteps = 3 Euler steps (2 guided with three contexts + 1 plain), against the CPU oracle on the same weights,
    prompt and initial noise.  What the 2-layer full-width test (tests/test_fullwidth_gpu.py configs[2]) cannot show is how
    the velocity error compounds over 28 layers and 3 steps; it is printed next to its bound."""
    from copy import deepcopy
    from oracle.unimedvl_cpu import KVCache
    from unimedvl_amd.kvcache import NaiveCache
    model, oracle, cfg, ntid = fdg
    hw, steps = 128, 4
    g = torch.Generator().manual_seed(58)
    prompt = torch.randint(1000, 150000, (32,), generator=g).tolist()
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_prompts([0], [0], ["p"], IdTok(prompt), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    cfg_text, cfg_img = NaiveCache(cfg.layers), deepcopy(gen)
    og = KVCache(cfg.layers, 1)
    okv, orope = oracle.update_text(og, [0], [0], [[ntid["bos_token_id"]] + prompt + [ntid["eos_token_id"]]])
    assert okv == kvl == [34] and orope == rope
    torch.manual_seed(59)
    gl = model.prepare_vae_latent(kvl, rope, [(hw, hw)], ntid)
    gt = model.prepare_vae_latent_cfg([0], [0], [(hw, hw)])
    gim = model.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)])
    noise = gl["packed_init_noises"].clone()
    trace = []
    model.generate_image(
        past_key_values=gen, cfg_text_past_key_values=cfg_text, cfg_img_past_key_values=cfg_img, num_timesteps=steps,
        cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
        timestep_shift=3.0, callback=lambda i, x: trace.append(x.clone()), **gl,
        cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
        cfg_text_key_values_lens=gt["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_packed_position_ids=gim["cfg_packed_position_ids"], cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
        cfg_img_key_values_lens=gim["cfg_key_values_lens"], cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"])
    otrace = []
    oracle.generate_image(
        og, [rope[0]], [(hw, hw)], noise, ntid, num_timesteps=steps, timestep_shift=3.0, cfg_interval=(0.4, 1.0), cfg_text_scale=4.0,
        cfg_text=(KVCache(cfg.layers, 1), [0]), cfg_img_scale=1.5, cfg_img=(og.clone(), [rope[0]]), cfg_renorm_min=0.0,
        cfg_renorm_type="global", trace=otrace)
    assert len(trace) == len(otrace) == steps - 1
    print("full depth, text-to-image: latent after each Euler step, engine vs oracle")
    for i, (x, ox) in enumerate(zip(trace, otrace)):
        d = (x.cpu().float() - ox.float()).abs()
        rng = ox.float().abs().max().item()
        step_size = (ox.float() - (otrace[i - 1].float() if i else noise.float())).abs().max().item()
        print(f"  step {i}: |diff| max {d.max().item():.4f} mean {d.mean().item():.5f} (latent range {rng:.2f}, largest change of this "
              f"step {step_size:.3f})")
        # Measured on MI355X (round 3): |diff| max / mean = 0.026 / 0.0055, 0.062 / 0.0125, 0.142 / 0.030 after steps 0 / 1 / 2 whose
        # largest latent changes are 0.60 / 1.16 / 2.73 (timestep_shift 3 with 4 timesteps makes the last step the big one): the
        # deviation is a steady 4.4 - 5.3 % (max) and 0.9 - 1.1 % (mean) of what the step moves, i.e. the velocity after 28 MoT
        # layers carries the same ~1 % mean / ~5 % worst-element error as the K / V of layer 27 in the understanding test above.
        # Bounds: 2x measured, relative to the step's largest change.
        assert torch.isfinite(x).all()
        assert d.max().item() <= 0.11 * step_size and d.mean().item() <= 0.023 * step_size, \
            f"step {i}: latent max {d.max().item()} mean {d.mean().item()} vs largest change {step_size}"
    assert gen.lens == cfg_img.lens == kvl and cfg_text.seq_lens == 0, "flow passes must not commit KV"
