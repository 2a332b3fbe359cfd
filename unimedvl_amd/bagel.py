"""Host-side unified model: the reference's ``Bagel`` inference interface
(codes/modeling/unimedvl/bagel.py:377-1392) over the MI355X engine.

Same method names, argument names and return conventions as the reference, so that
``InterleaveInferencer`` / ``chat``-style callers are drop-ins:
    prepare_prompts / forward_cache_update_text        bagel.py:377 / :412
    prepare_vit_images / forward_cache_update_vit      bagel.py:460 / :523
    prepare_vae_images / forward_cache_update_vae      bagel.py:617 / :697
    prepare_vae_latent / prepare_vae_latent_cfg        bagel.py:809 / :867
    generate_image / _forward_flow                     bagel.py:901 / :989
    prepare_start_tokens / generate_text / chat        bagel.py:1213 / :1236 / :1321
The prepare_* functions build the same packed index tensors on the host (they are the
boundary's data format); the forward_* functions consume what they need of them and
run everything else on the GPU through libunimedvl_hip.so.
"""
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch

from . import ops
from .config import UniMedVLConfig
from .data_utils import (PackedVitImages, get_flattened_position_ids_extrapolate, get_flattened_position_ids_interpolate, patchify)
from .decode import DecodeSession
from .prep import BagelPrep
from .kvcache import NaiveCache
from .llm import Qwen2MoT
from .vit import SiglipVisionModel
from .weights import GlueWeights, LLMWeights, ViTWeights

BF16 = torch.bfloat16


class Bagel(BagelPrep):
    def __init__(self, cfg: UniMedVLConfig, get, device="cuda", visual_gen=True, visual_und=True,
                 interpolate_pos=False):
        """`get(name)` returns reference-named tensors (see weights.py / shapes.py)."""
        if not torch.cuda.is_available():
            raise RuntimeError("unimedvl_amd needs an MI355X (ROCm) device; there is no CPU fallback")
        BagelPrep.__init__(self, cfg, interpolate_pos)
        self.device = torch.device(device)
        with ops.device_scope(self.device):     # weight packing kernels launch on the current device
            self.language_model = Qwen2MoT(cfg, LLMWeights(cfg, get, self.device, load_gen=visual_gen), self.device)
            self.glue = GlueWeights(cfg, get, self.device, visual_gen, visual_und)
            self.vit_model = SiglipVisionModel(cfg, ViTWeights(cfg, get, self.device), self.device) if visual_und else None
        self.vae_model = None
        self.hidden_size = cfg.hidden
        self.use_moe = True
        self.num_heads = cfg.heads
        self.latent_patch_size = cfg.latent_patch
        self.latent_downsample = cfg.latent_downsample
        self.max_latent_size = cfg.max_latent
        self.latent_channel = cfg.z_channels
        self.patch_latent_dim = cfg.latent_patch ** 2 * cfg.z_channels
        self.vit_patch_size = cfg.patch
        self.vit_max_num_patch_per_side = cfg.vit_side
        self.vit_hidden_size = cfg.vit_hidden
        self.get_flattened_position_ids = (get_flattened_position_ids_interpolate if interpolate_pos
                                           else get_flattened_position_ids_extrapolate)
        self.config = SimpleNamespace(
            llm_config=SimpleNamespace(num_hidden_layers=cfg.layers, hidden_size=cfg.hidden,
                                       num_attention_heads=cfg.heads, layer_module="Qwen2MoTDecoderLayer"),
            visual_gen=visual_gen, visual_und=visual_und, latent_patch_size=cfg.latent_patch,
            max_latent_size=cfg.max_latent, vit_max_num_patch_per_side=cfg.vit_side)
        self.decode_use_graph = True
        self.eos_check_every = 16
        import os
        self.prefill_graph = os.environ.get("UMV_PREFILL_GRAPH", "1") not in ("0", "")   # graph replay of image spans into reserved caches
        # prepare_vit_images hands the engine the transformed images and the patch tokens are made ON THE DEVICE (umv_patchify_f32_bf16);
        # generation_input["packed_vit_tokens"] is then a data_utils.PackedVitImages, which still is the reference's tensor for
        # anyone who asks (UMV_DEVICE_PATCHIFY=0: the reference's host-side permute, 4 ms per 448 x 448 image)
        self.device_patchify = os.environ.get("UMV_DEVICE_PATCHIFY", "1") not in ("0", "")
        self._vit_graphs = {}
        self.prefill_graph_max = 8          # captured image-span graphs kept (one per cache and patch grid; ~0.1 GB of activations each)
        self.chat_cache_tokens = 0          # > 0: chat() prefills / decodes in one pooled, reserved cache of that capacity
        self._chat_cache = None

    def eval(self):
        return self

    # ------------------------------------------------------------------ text
    @torch.no_grad()
    @ops.on_device
    def forward_cache_update_text(self, past_key_values: NaiveCache, packed_text_ids, packed_text_position_ids,
                                  text_token_lens, packed_text_indexes=None, packed_key_value_indexes=None,
                                  key_values_lens=None):
        emb = self.language_model.embed_tokens(packed_text_ids)
        out = self.language_model.forward_inference(
            packed_query_sequence=emb, query_lens=text_token_lens, packed_query_position_ids=packed_text_position_ids,
            packed_query_indexes=packed_text_indexes, past_key_values=past_key_values,
            packed_key_value_indexes=packed_key_value_indexes, key_values_lens=key_values_lens,
            update_past_key_values=True, is_causal=True, mode="und")
        return out.past_key_values

    # ------------------------------------------------------------------ ViT images
    @ops.on_device
    def encode_vit(self, packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens):
        """ViT tower + connector + vit_pos_embed (bagel.py:581-592); returns [N, hidden] bf16 (pre-scatter)."""
        lens = vit_token_seqlens.to("cpu")
        cu = torch.nn.functional.pad(torch.cumsum(lens, dim=0), (1, 0)).to(torch.int32)
        vit = self.vit_model(packed_pixel_values=packed_vit_tokens, packed_flattened_position_ids=packed_vit_position_ids,
                             cu_seqlens=cu, max_seqlen=int(lens.max()))
        h = ops.gemm(vit, self.glue.conn1, act="gelu_tanh")
        return ops.gemm(h, self.glue.conn2)

    # ---- image-span prefill from a HIP graph (profiles/HISTORY.md section 7.4)
    # One 448x448 image is ~460 kernel launches (ViT tower, connector, 28 LLM layers over 1026 tokens): 33 ms of GPU work but
    # ~55 ms of wall time when every launch goes through ctypes.  The shapes depend only on the image's patch grid, and all
    # per-request values (pixels, which cache segment / slot the tokens go to, rope position, keys visible) are read by the
    # kernels from device memory, so for a cache whose slabs stay put (NaiveCache.reserve) the whole span is captured once
    # per (cache, grid) and replayed: the request's pixels and a ~12 KB plan are uploaded, then one graph launch.
    def _vit_graph_key(self, cache, gi):
        if not self.prefill_graph or cache is None or not getattr(cache, "reserved", False) or cache.slabs is None:
            return None
        lens = [int(v) for v in gi["vit_token_seqlens"].tolist()]
        if len(lens) != 1:                       # one image per call (the scripts' and the batcher's admission path)
            return None
        qlens = [int(v) for v in gi["packed_seqlens"].tolist()]
        if max(c + q for c, q in zip(cache.lens, qlens)) > cache.cap:
            return None                          # would re-allocate: the eager path handles (and reports) that
        pos = gi["packed_vit_position_ids"]
        # the captured graph bakes in the K / V^T slab addresses of EVERY layer: all of them are part of the key, so a graph can
        # only be replayed on storage laid out exactly like the one it was captured on (a new cache that happens to reuse the
        # layer-0 address alone must not match - it would be written through stale addresses for the other layers)
        slabs = tuple(p for sl in cache.slabs for p in (sl.k.data_ptr(), sl.vt.data_ptr()))
        # ... and the input kind: a plan made for images (device patchify) cannot be refreshed from a patch tensor or the reverse
        kind = "images" if isinstance(gi["packed_vit_tokens"], PackedVitImages) else "tokens"
        return (slabs, cache.cap, len(cache.lens), lens[0], int(pos[0]), int(pos[-1]), int(gi["packed_text_ids"][0]),
                int(gi["packed_text_ids"][-1]), kind)

    def _vit_graph_run(self, key, cache, gi):
        dev, lm = self.device, self.language_model
        qlens = [int(v) for v in gi["packed_seqlens"].tolist()]
        nseg = len(cache.lens)
        if len(qlens) != nseg:
            raise ValueError(f"cache holds {nseg} samples, call has {len(qlens)}")
        cu = torch.nn.functional.pad(torch.cumsum(gi["vit_token_seqlens"].to("cpu"), dim=0), (1, 0)).to(torch.int32)
        g = self._vit_graphs.get(key)
        if g is None:
            g = SimpleNamespace()
            g.text_ids = gi["packed_text_ids"].to(device=dev, dtype=torch.int64)
            g.text_rows = gi["packed_text_indexes"].to(device=dev, dtype=torch.int32)
            g.vit_rows = gi["packed_vit_token_indexes"].to(device=dev, dtype=torch.int32)
            g.vplan = self.vit_model.make_plan(gi["packed_vit_tokens"], gi["packed_vit_position_ids"], cu, int(gi["vit_token_seqlens"].max()))
            g.lplan = lm.make_plan(qlens, gi["packed_position_ids"], cache.lens)
            g.lplan.max_kv = cache.cap
            T = sum(qlens)

            def body():
                seq = torch.zeros((T, self.hidden_size), dtype=BF16, device=dev)
                lm.embed_tokens(g.text_ids, out=seq, out_rows=g.text_rows)
                vit = self.vit_model(plan=g.vplan)
                conn = ops.gemm(ops.gemm(vit, self.glue.conn1, act="gelu_tanh"), self.glue.conn2)
                ops.add_rows(conn, seq, table=self.glue.vit_pos, idx=g.vplan["pos_ids"], out_rows=g.vit_rows)
                lm.forward_inference(packed_query_sequence=seq, query_lens=g.lplan.qlens, packed_query_position_ids=None,
                                     past_key_values=cache, update_past_key_values=False, is_causal=False, mode="und", plan=g.lplan)
            s = torch.cuda.Stream(device=dev)        # warm-up outside the capture (lazy module loading, allocator)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                body()
            torch.cuda.current_stream().wait_stream(s)
            g.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.graph):
                body()
            self._vit_graphs[key] = g
            while len(self._vit_graphs) > self.prefill_graph_max:      # arbitrary image sizes: keep the most recent grids only
                self._vit_graphs.pop(next(iter(self._vit_graphs)))
        else:
            self._vit_graphs[key] = self._vit_graphs.pop(key)          # most recently used last
            self.vit_model.make_plan(gi["packed_vit_tokens"], gi["packed_vit_position_ids"], cu, int(gi["vit_token_seqlens"].max()), into=g.vplan)
            lm.make_plan(qlens, gi["packed_position_ids"], cache.lens, into=g.lplan)
        g.graph.replay()
        cache.lens = [c + q for c, q in zip(cache.lens, qlens)]
        return cache

    @torch.no_grad()
    @ops.on_device
    def forward_cache_update_vit(self, past_key_values: NaiveCache, packed_text_ids, packed_text_indexes,
                                 packed_vit_tokens, packed_vit_token_indexes, packed_vit_position_ids, vit_token_seqlens,
                                 packed_position_ids, packed_seqlens, packed_indexes=None, packed_key_value_indexes=None,
                                 key_values_lens=None):
        dev = self.device
        if key_values_lens is not None and past_key_values is not None and past_key_values.slabs is not None and \
                [int(v) for v in key_values_lens.tolist()] != list(past_key_values.lens):
            raise ValueError(f"key_values_lens {key_values_lens.tolist()} disagree with the cache ({past_key_values.lens})")
        gi = dict(packed_text_ids=packed_text_ids, packed_text_indexes=packed_text_indexes, packed_vit_tokens=packed_vit_tokens,
                  packed_vit_token_indexes=packed_vit_token_indexes, packed_vit_position_ids=packed_vit_position_ids,
                  vit_token_seqlens=vit_token_seqlens, packed_position_ids=packed_position_ids, packed_seqlens=packed_seqlens)
        key = self._vit_graph_key(past_key_values, gi)
        if key is not None:
            return self._vit_graph_run(key, past_key_values, gi)
        T = int(packed_seqlens.sum())
        seq = torch.zeros((T, self.hidden_size), dtype=BF16, device=dev)
        self.language_model.embed_tokens(packed_text_ids, out=seq,
                                         out_rows=packed_text_indexes.to(device=dev, dtype=torch.int32))
        conn = self.encode_vit(packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens)
        ops.add_rows(conn, seq, table=self.glue.vit_pos, idx=packed_vit_position_ids.to(device=dev, dtype=torch.int64),
                     out_rows=packed_vit_token_indexes.to(device=dev, dtype=torch.int32))
        out = self.language_model.forward_inference(
            packed_query_sequence=seq, query_lens=packed_seqlens, packed_query_position_ids=packed_position_ids,
            packed_query_indexes=packed_indexes, past_key_values=past_key_values,
            packed_key_value_indexes=packed_key_value_indexes, key_values_lens=key_values_lens,
            update_past_key_values=True, is_causal=False, mode="und")
        return out.past_key_values

    # ------------------------------------------------------------------ VAE-encoded images (edit / reconstruction)
    @ops.on_device
    def time_embed(self, t_values):
        """TimestepEmbedder (modeling_utils.py:87-109) for a vector of timesteps -> [n, hidden] bf16: the 256-wide sinusoid
        (umv_timestep_embed; the 128 frequencies come from torch once, so t * freqs has the reference's bits), then the two
        linears with the SiLU in the first one's epilogue."""
        import math
        half = 128
        if getattr(self, "_t_freqs", None) is None:
            self._t_freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(self.device)
        t = torch.as_tensor(t_values, dtype=torch.float32).reshape(-1).to(self.device)
        emb = ops.timestep_embed(t, self._t_freqs)
        h = ops.gemm(emb, self.glue.time0, act="silu")
        return ops.gemm(h, self.glue.time2)

    @torch.no_grad()
    @ops.on_device
    def forward_cache_update_vae(self, vae_model, past_key_values: NaiveCache, padded_images, patchified_vae_latent_shapes,
                                 packed_vae_position_ids, packed_timesteps, packed_vae_token_indexes, packed_text_ids,
                                 packed_text_indexes, packed_position_ids, packed_seqlens, packed_indexes=None,
                                 key_values_lens=None, packed_key_value_indexes=None, noise=None):
        dev = self.device
        T = int(packed_seqlens.sum())
        seq = torch.zeros((T, self.hidden_size), dtype=BF16, device=dev)
        text_rows = packed_text_indexes.to(device=dev, dtype=torch.int32)
        vae_rows = packed_vae_token_indexes.to(device=dev, dtype=torch.int32)
        self.language_model.embed_tokens(packed_text_ids, out=seq, out_rows=text_rows)
        packed_latent = vae_model.encode_packed(padded_images, patchified_vae_latent_shapes, self.latent_patch_size,
                                                noise=noise)                      # [sum h*w, p*p*c] bf16
        t_emb = self.time_embed(packed_timesteps)                               # [1, hidden]
        x = ops.gemm(packed_latent, self.glue.vae2llm)
        ops.add_rows(x, seq, bcast=t_emb[0], table=self.glue.latent_pos,
                     idx=packed_vae_position_ids.to(device=dev, dtype=torch.int64), out_rows=vae_rows)
        out = self.language_model.forward_inference(
            packed_query_sequence=seq, query_lens=packed_seqlens, packed_query_position_ids=packed_position_ids,
            packed_query_indexes=packed_indexes, past_key_values=past_key_values, key_values_lens=key_values_lens,
            packed_key_value_indexes=packed_key_value_indexes, update_past_key_values=True, is_causal=False, mode="gen",
            packed_vae_token_indexes=packed_vae_token_indexes, packed_text_indexes=packed_text_indexes)
        return out.past_key_values

    # ------------------------------------------------------------------ image generation
    @torch.no_grad()
    @ops.on_device
    def generate_image(self, packed_text_ids, packed_text_indexes, packed_init_noises, packed_vae_position_ids,
                       packed_vae_token_indexes, packed_seqlens, packed_position_ids, packed_indexes=None,
                       past_key_values: NaiveCache = None, key_values_lens=None, packed_key_value_indexes=None,
                       num_timesteps: int = 24, timestep_shift: float = 1.0, cfg_renorm_min: float = 0.0,
                       cfg_renorm_type: str = "global", cfg_interval: Optional[Tuple[float, float]] = (0, 1),
                       cfg_text_scale: float = 1.0, cfg_text_packed_query_indexes=None,
                       cfg_text_packed_position_ids=None, cfg_text_past_key_values: Optional[NaiveCache] = None,
                       cfg_text_key_values_lens=None, cfg_text_packed_key_value_indexes=None,
                       cfg_img_scale: float = 1.0, cfg_img_packed_query_indexes=None, cfg_img_packed_position_ids=None,
                       cfg_img_past_key_values: Optional[NaiveCache] = None, cfg_img_key_values_lens=None,
                       cfg_img_packed_key_value_indexes=None, cfg_type: str = "parallel", callback=None,
                       cfg_renorm_batch_semantics: str = "per_sample"):
        """Rectified-flow Euler sampler with dual CFG + renorm (bagel.py:901-986, 989-1211).

        cfg_renorm_batch_semantics (only matters for cfg_renorm_type="global" with more than one sample in the packed batch):
        "per_sample" (default) takes the renorm norms per sample, i.e. a packed call equals B single-sample calls - the only
        way the reference's inferencer calls this method (inferencer.py:165-232) and what keeps a request's image independent
        of how a batch is sharded over GPUs; "reference" reproduces bagel.py:1196-1198 literally: ONE norm over all the
        latent tokens of the packed batch, which couples the samples."""
        sess = FlowSession(self, locals())
        while not sess.finished:
            sess.step(1)
            if callback is not None:
                callback(sess.i - 1, sess.x_t)
        return sess.latents()

    @staticmethod
    def _contexts_identical(ca, pos_a, cb, pos_b):
        """True only if two KV contexts hold bit-identical keys/values for every layer and the query
        position ids agree (then a forward pass over either gives the same result)."""
        if ca is None or cb is None or pos_a is None or pos_b is None:
            return False
        if ca is cb:
            return torch.equal(pos_a.cpu(), pos_b.cpu())
        if list(ca.lens) == list(cb.lens) and ca.shares_storage_with(cb):     # one is a snapshot of the other: no compare, no sync
            return torch.equal(pos_a.cpu().to(torch.long), pos_b.cpu().to(torch.long))
        if ca.slabs is None or cb.slabs is None or list(ca.lens) != list(cb.lens):
            return False
        if not torch.equal(pos_a.cpu().to(torch.long), pos_b.cpu().to(torch.long)):
            return False
        n = max(ca.lens)
        if n == 0:
            return True
        if min(ca.lens) != n:      # ragged batch: slots past a short sample's length may hold scratch; do not guess
            return False
        diff = torch.zeros((), dtype=torch.int64, device=ca.slabs[0].k.device)
        for sa, sb in zip(ca.slabs, cb.slabs):      # one host sync at the end, not one per layer
            diff += (sa.k[:, :, :n] != sb.k[:, :, :n]).sum() + (sa.vt[:, :, :, :n] != sb.vt[:, :, :, :n]).sum()
        return int(diff.item()) == 0

    # ------------------------------------------------------------------ text generation
    def _sampling_seed(self):
        """Next 62-bit key for the device-side sampler: one draw from the DEVICE's default generator - the generator the
        reference's multinomial consumes (bagel.py:1297-1299 samples on the GPU).  So torch.manual_seed(s) - which reseeds it -
        restarts the stream, also with the same s again (two identical calls each preceded by manual_seed(42) sample identical
        text), successive calls draw new keys, and torch's CPU generator is left untouched (the init noise
        prepare_vae_latent draws afterwards is what it would have been)."""
        return int(torch.randint(0, 2 ** 62, (1,), device=self.device).item())

    @torch.no_grad()
    @ops.on_device
    def generate_text(self, past_key_values: NaiveCache, packed_key_value_indexes=None, key_values_lens=None,
                      packed_start_tokens=None, packed_query_position_ids=None, max_length: int = 0,
                      do_sample: bool = False, temperature: float = 1.0, end_token_id: int = None,
                      return_logits: bool = False, per_sample_eos: bool = False):
        """Greedy decode (bagel.py:1236-1317).  Returns [steps, B] int64 whose row 0 holds the
        start tokens.  Like the reference, the batch stops when SAMPLE 0 emits end_token_id
        (bagel.py:1313); per_sample_eos=True is the batched extension (stops when every sample
        has emitted it; rows after a sample's EOS keep decoding and should be ignored)."""
        # do_sample: softmax(logits / temperature) + multinomial on the device (bagel.py:1297-1299).  The
        # draw is keyed by a seed derived from torch.initial_seed(), so torch.manual_seed(s) makes runs
        # reproducible; the stream itself is not torch's (no device can reproduce another's RNG).
        seed = self._sampling_seed() if do_sample else 0
        if key_values_lens is not None and [int(v) for v in key_values_lens.tolist()] != list(past_key_values.lens):
            raise ValueError("key_values_lens disagree with the cache")
        if max_length <= 0:   # nothing to decode (the reference would fail on torch.stack([]) here, bagel.py:1316)
            out = torch.zeros((0, len(past_key_values.lens)), dtype=torch.int64, device=self.device)
            return (out, torch.zeros((0, len(past_key_values.lens), self.cfg.vocab), dtype=BF16, device=self.device)) if return_logits else out
        sess = DecodeSession(self.language_model, past_key_values, packed_start_tokens, packed_query_position_ids,
                             max_length, use_graph=self.decode_use_graph and not return_logits,
                             do_sample=do_sample, temperature=temperature, seed=seed)
        logits = []
        steps = 0
        stop = None
        while steps < max_length:
            n = 1 if return_logits else min(self.eos_check_every, max_length - steps)
            sess.step(n)
            if return_logits:
                logits.append(sess.logits.clone())
            steps += n
            if end_token_id is not None:
                pred = sess.pred_ids[:steps].cpu()
                hit = pred == end_token_id
                if per_sample_eos:
                    if bool(hit.any(0).all()):
                        stop = int(hit.float().argmax(0).max()) + 1
                        break
                elif bool(hit[:, 0].any()):
                    stop = int(hit[:, 0].float().argmax()) + 1
                    break
        rows = stop if stop is not None else steps
        sess.commit(rows)
        out = sess.in_ids[:rows].clone()
        if return_logits:
            return out, torch.stack(logits[:rows], 0)
        return out

    # ------------------------------------------------------------------ convenience (evaluation path)
    @torch.no_grad()
    @ops.on_device
    def chat(self, tokenizer, new_token_ids, image_transform, images, prompt, max_length: int,
             do_sample: bool = False, temperature: float = 1.0):
        """ViT-only VQA convenience path (bagel.py:1321-1392)."""
        if self.chat_cache_tokens > 0:       # serving: one reserved cache reused by every request (stable slabs -> graph prefill)
            if self._chat_cache is None or self._chat_cache.cap < self.chat_cache_tokens:
                self._chat_cache = NaiveCache(self.cfg.layers)
                self._chat_cache.reserve(1, self.chat_cache_tokens, self.cfg.kv_heads, self.cfg.head_dim, self.device)
            cache = self._chat_cache
            cache.lens = [0]
        else:
            cache = NaiveCache(self.cfg.layers)
        newlens, new_rope = [0], [0]
        for image in images:
            gi, newlens, new_rope = self.prepare_vit_images(newlens, new_rope, [image], image_transform, new_token_ids)
            cache = self.forward_cache_update_vit(cache, **gi)
        gi, newlens, new_rope = self.prepare_prompts(newlens, new_rope, [prompt], tokenizer, new_token_ids)
        cache = self.forward_cache_update_text(cache, **gi)
        gi = self.prepare_start_tokens(newlens, new_rope, new_token_ids)
        ids = self.generate_text(past_key_values=cache, max_length=max_length, do_sample=do_sample,
                                 temperature=temperature, end_token_id=new_token_ids["eos_token_id"], **gi)
        output = tokenizer.decode(ids[:, 0].cpu())
        return output.split("<|im_end|>")[0].split("<|im_start|>")[1]


class FlowSession:
    """The rectified-flow sampler of Bagel.generate_image (bagel.py:901-986, 989-1211) as a RESUMABLE object: the setup of
    generate_image, then `step(n)` = n Euler steps.  generate_image runs it to the end; the mixed VQA + T2I scheduler
    (serving.MixedBatcher, BASELINE.json configs[4]) interleaves its steps with the decode steps of the VQA slots."""

    def __init__(self, model, a):
        m = self.m = model
        packed_text_ids, packed_text_indexes = a["packed_text_ids"], a["packed_text_indexes"]
        packed_init_noises, packed_vae_position_ids = a["packed_init_noises"], a["packed_vae_position_ids"]
        packed_vae_token_indexes, packed_seqlens, packed_position_ids = a["packed_vae_token_indexes"], a["packed_seqlens"], a["packed_position_ids"]
        past_key_values, key_values_lens = a["past_key_values"], a["key_values_lens"]
        num_timesteps, timestep_shift = a["num_timesteps"], a["timestep_shift"]
        cfg_renorm_type, cfg_text_scale, cfg_img_scale = a["cfg_renorm_type"], a["cfg_text_scale"], a["cfg_img_scale"]
        cfg_text_past_key_values, cfg_text_packed_position_ids = a["cfg_text_past_key_values"], a["cfg_text_packed_position_ids"]
        cfg_img_past_key_values, cfg_img_packed_position_ids = a["cfg_img_past_key_values"], a["cfg_img_packed_position_ids"]
        self.cfg_interval, self.cfg_renorm_min = a["cfg_interval"], a["cfg_renorm_min"]
        self.cfg_text_scale, self.cfg_img_scale = cfg_text_scale, cfg_img_scale
        batch_sem = a.get("cfg_renorm_batch_semantics", "per_sample")
        if batch_sem not in ("per_sample", "reference"):
            raise ValueError(f"cfg_renorm_batch_semantics must be 'per_sample' or 'reference', got {batch_sem!r}")
        self._setup(model, packed_text_ids, packed_text_indexes, packed_init_noises, packed_vae_position_ids,
                           packed_vae_token_indexes, packed_seqlens, packed_position_ids, past_key_values, key_values_lens,
                           num_timesteps, timestep_shift, cfg_renorm_type, cfg_text_scale, cfg_text_past_key_values,
                           cfg_text_packed_position_ids, cfg_img_scale, cfg_img_past_key_values, cfg_img_packed_position_ids, batch_sem)

    def _setup(self, m, packed_text_ids, packed_text_indexes, packed_init_noises, packed_vae_position_ids,
               packed_vae_token_indexes, packed_seqlens, packed_position_ids, past_key_values, key_values_lens,
               num_timesteps, timestep_shift, cfg_renorm_type, cfg_text_scale, cfg_text_past_key_values,
               cfg_text_packed_position_ids, cfg_img_scale, cfg_img_past_key_values, cfg_img_packed_position_ids, batch_sem):
        s = self     # (m = the Bagel model; the body below is generate_image's former setup)
        if cfg_renorm_type not in ("global", "channel", "text_channel"):
            raise NotImplementedError(f"{cfg_renorm_type} is not suppoprted")
        dev = m.device
        lm, g = m.language_model, m.glue
        x_t = packed_init_noises.to(device=dev, dtype=torch.float32).contiguous().clone()
        N, D = x_t.shape
        seqlens = [int(v) for v in packed_seqlens.tolist()]
        T = sum(seqlens)
        # schedule on the host in fp32, exactly as the reference builds it (bagel.py:937-940)
        ts = torch.linspace(1, 0, num_timesteps)
        ts = timestep_shift * ts / (1 + (timestep_shift - 1) * ts)
        dts = ts[:-1] - ts[1:]
        ts = ts[:-1]
        t_emb_all = m.time_embed(ts)                           # rows identical within a step: compute once
        text_rows = packed_text_indexes.to(device=dev, dtype=torch.int32)
        vae_rows = packed_vae_token_indexes.to(device=dev, dtype=torch.int32)
        vae_pos = packed_vae_position_ids.to(device=dev, dtype=torch.int64)
        seg_off = [0]
        for n in seqlens:
            seg_off.append(seg_off[-1] + n - 2)
        seg_off_d = torch.tensor(seg_off, dtype=torch.int32).to(dev)
        # The reference runs the conditional, no-text and no-image passes one after another
        # (bagel.py:1120-1171).  They share the query tokens and the weights and differ only in their KV
        # context, and samples are independent, so here they are ONE packed forward over nctx*B segments
        # of a merged cache: the weights stream once instead of three times and the GEMMs see 3x the rows.
        B = len(seqlens)
        use_text = cfg_text_scale > 1.0
        use_img = use_text and cfg_img_scale > 1.0   # the reference computes the image pass but drops it when
        ctxs = [(past_key_values, packed_position_ids)]                      # cfg_text_scale <= 1 (bagel.py:1173,1208)
        if use_text:
            ctxs.append((cfg_text_past_key_values, cfg_text_packed_position_ids))
        # Pure text-to-image: the "no image" context holds exactly the tokens of the conditional one
        # (inferencer.py:587,602), so its velocity equals v_t bit for bit (same rows through the same
        # deterministic kernels).  When the two caches and position ids are PROVABLY identical the third pass
        # is skipped and v_img := v_t; the CFG arithmetic is unchanged.
        img_same = use_img and m._contexts_identical(past_key_values, packed_position_ids,
                                                        cfg_img_past_key_values, cfg_img_packed_position_ids)
        if use_img and not img_same:
            ctxs.append((cfg_img_past_key_values, cfg_img_packed_position_ids))
        nctx = len(ctxs)
        cfgm = m.cfg
        for c, _ in ctxs:
            if c is None:
                raise ValueError("classifier-free guidance needs the cfg_* contexts")
        if key_values_lens is not None and past_key_values.slabs is not None and \
                [int(v) for v in key_values_lens.tolist()] != list(past_key_values.lens):
            raise ValueError("key_values_lens disagree with the cache")
        if nctx > 1:
            merged = NaiveCache.merged([c for c, _ in ctxs], [B] * nctx, max(seqlens), cfgm.kv_heads, cfgm.head_dim, dev)
            base = merged.view_segments(0, B)
        else:
            merged = base = past_key_values
        seq_all = torch.zeros((nctx * T, m.hidden_size), dtype=BF16, device=dev)
        rows_text = torch.cat([text_rows + c * T for c in range(nctx)])
        rows_vae = torch.cat([vae_rows + c * T for c in range(nctx)])
        lm.embed_tokens(packed_text_ids.repeat(nctx), out=seq_all, out_rows=rows_text)
        pos_all = torch.cat([p.to(torch.long).cpu() for _, p in ctxs])
        seqlens_all = torch.tensor(seqlens * nctx, dtype=torch.int)
        seq1 = seq_all[:T]

        def forward(n, cache):
            out = lm.forward_inference(
                packed_query_sequence=seq_all[:n * T], query_lens=seqlens_all[:n * B],
                packed_query_position_ids=pos_all[:n * T], past_key_values=cache, update_past_key_values=False,
                is_causal=False, mode="gen", packed_vae_token_indexes=rows_vae[:n * N], packed_text_indexes=rows_text[:n * 2 * B])
            return ops.gemm(out.packed_query_sequence, g.llm2vae)          # [n*T, D]; vae rows picked by the CFG kernel

        s.rtype = {"global": 0, "channel": 1, "text_channel": 2}[cfg_renorm_type]
        if batch_sem == "reference" and s.rtype == 0 and B > 1:
            # bagel.py:1196-1198: ONE norm over every latent token of the packed batch
            seg_off_d = torch.tensor([0, N], dtype=torch.int32).to(dev)
            s.renorm_segments = 1
        else:
            s.renorm_segments = B
        s.x_t, s.D, s.T, s.N, s.B, s.nctx = x_t, D, T, N, B, nctx
        s.ts, s.dts, s.t_emb_all = ts, dts, t_emb_all
        s.use_text, s.use_img, s.img_same = use_text, use_img, img_same
        s.seq_all, s.vae_pos, s.vae_rows, s.seg_off_d = seq_all, vae_pos, vae_rows, seg_off_d
        s.merged, s.base, s.forward, s.seqlens = merged, base, forward, seqlens
        s.i = 0
        return s

    @property
    def finished(self):
        return self.i >= len(self.ts)

    @property
    def steps_left(self):
        return len(self.ts) - self.i

    @torch.no_grad()
    def step(self, n=1):
        """n Euler steps (fewer if the schedule ends)."""
        s, m = self, self.m
        g = m.glue
        with ops.device_scope(m.device):
            for _ in range(n):
                if s.finished:
                    return
                i = s.i
                t = float(s.ts[i])
                guided = s.use_text and t > s.cfg_interval[0] and t <= s.cfg_interval[1]
                s_text, s_img = (s.cfg_text_scale, s.cfg_img_scale) if guided else (1.0, 1.0)
                xb = ops.cast_pad(s.x_t, s.D)
                h = ops.gemm(xb, g.vae2llm)
                T = s.T
                for c in range(s.nctx if guided else 1):
                    ops.add_rows(h, s.seq_all[c * T:(c + 1) * T], bcast=s.t_emb_all[i], table=g.latent_pos, idx=s.vae_pos, out_rows=s.vae_rows)
                if guided:
                    v = s.forward(s.nctx, s.merged)
                    v_t, v_text = v[:T], v[T:2 * T]
                    v_img = (v_t if s.img_same else v[2 * T:3 * T]) if s.use_img else None
                else:
                    v_t, v_text, v_img = s.forward(1, s.base), None, None
                ops.cfg_renorm_euler(s.x_t, v_t, v_text, v_img, s.vae_rows, s.seg_off_d, s.renorm_segments, s_text,
                                     s_img if s.use_img else 1.0, s.cfg_renorm_min, s.rtype, float(s.dts[i]))
                s.i += 1

    def latents(self):
        return self.x_t.split([n - 2 for n in self.seqlens])
