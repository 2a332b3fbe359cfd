"""Continuous (in-flight) batching for VQA decode - SURVEY.md section 8f rank 4 ("serving-grade decode").

The reference decodes one request at a time and stops a whole batch when SAMPLE 0 emits EOS
(codes/modeling/unimedvl/bagel.py:1262-1314).  Here B cache segments ("slots") decode together in ONE captured HIP
graph (decode.DecodeSession); every `check_every` steps the host looks at the generated ids, retires the slots that
produced <|im_end|> (or hit their token budget) and prefills the next queued request straight into the freed slot:

  * the KV cache is one slab per layer with a per-slot capacity reserved up front (kvcache.NaiveCache.reserve), so a new
    request is a prefill on a ONE-SEGMENT VIEW of that slab (NaiveCache.view_segments) - no copy, no re-allocation, the other
    slots' keys are untouched;
  * a request whose context does not fit that reservation GROWS the slabs (growable=True, the default: the reference's
    NaiveCache grows without bound, qwen2_navit.py:585-600): between two decode rounds the cache doubles
    (NaiveCache.ensure copies the committed keys), the decode session is re-captured on the new slabs and every active
    slot continues from its next token - answers are unchanged (tests/test_serving_gpu.py).  growable=False keeps the
    old contract: such a request is refused (ValueError) and nothing is re-allocated;
  * the graph reads each slot's next token / cache length / rope position from device memory
    (DecodeSession.set_slot), so re-pointing a slot needs no re-capture;
  * samples are independent rows of every kernel: a request's tokens are the same whichever slot and neighbours it gets
    (tests/test_serving_gpu.py checks this against one-request-at-a-time greedy decoding).
"""
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Deque, Dict, List, Optional

import torch

from . import ops

from .decode import DecodeSession
from .kvcache import NaiveCache, PagedCache


@dataclass
class _Request:
    rid: int
    images: List[Any]
    prompt: str
    max_new_tokens: int
    tokens: List[int] = field(default_factory=list)


class ContinuousBatcher:
    def __init__(self, model, tokenizer, new_token_ids, image_transform, slots: int = 8, max_context: int = 2048,
                 max_new_tokens: int = 256, check_every: int = 16, do_sample: bool = False, temperature: float = 1.0,
                 use_graph: bool = True, growable: bool = True, context_limit: Optional[int] = None, paged: bool = False,
                 pool_pages: Optional[int] = None):
        """max_context: the context (prompt + images) the slots are RESERVED for; with growable=True longer requests enlarge
        the cache (up to context_limit tokens of context when given), with growable=False they are refused.
        paged=True: block-table KV (kvcache.PagedCache) instead of slabs - the slots draw 256-token pages from ONE pool of `pool_pages`
        pages per layer (default: what max_context + max_new_tokens needs for every slot, plus as much again), a request of any length
        up to context_limit (default 32 768) is admitted without re-allocating or re-capturing anything, and a finished request returns
        its pages.  Same answers as the slab cache (tests/test_paged_kv_gpu.py)."""
        self.model, self.tokenizer, self.new_token_ids, self.image_transform = model, tokenizer, new_token_ids, image_transform
        self.device = model.device
        self.slots, self.check_every = int(slots), int(check_every)
        self.max_context, self.default_new = int(max_context), int(max_new_tokens)
        self.do_sample, self.temperature, self.use_graph = do_sample, temperature, use_graph
        self.growable, self.context_limit = bool(growable), context_limit
        self._grew = False
        cfg = model.cfg
        self.paged = bool(paged)
        per_slot = self.max_context + self.default_new + self.check_every + 8
        if self.paged:
            pages = 2 * self.slots * ((per_slot + ops.KV_PAGE - 1) // ops.KV_PAGE) + 1 if pool_pages is None else int(pool_pages)
            self.cache = PagedCache(cfg.layers, pool_pages=pages, max_context=(context_limit or 32768) + self.default_new + self.check_every + 8)
        else:
            self.cache = NaiveCache(cfg.layers)
        self.cache.reserve(self.slots, per_slot, cfg.kv_heads, cfg.head_dim, model.device)
        self.queue: Deque[_Request] = deque()
        self.results: Dict[int, str] = {}
        self._next_id = 0
        self.stats = {"decode_steps": 0, "prefills": 0, "tokens": 0, "cache_grows": 0}

    # ------------------------------------------------------------------ API
    def submit(self, images, prompt: str, max_new_tokens: Optional[int] = None) -> int:
        """images: one image / a list of images (PIL or [3,H,W] tensors, as the transform accepts) or None."""
        imgs = [] if images is None else (list(images) if isinstance(images, (list, tuple)) else [images])
        budget = self.default_new if max_new_tokens is None else int(max_new_tokens)
        if budget > self.default_new and not self.growable:
            raise ValueError(f"max_new_tokens {budget} exceeds the batcher's reserve of {self.default_new}")
        rid = self._next_id
        self._next_id += 1
        self.queue.append(_Request(rid, imgs, prompt, budget))
        return rid

    @torch.no_grad()
    @ops.on_device
    def run(self) -> Dict[int, str]:
        """Serve everything that has been submitted; returns {request id: answer}."""
        m, cache, B = self.model, self.cache, self.slots
        bos, eos = self.new_token_ids["bos_token_id"], self.new_token_ids["eos_token_id"]
        active: List[Optional[_Request]] = [None] * B
        state = [(bos, 0, 0)] * B                       # (next token, kv_len, rope position) per slot
        for b in range(B):
            self._free_slot(b)
        first = []
        for b in range(B):
            if self.queue:
                active[b] = self.queue.popleft()
                first.append(b)
        for b, st in zip(first, self._prefill_many(first, [active[b] for b in first])):
            state[b] = st
        seed = m._sampling_seed() if self.do_sample else 0

        def new_session():        # (re-)capture the decode step on the cache's current slabs; every slot continues from `state`
            start = torch.tensor([s[0] for s in state], dtype=torch.int64)
            pos = torch.tensor([s[2] for s in state], dtype=torch.int64)
            lens_now = list(cache.lens)
            sn = DecodeSession(m.language_model, cache, start, pos, self.check_every, use_graph=self.use_graph,
                               do_sample=self.do_sample, temperature=self.temperature, seed=seed)
            cache.lens = lens_now                       # the session only reads them; this loop owns the bookkeeping
            self._grew = False
            return sn
        sess = new_session()
        while any(r is not None for r in active) or self._other_work():
            if not any(r is not None for r in active):  # only the other kind of work is left (MixedBatcher: flow passes)
                self._between_rounds()
                continue
            k = self.check_every
            if self.paged:      # pages for this round's tokens (the captured step reads the table from device memory; an idle slot, parked
                cfg = m.cfg     # at the start of its segment, keeps one page of its own to write to)
                cache.ensure_tokens([n + k + 1 for n in cache.lens], cfg.kv_heads, cfg.head_dim, self.device)
            sess.rewind_outputs()
            sess.step(k)
            self._between_rounds()                      # (queued behind the decode steps; the host reads the ids after both)
            self.stats["decode_steps"] += k
            ids = sess.pred_ids[:k].cpu()               # [k, B]; the only host sync of the round
            freed = []
            for b in range(B):
                req = active[b]
                if req is None:
                    sess.set_slot(b, bos, 0, 0)         # idle slot: keep it parked at the start of its segment
                    state[b] = (bos, 0, 0)
                    continue
                col = ids[:, b].tolist()
                done = False
                for t in col:
                    if t == eos or len(req.tokens) >= req.max_new_tokens:
                        done = True
                        break
                    req.tokens.append(int(t))
                if len(req.tokens) >= req.max_new_tokens:
                    done = True
                if not done:
                    cache.lens[b] += k                  # the graph advanced this slot's device counters by k as well
                    state[b] = (col[-1], cache.lens[b], state[b][2] + k)
                    continue
                self._finish(req)
                active[b] = None
                self._free_slot(b)
                if self.queue:
                    active[b] = self.queue.popleft()
                    freed.append(b)
                else:
                    sess.set_slot(b, bos, 0, 0)
                    state[b] = (bos, 0, 0)
            # every slot that was freed this round is refilled by ONE batched prefill (images of the new requests share the
            # ViT and LLM forward; the other slots take part with zero query tokens)
            for b, (tok, kvl, rope) in zip(freed, self._prefill_many(freed, [active[b] for b in freed])):
                sess.set_slot(b, tok, kvl, rope)
                state[b] = (tok, kvl, rope)
            if self._grew:                              # an admitted request enlarged the cache: new slabs, new captured step
                sess = new_session()
        return dict(self.results)

    # ------------------------------------------------------------------ hooks (MixedBatcher)
    def _other_work(self) -> bool:
        return False

    def _free_slot(self, b: int):
        """slot b's request is done: forget its context (a paged cache gets its pages back)"""
        if self.paged:
            self.cache.release(b)
        else:
            self.cache.lens[b] = 0

    def _between_rounds(self):
        pass

    # ------------------------------------------------------------------ internals
    def _prefill(self, b: int, req: _Request):
        """Image(s) then the prompt into slot b (Bagel.chat's order, bagel.py:1321-1392); returns the slot's decode state."""
        m = self.model
        self._free_slot(b)
        view = self.cache.view_segments(b, b + 1)
        view.lens = [0]
        cap = view.cap
        kvl, rope = [0], [0]

        def room(n_ctx):          # growing re-allocates the pooled cache: commit what the view holds, take the view again
            nonlocal view, cap
            self.cache.lens[b] = view.lens[0]
            new_cap = self._check_room(n_ctx, req, cap)
            if new_cap != cap:
                keep = view.lens[0]
                view = self.cache.view_segments(b, b + 1)
                view.lens = [keep]
                cap = view.cap
        for image in req.images:
            gi, kvl, rope = m.prepare_vit_images(kvl, rope, [image], self.image_transform, self.new_token_ids)
            room(kvl[0])
            view = m.forward_cache_update_vit(view, **gi)
        gi, kvl, rope = m.prepare_prompts(kvl, rope, [req.prompt], self.tokenizer, self.new_token_ids)
        room(kvl[0])
        view = m.forward_cache_update_text(view, **gi)
        if view.cap != cap:
            raise RuntimeError("slot view was re-allocated: the request does not fit the reserved capacity")
        self.cache.lens[b] = view.lens[0]
        self.stats["prefills"] += 1
        return (self.new_token_ids["bos_token_id"], view.lens[0], rope[0])

    def _prefill_many(self, slots: List[int], reqs: List[_Request]):
        """Prefill several newly admitted requests together (slots ascending): image j of every request goes through one
        ViT + one packed LLM forward over the WHOLE slot cache, in which the other slots are segments with zero query
        tokens (their keys are not touched: the kernels address the cache by per-token segment / slot indices and skip
        segments without queries).  Falls back to one prefill per request when the requests' image counts differ."""
        if not slots:
            return []
        if len({len(r.images) for r in reqs}) != 1:
            return [self._prefill(b, r) for b, r in zip(slots, reqs)]
        # (a single admission takes this path too: on the whole reserved cache its image span replays from a HIP graph,
        # bagel.Bagel.forward_cache_update_vit)
        m, cache, B = self.model, self.cache, self.slots
        cap = cache.cap
        for b in slots:
            self._free_slot(b)
        k = len(slots)
        kvl, rope = [0] * k, [0] * k

        def spread(per_request):          # [k] query lengths -> [B] with zeros for the slots that do not take part
            full = torch.zeros(B, dtype=per_request.dtype)
            full[torch.tensor(slots)] = per_request.cpu()
            return full
        for j in range(len(reqs[0].images)):
            gi, kvl, rope = m.prepare_vit_images(kvl, rope, [r.images[j] for r in reqs], self.image_transform, self.new_token_ids)
            for n, r in zip(kvl, reqs):
                cap = self._check_room(n, r, cap)
            gi["packed_seqlens"] = spread(gi["packed_seqlens"])
            gi["key_values_lens"] = gi["packed_key_value_indexes"] = gi["packed_indexes"] = None   # written for a k-sample cache
            m.forward_cache_update_vit(cache, **gi)
        gi, kvl, rope = m.prepare_prompts(kvl, rope, [r.prompt for r in reqs], self.tokenizer, self.new_token_ids)
        for n, r in zip(kvl, reqs):
            cap = self._check_room(n, r, cap)
        gi["text_token_lens"] = spread(gi["text_token_lens"])
        gi["key_values_lens"] = gi["packed_key_value_indexes"] = gi["packed_text_indexes"] = None
        m.forward_cache_update_text(cache, **gi)
        if cache.cap != cap:
            raise RuntimeError("the slot cache was re-allocated: a request does not fit the reserved capacity")
        for b, n in zip(slots, kvl):
            assert cache.lens[b] == n
        self.stats["prefills"] += k
        self.stats["batched_prefills"] = self.stats.get("batched_prefills", 0) + 1
        return [(self.new_token_ids["bos_token_id"], n, r) for n, r in zip(kvl, rope)]

    def _check_room(self, ctx_tokens: int, req: _Request, cap: int) -> int:
        """Room for `ctx_tokens` of context + the request's new tokens in a slot?  Returns the capacity to go on with: the
        same, or - growable - the enlarged one (the pooled cache is re-allocated HERE, before the forward that would write
        past the old slabs; run() re-captures the decode session afterwards)."""
        need = ctx_tokens + req.max_new_tokens + self.check_every + 1
        if self.paged:          # pages are taken as the context grows; the only bound is the page table's reach
            if need > cap:
                raise ValueError(f"request {req.rid}: {need} tokens of context + answer exceed the page table's reach of {cap}")
            return cap
        if ctx_tokens <= self.max_context and need <= cap:
            return cap
        if not self.growable:
            raise ValueError(f"request {req.rid}: context of {ctx_tokens} tokens exceeds max_context={self.max_context}")
        if self.context_limit is not None and ctx_tokens > self.context_limit:
            raise ValueError(f"request {req.rid}: context of {ctx_tokens} tokens exceeds context_limit={self.context_limit}")
        if need > cap:
            cfg = self.model.cfg
            lens = list(self.cache.lens)
            self.cache.reserved = False
            self.cache.ensure(self.slots, need, cfg.kv_heads, cfg.head_dim, self.model.device)     # >= 2 x the old capacity
            self.cache.reserved = True
            self.cache.lens = lens
            self._grew = True
            self.stats["cache_grows"] += 1
        return self.cache.cap

    def _finish(self, req: _Request):
        bos = self.new_token_ids["bos_token_id"]
        text = self.tokenizer.decode(torch.tensor([bos] + req.tokens, dtype=torch.int64))
        self.results[req.rid] = text.split("<|im_end|>")[0].split("<|im_start|>")[1]   # inferencer.py:277-278
        self.stats["tokens"] += len(req.tokens)


@dataclass
class _T2IRequest:
    rid: int
    prompt: str
    image_shape: Any
    params: tuple              # (num_timesteps, timestep_shift, cfg_text_scale, cfg_img_scale, cfg_interval, renorm_min, renorm_type)
    noise: Optional[torch.Tensor] = None


class MixedBatcher(ContinuousBatcher):
    """VQA decode slots AND text-to-image jobs in one step stream - BASELINE.json configs[4] ("mixed VQA + T2I interleaved
    batch"; with cfg.llm_weight_dtype = "fp8" the decode steps stream the e4m3 images and, with llm_act_dtype = "fp8", the
    flow passes run on the fp8 matrix instruction).

    The reference interleaves understanding and generation inside ONE request's context loop (inferencer.py:552-637: text
    and image items update the context in order, then either gen_text or gen_image runs) and serves one request at a time.
    Here requests of both kinds are in flight together: a round of the serving loop is `check_every` captured decode steps
    for all VQA slots (Bagel.generate_text, bagel.py:1236-1317) followed by `flow_steps_per_round` Euler steps of the active
    text-to-image group (Bagel.generate_image, bagel.py:901-1211: each guided step is one packed forward over the
    conditional / no-text / no-image contexts of every image of the group), all queued on the same stream before the host
    looks at the new token ids.  Retired VQA slots are refilled in flight as in ContinuousBatcher; a finished image group is
    decoded by the VAE in one batch (inferencer.py:234-256) and the next group of queued prompts is admitted.

    Samples are independent rows of every kernel and the two kinds of work share nothing but the weights, so every answer
    and every image equals what the request gets when it is served alone (tests/test_serving_gpu.py, test_fullwidth_gpu.py)."""

    def __init__(self, model, vae_model, tokenizer, new_token_ids, image_transform, slots: int = 8, t2i_batch: int = 4,
                 flow_steps_per_round: int = 2, **kw):
        super().__init__(model, tokenizer, new_token_ids, image_transform, slots=slots, **kw)
        self.vae = vae_model
        self.t2i_batch, self.flow_steps_per_round = int(t2i_batch), int(flow_steps_per_round)
        self.t2i_queue: Deque[_T2IRequest] = deque()
        self._flow = None                 # (FlowSession, [requests]) of the active group
        self.stats.update({"flow_steps": 0, "images": 0, "t2i_groups": 0, "interleaved_rounds": 0})

    def submit_t2i(self, prompt: str, image_shape=(256, 256), num_timesteps: int = 50, timestep_shift: float = 3.0,
                   cfg_text_scale: float = 4.0, cfg_img_scale: float = 1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min: float = 0.0,
                   cfg_renorm_type: str = "global", init_noise: Optional[torch.Tensor] = None) -> int:
        """Queue a text-to-image request (the defaults are interactive_image_generator.py:303-306's).  init_noise (optional,
        [h*w, patch*patch*z]) fixes the starting latent; otherwise it is drawn from torch's CPU generator at admission,
        as the reference does (bagel.py:893)."""
        rid = self._next_id
        self._next_id += 1
        params = (int(num_timesteps), float(timestep_shift), float(cfg_text_scale), float(cfg_img_scale),
                  tuple(float(v) for v in cfg_interval), float(cfg_renorm_min), str(cfg_renorm_type))
        self.t2i_queue.append(_T2IRequest(rid, prompt, tuple(int(v) for v in image_shape), params, init_noise))
        return rid

    # ------------------------------------------------------------------ the flow side of a round
    def _other_work(self) -> bool:
        return self._flow is not None or bool(self.t2i_queue)

    def _admit_t2i(self):
        """The next group: up to t2i_batch queued prompts with the same image shape and sampler parameters."""
        first = self.t2i_queue[0]
        group, rest = [], deque()
        while self.t2i_queue:
            r = self.t2i_queue.popleft()
            if len(group) < self.t2i_batch and r.image_shape == first.image_shape and r.params == first.params:
                group.append(r)
            else:
                rest.append(r)
        self.t2i_queue = rest
        m, ntid, n = self.model, self.new_token_ids, len(group)
        steps, shift, s_text, s_img, interval, rmin, rtype = first.params
        shapes = [first.image_shape] * n
        # gen context = the prompt; no-text context = empty; no-image context = the prompt again (inferencer.py:583-607)
        gen = NaiveCache(m.cfg.layers)
        gi, kvl, rope = m.prepare_prompts([0] * n, [0] * n, [r.prompt for r in group], self.tokenizer, ntid)
        gen = m.forward_cache_update_text(gen, **gi)
        gl = m.prepare_vae_latent(kvl, rope, shapes, ntid)
        if any(r.noise is not None for r in group):
            parts = list(gl["packed_init_noises"].split([gl["packed_init_noises"].shape[0] // n] * n))
            for j, r in enumerate(group):
                if r.noise is not None:
                    parts[j] = r.noise.to(parts[j].dtype).reshape(parts[j].shape)
            gl["packed_init_noises"] = torch.cat(parts, 0)
        gt = m.prepare_vae_latent_cfg([0] * n, [0] * n, shapes)
        gim = m.prepare_vae_latent_cfg(kvl, rope, shapes)
        from .bagel import FlowSession
        args = dict(gl)
        args.update(dict(
            past_key_values=gen, key_values_lens=gl.get("key_values_lens"), num_timesteps=steps, timestep_shift=shift,
            cfg_renorm_min=rmin, cfg_renorm_type=rtype, cfg_interval=interval, cfg_text_scale=s_text, cfg_img_scale=s_img,
            cfg_text_past_key_values=NaiveCache(m.cfg.layers), cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_img_past_key_values=gen.snapshot(), cfg_img_packed_position_ids=gim["cfg_packed_position_ids"]))
        self._flow = (FlowSession(m, args), group)
        self.stats["t2i_groups"] += 1

    def _between_rounds(self):
        if self._flow is None:
            if not self.t2i_queue:
                return
            self._admit_t2i()
        flow, group = self._flow
        before = flow.i
        flow.step(self.flow_steps_per_round)
        self.stats["flow_steps"] += flow.i - before
        self.stats["interleaved_rounds"] += 1
        if flow.finished:
            m = self.model
            lats = list(flow.latents())
            px = self.vae.decode_tokens_batch_to_uint8(lats, group[0].image_shape, m.latent_downsample, m.latent_patch_size)
            for j, r in enumerate(group):
                self.results[r.rid] = px[j].cpu()
                self.latents[r.rid] = lats[j].clone()
            self.stats["images"] += len(group)
            self._flow = None

    def run(self):
        """Serve everything submitted: {request id: answer text | uint8 [H, W, 3] image}; self.latents keeps the final latent
        tokens of every image (tests compare them with the requests served alone)."""
        self.latents: Dict[int, torch.Tensor] = {}
        if self.queue:
            return super().run()
        with torch.no_grad(), ops.device_scope(self.device):
            while self._other_work():          # no VQA request at all: only image groups
                self._between_rounds()
        return dict(self.results)
