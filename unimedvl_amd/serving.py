"""Continuous (in-flight) batching for VQA decode - SURVEY.md section 8f rank 4 ("serving-grade decode").

The reference decodes one request at a time and stops a whole batch when SAMPLE 0 emits EOS
(codes/modeling/unimedvl/bagel.py:1262-1314).  Here B cache segments ("slots") decode together in ONE captured HIP
graph (decode.DecodeSession); every `check_every` steps the host looks at the generated ids, retires the slots that
produced <|im_end|> (or hit their token budget) and prefills the next queued request straight into the freed slot:

  * the KV cache is one slab per layer with a fixed per-slot capacity (kvcache.NaiveCache.reserve), so a new request is
    a prefill on a ONE-SEGMENT VIEW of that slab (NaiveCache.view_segments) - no copy, no re-allocation, the other slots'
    keys are untouched;
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
from .kvcache import NaiveCache


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
                 use_graph: bool = True):
        self.model, self.tokenizer, self.new_token_ids, self.image_transform = model, tokenizer, new_token_ids, image_transform
        self.device = model.device
        self.slots, self.check_every = int(slots), int(check_every)
        self.max_context, self.default_new = int(max_context), int(max_new_tokens)
        self.do_sample, self.temperature, self.use_graph = do_sample, temperature, use_graph
        cfg = model.cfg
        self.cache = NaiveCache(cfg.layers)
        self.cache.reserve(self.slots, self.max_context + self.default_new + self.check_every + 8, cfg.kv_heads, cfg.head_dim,
                           model.device)
        self.queue: Deque[_Request] = deque()
        self.results: Dict[int, str] = {}
        self._next_id = 0
        self.stats = {"decode_steps": 0, "prefills": 0, "tokens": 0}

    # ------------------------------------------------------------------ API
    def submit(self, images, prompt: str, max_new_tokens: Optional[int] = None) -> int:
        """images: one image / a list of images (PIL or [3,H,W] tensors, as the transform accepts) or None."""
        imgs = [] if images is None else (list(images) if isinstance(images, (list, tuple)) else [images])
        budget = self.default_new if max_new_tokens is None else int(max_new_tokens)
        if budget > self.default_new:
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
        cache.lens = [0] * B
        first = []
        for b in range(B):
            if self.queue:
                active[b] = self.queue.popleft()
                first.append(b)
        for b, st in zip(first, self._prefill_many(first, [active[b] for b in first])):
            state[b] = st
        start = torch.tensor([s[0] for s in state], dtype=torch.int64)
        pos = torch.tensor([s[2] for s in state], dtype=torch.int64)
        seed = m._sampling_seed() if self.do_sample else 0
        lens_now = list(cache.lens)
        sess = DecodeSession(m.language_model, cache, start, pos, self.check_every, use_graph=self.use_graph,
                             do_sample=self.do_sample, temperature=self.temperature, seed=seed)
        cache.lens = lens_now                           # the session only reads them; this loop owns the bookkeeping
        while any(r is not None for r in active):
            k = self.check_every
            sess.rewind_outputs()
            sess.step(k)
            self.stats["decode_steps"] += k
            ids = sess.pred_ids[:k].cpu()               # [k, B]; the only host sync of the round
            freed = []
            for b in range(B):
                req = active[b]
                if req is None:
                    sess.set_slot(b, bos, 0, 0)         # idle slot: keep it parked at the start of its segment
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
                    continue
                self._finish(req)
                active[b] = None
                cache.lens[b] = 0
                if self.queue:
                    active[b] = self.queue.popleft()
                    freed.append(b)
                else:
                    sess.set_slot(b, bos, 0, 0)
            # every slot that was freed this round is refilled by ONE batched prefill (images of the new requests share the
            # ViT and LLM forward; the other slots take part with zero query tokens)
            for b, (tok, kvl, rope) in zip(freed, self._prefill_many(freed, [active[b] for b in freed])):
                sess.set_slot(b, tok, kvl, rope)
        return dict(self.results)

    # ------------------------------------------------------------------ internals
    def _prefill(self, b: int, req: _Request):
        """Image(s) then the prompt into slot b (Bagel.chat's order, bagel.py:1321-1392); returns the slot's decode state."""
        m = self.model
        view = self.cache.view_segments(b, b + 1)
        view.lens = [0]
        cap = view.cap
        kvl, rope = [0], [0]
        for image in req.images:
            gi, kvl, rope = m.prepare_vit_images(kvl, rope, [image], self.image_transform, self.new_token_ids)
            self._check_room(kvl[0], req, cap)
            view = m.forward_cache_update_vit(view, **gi)
        gi, kvl, rope = m.prepare_prompts(kvl, rope, [req.prompt], self.tokenizer, self.new_token_ids)
        self._check_room(kvl[0], req, cap)
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
            cache.lens[b] = 0
        k = len(slots)
        kvl, rope = [0] * k, [0] * k

        def spread(per_request):          # [k] query lengths -> [B] with zeros for the slots that do not take part
            full = torch.zeros(B, dtype=per_request.dtype)
            full[torch.tensor(slots)] = per_request.cpu()
            return full
        for j in range(len(reqs[0].images)):
            gi, kvl, rope = m.prepare_vit_images(kvl, rope, [r.images[j] for r in reqs], self.image_transform, self.new_token_ids)
            for n, r in zip(kvl, reqs):
                self._check_room(n, r, cap)
            gi["packed_seqlens"] = spread(gi["packed_seqlens"])
            gi["key_values_lens"] = gi["packed_key_value_indexes"] = gi["packed_indexes"] = None   # written for a k-sample cache
            m.forward_cache_update_vit(cache, **gi)
        gi, kvl, rope = m.prepare_prompts(kvl, rope, [r.prompt for r in reqs], self.tokenizer, self.new_token_ids)
        for n, r in zip(kvl, reqs):
            self._check_room(n, r, cap)
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

    def _check_room(self, ctx_tokens: int, req: _Request, cap: int):
        if ctx_tokens > self.max_context or ctx_tokens + req.max_new_tokens + self.check_every + 1 > cap:
            raise ValueError(f"request {req.rid}: context of {ctx_tokens} tokens exceeds max_context={self.max_context}")

    def _finish(self, req: _Request):
        bos = self.new_token_ids["bos_token_id"]
        text = self.tokenizer.decode(torch.tensor([bos] + req.tokens, dtype=torch.int64))
        self.results[req.rid] = text.split("<|im_end|>")[0].split("<|im_start|>")[1]   # inferencer.py:277-278
        self.stats["tokens"] += len(req.tokens)
