"""Greedy KV-cache decode with all bookkeeping on the device.

The reference loop (codes/modeling/unimedvl/bagel.py:1262-1314) rebuilds its index
tensors with .tolist() splits every step, re-merges the whole KV tensor in every layer
and syncs with the host three times per token.  Here one decode step is a fixed kernel
sequence over static buffers (embed -> rmsnorm -> 28 x [QKV GEMM, fused q/k-norm + RoPE +
KV append + split-KV attention, combine, o_proj + residual, rmsnorm, SwiGLU GEMM,
down + residual, next rmsnorm] -> lm_head -> argmax -> advance), captured once into a HIP
graph and replayed; slot, position and kv_len counters live in device memory
(umv_decode_advance).

Each of the three N = 3584 / 4608 GEMMs (QKV, o, down) can run split along K ("split-K"): fp32 partial
sums that the kernel which follows anyway finishes - the attention kernel (or umv_qkv_post) sums the QKV
partials, umv_residual_rmsnorm_bf16 = partial sum + residual add + the next RMSNorm - so a split never
adds a launch and the summation order stays fixed.

Tried and measured slower on MI355X, no longer wired in here (profiles/HISTORY.md section 5b): RMSNorm folded into the next
GEMM's prologue (norm_w of umv_gemm_bf16: every workgroup pays the normalise + stage latency before its
first MFMA, step 3.40 -> 3.93 ms) and weight prefetch into the Infinity Cache on a parallel graph branch
(umv_prefetch: the branch does not overlap under graph replay, 3.43 -> 4.5-5.4 ms).
"""
import os

import torch

from . import ops
from .kvcache import NaiveCache

BF16 = torch.bfloat16


class DecodeSession:
    def __init__(self, llm, cache: NaiveCache, start_tokens, positions, max_length, use_graph=True, nsplit=None,
                 do_sample=False, temperature=1.0, seed=0):
        with ops.device_scope(llm.device):
            self._init(llm, cache, start_tokens, positions, max_length, use_graph, nsplit, do_sample, temperature, seed)

    @property
    def device(self):
        return self.dev

    def _init(self, llm, cache, start_tokens, positions, max_length, use_graph, nsplit, do_sample, temperature, seed):
        cfg, dev = llm.cfg, llm.device
        self.llm, self.cache, self.cfg, self.dev = llm, cache, cfg, dev
        B = len(cache.lens)
        self.B = B
        self.max_length = max_length
        nq, nkv, hd, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        if hasattr(cache, "ensure_tokens"):      # paged cache: pages for every slot's own decode horizon (the captured step reads the table)
            cache.ensure_tokens([n + max_length + 1 for n in cache.lens], nkv, hd, dev)
        else:
            cache.ensure(B, max(cache.lens) + max_length + 1, nkv, hd, dev)
        self.lens0 = list(cache.lens)
        i32 = dict(dtype=torch.int32, device=dev)
        self.ids = start_tokens.to(device=dev, dtype=torch.int64).clone()
        self.tok_seg = torch.arange(B, **i32)
        self.tok_slot = torch.tensor(cache.lens, dtype=torch.int32).to(dev)
        pmax = int(positions.max()) if positions.numel() else 0
        if (int(positions.min()) if positions.numel() else 0) < 0 or pmax + max_length >= cfg.max_position:
            raise ValueError(f"decode would reach rope position {pmax + max_length} >= max_position_embeddings {cfg.max_position}")
        self.tok_pos = positions.to(device=dev, dtype=torch.int32).clone()
        self.kv_len = self.tok_slot + 1
        self.cu_q = torch.arange(B + 1, **i32)
        self.in_ids = torch.zeros((max_length, B), dtype=torch.int64, device=dev)    # token fed at each step
        self.pred_ids = torch.zeros((max_length, B), dtype=torch.int64, device=dev)  # token predicted at each step
        if max_length > 0:
            self.in_ids[0].copy_(self.ids)
        # step counter(s): entry 0 is THE counter (sampling, the unfused step end); the fused argmax step end keeps one per sample
        # so that its workgroups share no word (umv_decode_step_end_argmax)
        self.step_idx = torch.zeros(max(B, 1), dtype=torch.int64, device=dev)
        max_kv = max(cache.lens) + max_length + 1
        # split the key range so that every wavefront walks ~2 blocks of 32 keys (latency bound otherwise)
        if nsplit is None and os.environ.get("UMV_DECODE_NSPLIT"):
            nsplit = int(os.environ["UMV_DECODE_NSPLIT"])   # tuning only
        # ... but no more than ~1024 waves in all: with many samples the partial (O, m, l) traffic and the combine grow with the
        # split count (B = 32, context 1156: 4.45 ms per step at 8 splits, 4.53 at 19; B = 8 is unchanged by the cap)
        # round 4 (single-wave workgroups; profiles/r04_decode_nsplit_sweep.txt): B = 32, context 1156: 6 splits 4.009 ms, 8 splits 4.061,
        # 4 splits 4.046, 12 splits 4.064; B = 8, context 1060: 24 splits 3.116, 17 splits 3.125, 28 splits 3.131 -> ~768 waves at
        # 9 .. 32 samples (1024 otherwise: unmeasured, unchanged) and ~48 keys per split up to 8 samples
        if nsplit is None:
            keys = 48 if B <= 8 else 64
            nsplit = max(1, min(32, (max_kv + keys - 1) // keys, max(1, (768 if 8 < B <= 32 else 1024) // (B * nkv))))
        self.nsplit = nsplit
        # waves per attention workgroup splitting its key range with an LDS merge (umv_attn_args.wave_split: 0 = single-wave workgroups);
        # UMV_DECODE_WSPLIT=2|4 for the A/B of profiles/r06_decode_wave_split.txt
        self.wave_split = int(os.environ.get("UMV_DECODE_WSPLIT", "0") or 0)
        self.ws = ops.attn_workspace(B, nq, hd, 1, self.nsplit, dev) if self.nsplit > 1 else None
        self.max_kv = max_kv
        # static activations
        self.seq = torch.empty((B, H), dtype=BF16, device=dev)
        self.x = torch.empty((B, H), dtype=BF16, device=dev)
        self.qkv = torch.empty((B, (nq + 2 * nkv) * hd), dtype=BF16, device=dev)
        self.q = torch.empty((B, nq, hd), dtype=BF16, device=dev)
        self.o = torch.empty((B, nq * hd), dtype=BF16, device=dev)
        self.act = torch.empty((B, cfg.inter), dtype=BF16, device=dev)
        self.hn = torch.empty((B, H), dtype=BF16, device=dev)
        self.logits = torch.empty((B, cfg.vocab), dtype=BF16, device=dev)
        # greedy: the argmax rides on the lm_head epilogue (one key per 16-column tile and row) and is finished by the
        # step-end kernel - the separate 25 us argmax launch over the logits is gone (UMV_DECODE_FUSED_ARGMAX=0 restores it)
        # sampling rides the same way (round 5): the keys order bf16(logit / T) + Gumbel noise, their maximum is one draw from the softmax -
        # the sampled step has the greedy step's two-launch tail (lm_head -> sample -> step_end was 3.29 ms against 3.12)
        self.fused_argmax = B <= 64 and os.environ.get("UMV_DECODE_FUSED_ARGMAX", "1") not in ("0", "")
        if self.fused_argmax:
            self.amax_part = torch.zeros((B, (cfg.vocab + 15) // 16), dtype=torch.int64, device=dev)
        w = llm.w
        # K splits of the QKV / o / down GEMMs: "q,o,d" (1 = that GEMM is not split), "0" = none, "auto" by batch / weights.
        # Measured on MI355X (bench.py --batch B, ms per step), no split -> 3,4,4:
        #   B=12: 3.72 -> 3.59, 16: 4.09 -> 3.85, 32: 5.48 -> 4.67, 64: 9.07 -> 6.35; more splits are slower (4,4,8: 4.95 at
        #   B=32); e4m3 weights: x is as many bytes as the weights already at B=8, the split pays there too (2.63 -> 2.52).
        # bf16 at B <= 8 (tools/skinny_bench.py, us per GEMM): QKV 11.4 -> 8.5 with 3 splits, down 25.9 -> 23.3 with 4,
        #   o_proj 8.7 either way (its exact-partition image without a split keeps the residual add in the GEMM).
        sk = os.environ.get("UMV_DECODE_SPLITK", "auto")
        if sk == "auto":
            # (B <= 8 bf16: 3,4,4 3.151 ms vs 3,1,4 3.161 - a wash, so one setting for every batch and weight type; it also
            # makes the exact-partition copy of the o_proj weights unnecessary)
            # 65..128 rows: the same scheme on the 128 x 128 MFMA tile (umv_gemm_bf16 routes k_splits > 1 there above 64 rows):
            # without it down_proj is 14 workgroups of 256 x 256 (151 us); bf16 weights only (the e4m3 image is an M <= 64 layout)
            sk = "3,4,4" if B <= 64 else ("6,8,8" if B <= 128 and not getattr(w, "fp8", False) else "0")
        self.sk = (1, 1, 1) if sk in ("0", "") else tuple(max(1, int(v)) for v in sk.split(","))
        if len(self.sk) != 3 or any(v > 64 for v in self.sk):
            raise ValueError(f"UMV_DECODE_SPLITK={sk!r}: expected 'q,o,d' with 1 <= splits <= 64")
        if self.sk != (1, 1, 1) and B > 128:
            raise ValueError("split-K decode mode serves at most 128 samples per step")
        sq, so, sd = self.sk
        self.p_qkv = torch.empty((sq, B, (nq + 2 * nkv) * hd), dtype=torch.float32, device=dev) if sq > 1 else None
        self.p_h = torch.empty((max(so, sd), B, H), dtype=torch.float32, device=dev) if max(so, sd) > 1 else None
        # decode-only weight copies with exact-partition tiles (N/256 rows per tile) for the GEMMs that run without a split
        exact = os.environ.get("UMV_DECODE_EXACT_TILES", "1") not in ("0", "") and B <= 64
        need = (sq == 1, so == 1, sd == 1)
        if exact and any(need):
            if not hasattr(w, "decode_copies"):
                w.decode_copies = [[None, None, None] for _ in w.und]
            for lw, slot in zip(w.und, w.decode_copies):
                for i, lin in enumerate((lw.qkv, lw.o, lw.down)):
                    if need[i] and slot[i] is None:
                        slot[i] = lin.for_decode()
            self.dec = [tuple(c if c is not None else lin for c, lin in zip(slot, (lw.qkv, lw.o, lw.down)))
                        for lw, slot in zip(w.und, w.decode_copies)]
        else:
            self.dec = [(lw.qkv, lw.o, lw.down) for lw in w.und]
        # (q/k norm + RoPE + KV append folded into the attention kernel - umv_attn_decode_fused of experimental/ - wins only when the
        # QKV GEMM is not split: B = 8: 3.32 vs 3.39 ms without a split, 3.28 vs 3.24 with the default 3-way split, B = 32: 4.80 vs
        # 4.52; the default path splits, so the step always runs umv_qkv_post + umv_attn_varlen)
        self.do_sample, self.temperature, self.seed = bool(do_sample), float(temperature), int(seed)
        self.steps_done = 0
        self.graph = None
        if use_graph:
            self._capture()

    def _step(self):
        cfg, w, c = self.cfg, self.llm.w, self.cache
        nq, nkv, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
        L = cfg.layers
        sq, so, sd = self.sk
        ops.embed_gather(w.embed, self.ids, out=self.seq)      # in_ids[step] already holds these tokens (decode_step_end)
        ops.rmsnorm(self.seq, w.und[0].in_norm, cfg.rms_eps, out=self.x)
        for l in range(L):
            lw = w.und[l]
            qkv_w, o_w, down_w = self.dec[l]
            # ---- attention block: QKV (bias in the GEMM epilogue, or added by the consumer of the partial sums)
            if sq > 1:
                ops.gemm_splitk(self.x, lw.qkv, self.p_qkv, sq)
                part = dict(partials=self.p_qkv, bias=lw.qkv.bias)
            else:
                ops.gemm(self.x, qkv_w, out=self.qkv)
                part = {}
            ops.qkv_post(None if part else self.qkv, self.q, c.slabs[l], self.tok_seg, self.tok_slot, self.tok_pos, nq, nkv, hd,
                         cfg.rms_eps, lw.q_norm, lw.k_norm, cos_tab=w.cos, sin_tab=w.sin, **part)
            ops.attention(self.q, self.o, c.slabs[l], self.cu_q, self.kv_len, nq, nkv, hd, True, 1, self.max_kv,
                          self.nsplit, self.ws, wave_split=self.wave_split)
            # ---- o_proj + residual, then the post-attention norm
            if so > 1:
                ops.gemm_splitk(self.o, lw.o, self.p_h[:so], so)
                ops.residual_rmsnorm(self.p_h[:so], self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
            else:
                ops.gemm(self.o, o_w, out=self.seq, residual=self.seq)
                ops.rmsnorm(self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
            # ---- MLP: SwiGLU in the gate/up epilogue, down + residual, then the NEXT layer's input norm (or the final norm)
            ops.gemm(self.x, lw.gate_up, out=self.act)
            last = l + 1 == L
            nxt, dst = (w.norm, self.hn) if last else (w.und[l + 1].in_norm, self.x)
            if sd > 1:
                ops.gemm_splitk(self.act, lw.down, self.p_h[:sd], sd)
                ops.residual_rmsnorm(self.p_h[:sd], self.seq, nxt, cfg.rms_eps, out=dst)
            else:
                ops.gemm(self.act, down_w, out=self.seq, residual=self.seq)
                ops.rmsnorm(self.seq, nxt, cfg.rms_eps, out=dst)
        if self.fused_argmax:
            ops.gemm(self.hn, w.lm_head, out=self.logits, argmax_partial=self.amax_part,
                     sample=(self.temperature, self.seed, self.step_idx) if self.do_sample else None)
            # ids = argmax; pred_ids[step] = in_ids[step + 1] = ids; slot / position / kv_len / step += 1: one launch
            ops.decode_step_end_argmax(self.tok_slot, self.tok_pos, self.kv_len, self.amax_part, self.ids, self.in_ids, self.pred_ids,
                                       self.step_idx)
            return
        ops.gemm(self.hn, w.lm_head, out=self.logits)
        if self.do_sample:
            ops.sample(self.logits, self.temperature, self.seed, step=self.step_idx, out=self.ids)
        else:
            ops.argmax(self.logits, out=self.ids)
        # pred_ids[step] = in_ids[step + 1] = ids; slot / position / kv_len / step += 1: one launch
        ops.decode_step_end(self.tok_slot, self.tok_pos, self.kv_len, self.ids, self.in_ids, self.pred_ids, self.step_idx)

    def _capture(self):
        # the captured step appends at slot = lens0 and bumps the counters; warm up on a side
        # stream first (lazy module loading), then restore the counters so capture sees a clean state
        saved = [t.clone() for t in (self.ids, self.tok_slot, self.tok_pos, self.kv_len, self.step_idx)]
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for t, v in zip((self.ids, self.tok_slot, self.tok_pos, self.kv_len, self.step_idx), saved):
            t.copy_(v)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        # capture does not execute; counters are still at their initial values
        self.graph = g

    @ops.on_device
    def step(self, n=1):
        for _ in range(n):
            if self.steps_done >= self.max_length:
                raise RuntimeError("decode session exhausted")
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
            self.steps_done += 1

    # ---- continuous batching support (serving.py): the captured step reads these from device memory, so a slot can be
    # re-pointed at a new request between replays without re-capturing
    @ops.on_device
    def set_slot(self, b, token, kv_len, pos):
        """Sample b continues from `token` with `kv_len` tokens already in its cache segment and rope position `pos`."""
        self.ids[b:b + 1].fill_(int(token))
        self.tok_slot[b:b + 1].fill_(int(kv_len))
        self.kv_len[b:b + 1].fill_(int(kv_len) + 1)
        self.tok_pos[b:b + 1].fill_(int(pos))

    @ops.on_device
    def rewind_outputs(self):
        """Start writing in_ids / pred_ids at row 0 again (the caller has harvested the previous rows)."""
        self.step_idx.zero_()
        self.in_ids[0].copy_(self.ids)      # the tokens the next step is fed (set_slot may have changed them)
        self.steps_done = 0

    def commit(self, steps=None):
        """Make the cache's host-side lengths reflect `steps` decoded tokens."""
        steps = self.steps_done if steps is None else steps
        self.cache.lens = [l + steps for l in self.lens0]
