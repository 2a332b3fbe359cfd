#!/usr/bin/env python
"""A/B of the decode step with the EXPERIMENTAL fused attention (q/k norm + RoPE + KV append inside the attention kernel,
experimental/csrc/attention_decode.hip) against the shipped qkv_post + attention pair, at 14B dims, B x context 1060, under a
HIP graph.  MODE = shipped | fused (fused needs UMV_DECODE_SPLITK with an unsplit QKV, e.g. "1,4,4").  Tuning only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402
from unimedvl_amd.bagel import Bagel  # noqa: E402
from unimedvl_amd.config import UniMedVLConfig  # noqa: E402
from unimedvl_amd.decode import DecodeSession  # noqa: E402
from unimedvl_amd.kvcache import NaiveCache  # noqa: E402
from unimedvl_amd.weights import random_getter  # noqa: E402

MODE = os.environ.get("MODE", "shipped")
B, CTX, STEPS = int(os.environ.get("B", "8")), 1060, 256
cfg = UniMedVLConfig()
dev = torch.device("cuda", 0)
model = Bagel(cfg, random_getter(cfg, dev, seed=1234), device=dev, visual_gen=False, visual_und=False)
cache = NaiveCache(cfg.layers)
cache.reserve(B, CTX + STEPS + 40, cfg.kv_heads, cfg.head_dim, dev)
for sl in cache.slabs:
    sl.k.normal_()
    sl.vt.normal_()
cache.lens = [CTX] * B


class Fused(DecodeSession):
    def _step(self):
        from experimental import ops as xops
        cfg, w, c = self.cfg, self.llm.w, self.cache
        nq, nkv, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
        L = cfg.layers
        sq, so, sd = self.sk
        assert sq == 1
        ops.embed_gather(w.embed, self.ids, out=self.seq)
        ops.rmsnorm(self.seq, w.und[0].in_norm, cfg.rms_eps, out=self.x)
        for l in range(L):
            lw = w.und[l]
            qkv_w, o_w, down_w = self.dec[l]
            ops.gemm(self.x, qkv_w, out=self.qkv)
            xops.attn_decode_fused(self.qkv, self.o, c.slabs[l], self.cu_q, self.kv_len, self.tok_pos, nq, nkv, hd,
                                   cfg.rms_eps, lw.q_norm, lw.k_norm, w.cos, w.sin, self.nsplit, self.ws)
            if so > 1:
                ops.gemm_splitk(self.o, lw.o, self.p_h[:so], so)
                ops.residual_rmsnorm(self.p_h[:so], self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
            else:
                ops.gemm(self.o, o_w, out=self.seq, residual=self.seq)
                ops.rmsnorm(self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
            ops.gemm(self.x, lw.gate_up, out=self.act)
            last = l + 1 == L
            nxt, dst = (w.norm, self.hn) if last else (w.und[l + 1].in_norm, self.x)
            if sd > 1:
                ops.gemm_splitk(self.act, lw.down, self.p_h[:sd], sd)
                ops.residual_rmsnorm(self.p_h[:sd], self.seq, nxt, cfg.rms_eps, out=dst)
            else:
                ops.gemm(self.act, down_w, out=self.seq, residual=self.seq)
                ops.rmsnorm(self.seq, nxt, cfg.rms_eps, out=dst)
        ops.gemm(self.hn, w.lm_head, out=self.logits, argmax_partial=self.amax_part)
        ops.decode_step_end_argmax(self.tok_slot, self.tok_pos, self.kv_len, self.amax_part, self.ids, self.in_ids, self.pred_ids, self.step_idx)


cls = Fused if MODE == "fused" else DecodeSession
start = torch.full((B,), 1234, dtype=torch.int64)
sess = cls(model.language_model, cache, start, torch.full((B,), CTX, dtype=torch.int64), STEPS + 16, use_graph=True)
sess.step(8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
sess.step(STEPS)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / STEPS
print(f"MODE={MODE} SPLITK={os.environ.get('UMV_DECODE_SPLITK', 'auto')} B={B}: {ms:.4f} ms/step  {B / ms * 1e3:.1f} tokens/s")
