"""Packed Qwen2-MoT forward on the HIP kernels.

Mirrors ``Qwen2ForCausalLM.forward_inference`` / ``Qwen2Model.forward_inference`` /
``Qwen2MoTDecoderLayer.forward_inference`` / ``PackedAttentionMoT.forward_inference``
(codes/modeling/unimedvl/qwen2_navit.py:1243, 1115-1176, 843-902, 525-626): same
arguments, same meaning.  Differences in mechanism, not in results:
  * the KV cache is appended in place (kvcache.NaiveCache) instead of being re-merged
    through packed_query_indexes / packed_key_value_indexes, which are therefore only
    validated, not used;
  * q/k/v projections are one fused GEMM, gate/up one SwiGLU GEMM, residual adds live in
    GEMM epilogues, q/k-norm + RoPE + cache write are one kernel;
  * MoT routing (mode="gen") runs each expert's GEMM over its row subset through a row
    index list instead of gather / zeros_like / scatter passes.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .config import UniMedVLConfig
from .kvcache import NaiveCache
from .weights import LLMWeights

BF16 = torch.bfloat16


@dataclass
class BaseNavitOutputWithPast:
    packed_query_sequence: torch.Tensor = None
    past_key_values: Optional[NaiveCache] = None


class ForwardPlan:
    """device-side bookkeeping of one packed forward (Qwen2MoT.make_plan)"""
    __slots__ = ("buf", "qlens", "max_kv", "tok_seg", "tok_slot", "tok_pos", "cu_q", "kv_len")


def _host_list(x):
    if isinstance(x, torch.Tensor):
        return [int(v) for v in x.tolist()]
    return [int(v) for v in x]


class Qwen2MoT:
    def __init__(self, cfg: UniMedVLConfig, weights: LLMWeights, device):
        self.cfg = cfg
        self.w = weights
        self.device = device
        self.decode_nsplit = 8
        import os
        # measured on MI355X: no gain (2.03 vs 2.02 img/s) - the 256x256 GEMM's 8 waves x 166 VGPRs + 128 KiB LDS
        # leave no room for the skinny GEMM to co-reside - so the two-stream variant is off by default
        self.overlap_experts = os.environ.get("UMV_OVERLAP_EXPERTS", "0") not in ("0", "")
        self._side = None

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ------------------------------------------------------------------ embeddings / head
    @ops.on_device
    def embed_tokens(self, ids, out=None, out_rows=None):
        if not ids.is_cuda and ids.numel():      # host-side ids (every prefill): the gather kernel indexes the table unchecked
            lo, hi = int(ids.min()), int(ids.max())
            if lo < 0 or hi >= self.w.embed.shape[0]:
                raise ValueError(f"token ids must lie in [0, {self.w.embed.shape[0]}); got [{lo}, {hi}]")
        ids = ids.to(device=self.device, dtype=torch.int64)
        return ops.embed_gather(self.w.embed, ids, out=out, out_rows=out_rows)

    @ops.on_device
    def lm_head(self, h):
        return ops.gemm(h, self.w.lm_head)

    # ------------------------------------------------------------------ per-token bookkeeping of one forward
    @ops.on_device
    def make_plan(self, qlens, position_ids, cache_lens, into: "ForwardPlan" = None) -> "ForwardPlan":
        """Everything a forward needs from the host, as device tensors: segment / slot / rope position of every query token,
        cu_seqlens of the queries, keys visible per segment after the call.  One small upload.  `into` refreshes an
        existing plan IN PLACE (same device addresses) - what a captured HIP graph of the forward reads at replay."""
        cfg, dev = self.cfg, self.device
        qlens = [int(q) for q in qlens]
        seg, slot = [], []
        for s, (c, q) in enumerate(zip(cache_lens, qlens)):
            seg += [s] * q
            slot += list(range(c, c + q))
        pos = position_ids.to(dtype=torch.int32).cpu() if isinstance(position_ids, torch.Tensor) else torch.tensor(position_ids, dtype=torch.int32)
        # the rotary tables hold cfg.max_position rows and the kernels index them unchecked
        pmax = int(pos.max()) if pos.numel() else 0
        pmin = int(pos.min()) if pos.numel() else 0
        if pmin < 0 or pmax >= cfg.max_position:
            raise ValueError(f"position ids must lie in [0, {cfg.max_position}) (max_position_embeddings); got [{pmin}, {pmax}]")
        if pos.numel() != len(seg):
            raise ValueError("position ids do not match the query lengths")
        cu = [0]
        for q in qlens:
            cu.append(cu[-1] + q)
        lens_after = [c + q for c, q in zip(cache_lens, qlens)]
        T, nseg = len(seg), len(qlens)
        host = torch.cat([torch.tensor(seg, dtype=torch.int32), torch.tensor(slot, dtype=torch.int32), pos.reshape(-1),
                          torch.tensor(cu, dtype=torch.int32), torch.tensor(lens_after, dtype=torch.int32)])
        if into is None:
            buf = host.to(dev, non_blocking=True)
            p = ForwardPlan()
            p.buf, p.qlens, p.max_kv = buf, qlens, 0
            p.tok_seg, p.tok_slot, p.tok_pos = buf[:T], buf[T:2 * T], buf[2 * T:3 * T]
            p.cu_q, p.kv_len = buf[3 * T:3 * T + nseg + 1], buf[3 * T + nseg + 1:3 * T + 2 * nseg + 1]
            return p
        if len(into.qlens) != nseg or sum(into.qlens) != T or max(into.qlens) != max(qlens):
            raise ValueError("a plan can only be refreshed for the same token count, segment count and longest segment")
        into.buf.copy_(host, non_blocking=True)     # (which segments hold the tokens may change: the kernels read it from here)
        into.qlens = qlens
        return into

    # ------------------------------------------------------------------ forward
    @ops.on_device
    def forward_inference(self, packed_query_sequence, query_lens, packed_query_position_ids,
                          packed_query_indexes=None, past_key_values: NaiveCache = None, key_values_lens=None,
                          packed_key_value_indexes=None, update_past_key_values=True, is_causal=True, mode="und",
                          packed_vae_token_indexes=None, packed_text_indexes=None, plan: "ForwardPlan" = None) -> BaseNavitOutputWithPast:
        cfg, w, dev = self.cfg, self.w, self.device
        nq, nkv, hd, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        seq = packed_query_sequence
        if seq.dtype != BF16:
            seq = seq.to(BF16)
        seq = seq.contiguous()
        T = seq.shape[0]
        qlens = _host_list(query_lens)
        nseg = len(qlens)
        assert sum(qlens) == T, "query_lens do not add up to the packed sequence"
        cache = past_key_values if past_key_values is not None else NaiveCache(cfg.layers)
        if cache.slabs is None:
            if hasattr(cache, "ensure_tokens"):
                cache.ensure_tokens([0] * nseg, nkv, hd, dev)
            else:
                cache.ensure(nseg, max(qlens), nkv, hd, dev)
        if key_values_lens is not None:
            kvl = _host_list(key_values_lens)
            if kvl != list(cache.lens):
                raise ValueError(f"key_values_lens {kvl} disagree with the cache ({cache.lens})")
        if hasattr(cache, "ensure_tokens"):      # paged cache: pages per segment, only where the context grows
            cache.ensure_tokens([c + q for c, q in zip(cache.lens, qlens)], nkv, hd, dev)
        else:
            cache.ensure(nseg, max(c + q for c, q in zip(cache.lens, qlens)), nkv, hd, dev)
        if plan is None:
            plan = self.make_plan(qlens, packed_query_position_ids, cache.lens)
        elif plan.qlens != qlens:
            raise ValueError("forward plan was made for other query lengths")
        tok_seg, tok_slot, tok_pos, cu_q, kv_len = plan.tok_seg, plan.tok_slot, plan.tok_pos, plan.cu_q, plan.kv_len
        lens_after = [c + q for c, q in zip(cache.lens, qlens)]
        max_q, max_kv = max(qlens), max(max(lens_after), plan.max_kv)

        gen = mode == "gen"
        expert = text_rows = vae_rows = None
        if gen:
            assert packed_vae_token_indexes is not None and packed_text_indexes is not None
            vae_rows = packed_vae_token_indexes.to(device=dev, dtype=torch.int32)
            text_rows = packed_text_indexes.to(device=dev, dtype=torch.int32)
            expert = torch.zeros(T, dtype=torch.int32, device=dev)
            expert[vae_rows.long()] = 1
            n_text, n_vae = text_rows.numel(), vae_rows.numel()

        decode = max_q == 1 and not gen
        nsplit = self.decode_nsplit if (decode and max_kv >= 256) else 1
        ws = ops.attn_workspace(nseg, nq, hd, max_q, nsplit, dev) if nsplit > 1 else None

        x = torch.empty_like(seq)
        qkv = torch.empty((T, (nq + 2 * nkv) * hd), dtype=BF16, device=dev)
        qbuf = torch.empty((T, nq, hd), dtype=BF16, device=dev)
        obuf = torch.empty((T, nq * hd), dtype=BF16, device=dev)
        act = torch.empty((T, cfg.inter), dtype=BF16, device=dev)
        if seq.data_ptr() == packed_query_sequence.data_ptr():
            seq = seq.clone()   # residual stream is updated in place; never clobber the caller's tensor

        # MoT routing: the few text rows stream the `und` expert's weights (HBM-bound skinny GEMM) while the latent
        # rows run the `gen` expert's MFMA-bound tiled GEMM.  They touch disjoint rows, so with overlap_experts the
        # two launch on different streams and share the chip instead of running back to back.
        main = torch.cuda.current_stream()
        side = self._side_stream() if (gen and self.overlap_experts) else None

        # W8A8 mode: every forward that is not a one-token decode step rounds its linear-layer inputs per row through e4m3
        act8 = bool(getattr(w, "act8", False)) and max_q > 1

        def routed(xin, und_lin, gen_lin, out, residual=None):
            if not gen:
                return ops.gemm(xin, und_lin, out=out, residual=residual, act8=act8)
            if side is None:
                ops.gemm(xin, und_lin, out=out, M=n_text, row_idx=text_rows, residual=residual, act8=act8)
                ops.gemm(xin, gen_lin, out=out, M=n_vae, row_idx=vae_rows, residual=residual, act8=act8)
                return out
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.gemm(xin, und_lin, out=out, M=n_text, row_idx=text_rows, residual=residual, act8=act8)
            ops.gemm(xin, gen_lin, out=out, M=n_vae, row_idx=vae_rows, residual=residual, act8=act8)
            main.wait_stream(side)
            return out

        for l in range(cfg.layers):
            lu = w.und[l]
            lg = w.gen[l] if gen else None
            ops.rmsnorm(seq, lu.in_norm, cfg.rms_eps, out=x, w_gen=lg.in_norm if gen else None, expert=expert)
            routed(x, lu.qkv, lg.qkv if gen else None, qkv)
            ops.qkv_post(qkv, qbuf, cache.slabs[l], tok_seg, tok_slot, tok_pos, nq, nkv, hd, cfg.rms_eps,
                         lu.q_norm, lu.k_norm, lg.q_norm if gen else None, lg.k_norm if gen else None, expert,
                         w.cos, w.sin, fp32_chain=gen)
            ops.attention(qbuf, obuf, cache.slabs[l], cu_q, kv_len, nq, nkv, hd, is_causal, max_q, max_kv, nsplit, ws)
            routed(obuf, lu.o, lg.o if gen else None, seq, residual=seq)
            ops.rmsnorm(seq, lu.post_norm, cfg.rms_eps, out=x, w_gen=lg.post_norm if gen else None, expert=expert)
            routed(x, lu.gate_up, lg.gate_up if gen else None, act)
            routed(act, lu.down, lg.down if gen else None, seq, residual=seq)
        out = ops.rmsnorm(seq, w.norm, cfg.rms_eps, w_gen=w.norm_gen if gen else None, expert=expert)
        if update_past_key_values:
            cache.lens = lens_after
        return BaseNavitOutputWithPast(packed_query_sequence=out, past_key_values=cache)
