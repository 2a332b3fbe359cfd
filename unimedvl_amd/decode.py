"""Greedy KV-cache decode with all bookkeeping on the device.

The reference loop (codes/modeling/unimedvl/bagel.py:1262-1314) rebuilds its index
tensors with .tolist() splits every step, re-merges the whole KV tensor in every layer
and syncs with the host three times per token.  Here one decode step is a fixed kernel
sequence over static buffers (embed -> 28 x [rmsnorm, QKV GEMM, q/k-norm+RoPE+KV append,
split-KV attention, o_proj+residual, rmsnorm, SwiGLU GEMM, down+residual] -> norm ->
lm_head -> argmax -> advance), captured once into a HIP graph and replayed; slot,
position and kv_len counters live in device memory (umv_decode_advance).
"""
import torch

from . import ops
from .kvcache import NaiveCache

BF16 = torch.bfloat16


class DecodeSession:
    def __init__(self, llm, cache: NaiveCache, start_tokens, positions, max_length, use_graph=True, nsplit=None,
                 fuse_norm=False, prefetch=None, do_sample=False, temperature=1.0, seed=0):
        cfg, dev = llm.cfg, llm.device
        self.llm, self.cache, self.cfg, self.dev = llm, cache, cfg, dev
        B = len(cache.lens)
        self.B = B
        self.max_length = max_length
        nq, nkv, hd, H = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden
        cache.ensure(B, max(cache.lens) + max_length + 1, nkv, hd, dev)
        self.lens0 = list(cache.lens)
        i32 = dict(dtype=torch.int32, device=dev)
        self.ids = start_tokens.to(device=dev, dtype=torch.int64).clone()
        self.tok_seg = torch.arange(B, **i32)
        self.tok_slot = torch.tensor(cache.lens, dtype=torch.int32).to(dev)
        self.tok_pos = positions.to(device=dev, dtype=torch.int32).clone()
        self.kv_len = self.tok_slot + 1
        self.cu_q = torch.arange(B + 1, **i32)
        self.in_ids = torch.zeros((max_length, B), dtype=torch.int64, device=dev)    # token fed at each step
        self.pred_ids = torch.zeros((max_length, B), dtype=torch.int64, device=dev)  # token predicted at each step
        self.step_idx = torch.zeros(1, dtype=torch.int64, device=dev)
        max_kv = max(cache.lens) + max_length + 1
        # split the key range so that every wavefront walks ~2 blocks of 32 keys (latency bound otherwise)
        import os as _os
        if nsplit is None and _os.environ.get("UMV_DECODE_NSPLIT"):
            nsplit = int(_os.environ["UMV_DECODE_NSPLIT"])   # tuning only
        self.nsplit = nsplit if nsplit is not None else max(1, min(32, (max_kv + 63) // 64))
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
        # RMSNorm can fold into the following GEMM's prologue (norm_w argument of umv_gemm_bf16), but on MI355X the
        # per-workgroup normalise+stage prologue costs more than the ~5 us standalone norm it replaces at B=8, even with
        # its loads batched into one round trip (qkv 11.2 -> 19.0 us, gate/up 42.7 -> 61.7 us, step 3.40 -> 3.93 ms): the
        # x / norm_w loads queue behind the first weight chunk and every workgroup pays an HBM latency before its first
        # MFMA.  The persistent variant (umv_gemm_decode, prologue once per CU) narrows it to +4..6 us per GEMM - still no
        # gain - so fusion stays off by default (UMV_DECODE_FUSE_NORM=1 to experiment).
        import os
        fuse_norm = os.environ.get("UMV_DECODE_FUSE_NORM", "1" if fuse_norm else "0") not in ("0", "")
        self.fuse_norm = fuse_norm and B <= 16 and H <= 4096 and not llm.w.fp8
        # Experimental (OFF): weight prefetch into the Infinity Cache on a parallel graph branch during the
        # latency-bound kernels (window sizes in MiB).  Measured on MI355X: the branch does not overlap with the
        # main chain under hipGraph replay and the step gets SLOWER (3.43 -> 4.5-5.4 ms), so it stays disabled;
        # UMV_DECODE_PREFETCH=1 re-enables it for experiments.
        import os
        pf = os.environ.get("UMV_DECODE_PREFETCH", "0" if prefetch is None else str(int(prefetch)))
        self.prefetch = pf not in ("0", "") and use_graph
        self.pf_w1 = int(os.environ.get("UMV_PF_W1", "24"))
        self.pf_w2 = int(os.environ.get("UMV_PF_W2", "64"))
        self.pf_w3 = int(os.environ.get("UMV_PF_W3", "24"))
        self.pf_blocks = int(os.environ.get("UMV_PF_BLOCKS", "128"))
        self.pf_stream = torch.cuda.Stream(device=dev) if self.prefetch else None
        # decode-only weight copies with exact-partition tiles (N/256 rows per tile) for the N=3584-class GEMMs
        exact = os.environ.get("UMV_DECODE_EXACT_TILES", "1") not in ("0", "") and B <= 64
        w = llm.w
        if exact and not hasattr(w, "decode_copies"):
            w.decode_copies = [(lw.qkv.for_decode(), lw.o.for_decode(), lw.down.for_decode()) for lw in w.und]
        self.dec = w.decode_copies if exact else [(lw.qkv, lw.o, lw.down) for lw in w.und]
        # split-K mode for the N = 3584 / 4608 GEMMs (bf16 weights): "qkv,o,down" split counts, "0" = off, "auto" by batch
        sk = os.environ.get("UMV_DECODE_SPLITK", "auto")
        self.sk = None
        if sk not in ("0", "") and B <= 64:
            if sk == "auto":
                # measured on MI355X (bench.py --batch B): B=8 3.448 -> 3.426 ms (noise), 12: 3.72 -> 3.59, 16: 4.09 -> 3.85,
                # 32: 5.48 -> 4.67, 64: 9.07 -> 6.35; more splits are slower (4,4,8: 4.95 at B=32; 4,4,16: 5.22)
                # e4m3 weights: x is as many bytes as the weights already at B=8, the split pays there too (2.63 -> 2.52 ms)
                sk = "3,4,4" if (B > 8 or w.fp8) else "0"
            if sk != "0":
                self.sk = tuple(int(v) for v in sk.split(","))
                assert len(self.sk) == 3 and all(1 < v <= 64 for v in self.sk)
                self.p_qkv = torch.empty((self.sk[0], B, (nq + 2 * nkv) * hd), dtype=torch.float32, device=dev)
                self.p_h = torch.empty((max(self.sk[1], self.sk[2]), B, H), dtype=torch.float32, device=dev)
        self.fuse_attn = os.environ.get("UMV_DECODE_FUSE_ATTN", "1") not in ("0", "") and hd == 128 and self.sk is None
        self.do_sample, self.temperature, self.seed = bool(do_sample), float(temperature), int(seed)
        self.steps_done = 0
        self.graph = None
        if use_graph:
            self._capture()

    def _prefetch(self, jobs):
        """Fork: stream the given (tensor, offset, nbytes) weight ranges into the Infinity Cache on the side
        stream while the main stream runs latency-bound kernels.  Returns nothing; call _join() before the
        consumer so the branch rejoins the (captured) main stream."""
        if not self.prefetch:
            return
        main = torch.cuda.current_stream()
        self.pf_stream.wait_stream(main)
        for t, off, n in jobs:
            ops.prefetch(t, n, off, blocks=self.pf_blocks, stream=self.pf_stream)

    def _join(self):
        if self.prefetch:
            torch.cuda.current_stream().wait_stream(self.pf_stream)

    def _step_splitk(self):
        """Same step with the N=3584/4608 GEMMs in split-K mode: 4 n-tiles per workgroup share each x fragment and the
        fp32 partial sums are finished by the kernel that follows anyway (qkv_post; residual add + the next RMSNorm)."""
        cfg, w, c = self.cfg, self.llm.w, self.cache
        nq, nkv, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
        L = cfg.layers
        sq, so, sd = self.sk
        self.in_ids.index_copy_(0, self.step_idx, self.ids.unsqueeze(0))
        ops.embed_gather(w.embed, self.ids, out=self.seq)
        ops.rmsnorm(self.seq, w.und[0].in_norm, cfg.rms_eps, out=self.x)
        for l in range(L):
            lw = w.und[l]
            ops.gemm_splitk(self.x, lw.qkv, self.p_qkv, sq)
            ops.qkv_post(None, self.q, c.slabs[l], self.tok_seg, self.tok_slot, self.tok_pos, nq, nkv, hd, cfg.rms_eps,
                         lw.q_norm, lw.k_norm, cos_tab=w.cos, sin_tab=w.sin, partials=self.p_qkv, bias=lw.qkv.bias)
            ops.attention(self.q, self.o, c.slabs[l], self.cu_q, self.kv_len, nq, nkv, hd, True, 1, self.max_kv,
                          self.nsplit, self.ws)
            ops.gemm_splitk(self.o, lw.o, self.p_h[:so], so)
            ops.residual_rmsnorm(self.p_h[:so], self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
            ops.gemm(self.x, lw.gate_up, out=self.act)
            ops.gemm_splitk(self.act, lw.down, self.p_h[:sd], sd)
            last = l + 1 == L
            ops.residual_rmsnorm(self.p_h[:sd], self.seq, w.norm if last else w.und[l + 1].in_norm, cfg.rms_eps,
                                 out=self.hn if last else self.x)
        ops.gemm(self.hn, w.lm_head, out=self.logits)
        if self.do_sample:
            ops.sample(self.logits, self.temperature, self.seed, step=self.step_idx, out=self.ids)
        else:
            ops.argmax(self.logits, out=self.ids)
        self.pred_ids.index_copy_(0, self.step_idx, self.ids.unsqueeze(0))
        ops.decode_advance(self.tok_slot, self.tok_pos, self.kv_len)
        self.step_idx.add_(1)

    def _step(self):
        if self.sk is not None:
            return self._step_splitk()
        cfg, w, c = self.cfg, self.llm.w, self.cache
        nq, nkv, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
        MB = 1 << 20
        self.in_ids.index_copy_(0, self.step_idx, self.ids.unsqueeze(0))
        ops.embed_gather(w.embed, self.ids, out=self.seq)
        for l in range(cfg.layers):
            lw = w.und[l]
            # window 1 (input norm): start pulling the QKV weights
            qkv_w, o_w, down_w = self.dec[l]
            self._prefetch([(qkv_w.wp, 0, self.pf_w1 * MB)])
            if self.fuse_norm:
                self._join()
                ops.gemm(self.seq, qkv_w, out=self.qkv, norm_w=lw.in_norm, norm_eps=cfg.rms_eps)
            else:
                ops.rmsnorm(self.seq, lw.in_norm, cfg.rms_eps, out=self.x)
                self._join()
                ops.gemm(self.x, qkv_w, out=self.qkv)
            # window 2 (RoPE/KV append, attention, combine): o_proj weights and the head of gate/up
            self._prefetch([(o_w.wp, 0, None), (lw.gate_up.wp, 0, self.pf_w2 * MB)])
            if self.fuse_attn:   # q/k norm + RoPE + KV append inside the attention kernel: one launch less per layer
                ops.attn_decode_fused(self.qkv, self.o, c.slabs[l], self.cu_q, self.kv_len, self.tok_pos, nq, nkv, hd,
                                      cfg.rms_eps, lw.q_norm, lw.k_norm, w.cos, w.sin, self.nsplit, self.ws)
            else:
                ops.qkv_post(self.qkv, self.q, c.slabs[l], self.tok_seg, self.tok_slot, self.tok_pos, nq, nkv, hd,
                             cfg.rms_eps, lw.q_norm, lw.k_norm, cos_tab=w.cos, sin_tab=w.sin)
                ops.attention(self.q, self.o, c.slabs[l], self.cu_q, self.kv_len, nq, nkv, hd, True, 1, self.max_kv,
                              self.nsplit, self.ws)
            self._join()
            ops.gemm(self.o, o_w, out=self.seq, residual=self.seq)
            if self.fuse_norm:
                ops.gemm(self.seq, lw.gate_up, out=self.act, norm_w=lw.post_norm, norm_eps=cfg.rms_eps)
            else:
                # window 3 (post-attention norm): a little more of gate/up
                self._prefetch([(lw.gate_up.wp, self.pf_w2 * MB, self.pf_w3 * MB)])
                ops.rmsnorm(self.seq, lw.post_norm, cfg.rms_eps, out=self.x)
                self._join()
                ops.gemm(self.x, lw.gate_up, out=self.act)
            ops.gemm(self.act, down_w, out=self.seq, residual=self.seq)
        if self.fuse_norm:
            ops.gemm(self.seq, w.lm_head, out=self.logits, norm_w=w.norm, norm_eps=cfg.rms_eps)
        else:
            ops.rmsnorm(self.seq, w.norm, cfg.rms_eps, out=self.hn)
            ops.gemm(self.hn, w.lm_head, out=self.logits)
        if self.do_sample:
            ops.sample(self.logits, self.temperature, self.seed, step=self.step_idx, out=self.ids)
        else:
            ops.argmax(self.logits, out=self.ids)
        self.pred_ids.index_copy_(0, self.step_idx, self.ids.unsqueeze(0))
        ops.decode_advance(self.tok_slot, self.tok_pos, self.kv_len)
        self.step_idx.add_(1)

    def _capture(self):
        # the captured step appends at slot = lens0 and bumps the counters; warm up on a side
        # stream first (lazy module loading), then restore the counters so capture sees a clean state
        saved = [t.clone() for t in (self.ids, self.tok_slot, self.tok_pos, self.kv_len, self.step_idx)]
        s = torch.cuda.Stream()
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
    def set_slot(self, b, token, kv_len, pos):
        """Sample b continues from `token` with `kv_len` tokens already in its cache segment and rope position `pos`."""
        self.ids[b:b + 1].fill_(int(token))
        self.tok_slot[b:b + 1].fill_(int(kv_len))
        self.kv_len[b:b + 1].fill_(int(kv_len) + 1)
        self.tok_pos[b:b + 1].fill_(int(pos))

    def rewind_outputs(self):
        """Start writing in_ids / pred_ids at row 0 again (the caller has harvested the previous rows)."""
        self.step_idx.zero_()
        self.steps_done = 0

    def commit(self, steps=None):
        """Make the cache's host-side lengths reflect `steps` decoded tokens."""
        steps = self.steps_done if steps is None else steps
        self.cache.lens = [l + steps for l in self.lens0]
