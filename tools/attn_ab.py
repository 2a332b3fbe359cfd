#!/usr/bin/env python
"""A/B of the two prefill attention kernels (UMV_ATTN_SHARED=0|1, read once per process): prints a sha256 of the outputs
for a few shapes plus timings, so two runs can be compared for bit-identity.  Usage: UMV_ATTN_SHARED=0 python tools/attn_ab.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def run(nseg, Lq, Lk, nq, nkv, hd, causal, reps=5):
    g = torch.Generator().manual_seed(nseg * 1000 + Lq + Lk + hd)
    cap = (Lk + 31) // 32 * 32
    slab = ops.KVSlab(nseg, nkv, cap, hd, "cuda")
    slab.k.copy_(torch.randn(slab.k.shape, generator=g).to(BF16))
    slab.vt.copy_(torch.randn(slab.vt.shape, generator=g).to(BF16))
    q = torch.randn(nseg * Lq, nq, hd, generator=g).to(BF16).cuda()
    out = torch.zeros_like(q)
    cu = torch.arange(0, (nseg + 1) * Lq, Lq, dtype=torch.int32).cuda()
    kvl = torch.full((nseg,), Lk, dtype=torch.int32).cuda()
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, causal, Lq, Lk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, causal, Lq, Lk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 4.0 * nseg * nq * Lq * Lk * hd * (0.5 if causal and Lq == Lk else 1.0)
    h = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:16]
    print(f"nseg={nseg} Lq={Lq} Lk={Lk} nq={nq} nkv={nkv} hd={hd} causal={int(causal)}  {us:9.1f} us {flops / us / 1e6:7.1f} TF/s  sha {h}")


for args in [(8, 1024, 1024, 16, 16, 72, False), (8, 1026, 1026, 28, 4, 128, False), (8, 34, 1060, 28, 4, 128, True),
             (12, 258, 390, 28, 4, 128, False), (3, 100, 1000, 28, 4, 128, True), (2, 77, 77, 16, 16, 72, False),
             (1, 4096, 4096, 28, 4, 128, True)]:
    run(*args)
