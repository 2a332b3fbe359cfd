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
    acc = ""
    if os.environ.get("ATTN_AB_REF", "0") != "0" and nseg * Lq * Lk * nq <= 8 * 1026 * 1026 * 28:      # error against exact fp32 attention
        G = nq // nkv
        worst, mean = 0.0, 0.0
        for sg in range(nseg):
            qs = q[sg * Lq:(sg + 1) * Lq].float()
            k = slab.k[sg, :, :Lk].float().repeat_interleave(G, 0)
            v = slab.vt[sg, :, :, :Lk].float().transpose(-1, -2).repeat_interleave(G, 0)
            sc = torch.einsum("qhd,hkd->hqk", qs, k) / hd ** 0.5
            if causal:
                qi = torch.arange(Lq, device="cuda")[:, None]
                ki = torch.arange(Lk, device="cuda")[None, :]
                sc = sc.masked_fill(ki > Lk - Lq + qi, float("-inf"))
            ref = torch.einsum("hqk,hkd->qhd", sc.softmax(-1), v)
            d = (out[sg * Lq:(sg + 1) * Lq].float() - ref).abs()
            worst, mean = max(worst, d.max().item()), mean + d.mean().item() / nseg
        acc = f"  vs fp32: max {worst:.5f} mean {mean:.6f}"
    if os.environ.get("ATTN_AB_SAVE"):
        os.makedirs(os.environ["ATTN_AB_SAVE"], exist_ok=True)
        torch.save(out.cpu(), os.path.join(os.environ["ATTN_AB_SAVE"], f"{nseg}_{Lq}_{Lk}_{nq}_{hd}_{int(causal)}.pt"))
    if os.environ.get("ATTN_AB_CMP"):
        o0 = torch.load(os.path.join(os.environ["ATTN_AB_CMP"], f"{nseg}_{Lq}_{Lk}_{nq}_{hd}_{int(causal)}.pt")).float()
        d = (out.cpu().float() - o0).abs()
        acc += f"  vs saved: max {d.max().item():.5f} equal {float((d == 0).float().mean()) * 100:.2f} %"
    print(f"nseg={nseg} Lq={Lq} Lk={Lk} nq={nq} nkv={nkv} hd={hd} causal={int(causal)}  {us:9.1f} us {flops / us / 1e6:7.1f} TF/s  sha {h}{acc}")


for args in [(8, 1024, 1024, 16, 16, 72, False), (8, 1026, 1026, 28, 4, 128, False), (8, 34, 1060, 28, 4, 128, True),
             (12, 258, 390, 28, 4, 128, False), (3, 100, 1000, 28, 4, 128, True), (2, 77, 77, 16, 16, 72, False),
             (1, 4096, 4096, 28, 4, 128, True)]:
    run(*args)
