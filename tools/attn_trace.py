#!/usr/bin/env python
"""Where a 32-key stage of the prefill attention goes (needs a UMV_ATTN_TRACE=1 build): the first 8 workgroups of (kv head 0, segment 0)
log five s_memtime stamps per stage for stages 4..19 - top of the stage, barrier passed (+ next stage's DMA issued), QK^T MFMAs issued,
softmax done, PV MFMAs issued - and the tool prints the median / p90 of each interval in shader cycles, next to the kernel's wall time.
    CASE=llm|vit python tools/attn_trace.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

case = os.environ.get("CASE", "llm")
nq, nkv, hd, L = (28, 4, 128, 1026) if case == "llm" else (16, 16, 72, 1024)
B = int(os.environ.get("B", "8"))
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(B * L, nq, hd, device="cuda", generator=g).to(torch.bfloat16)
slab = ops.KVSlab(B, nkv, ((L + 31) // 32) * 32, hd, "cuda")
slab.k.copy_(torch.randn(slab.k.shape, device="cuda", generator=g).to(torch.bfloat16))
slab.vt.copy_(torch.randn(slab.vt.shape, device="cuda", generator=g).to(torch.bfloat16))
cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device="cuda")
kvl = torch.full((B,), L, dtype=torch.int32, device="cuda")
out = torch.zeros_like(q)
trace = torch.zeros(8 * 4 * 16 * 5, dtype=torch.int64, device="cuda")
for _ in range(5):
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, L, L, 1, trace)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, L, L, 1, trace)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
fl = 4.0 * B * L * L * nq * hd
print(f"{case}: {B} x {L} tokens, {nq}/{nkv} heads of {hd}: {us:.1f} us per call, {fl / us / 1e6:.0f} TF/s")
t = trace.cpu().view(8, 4, 16, 5).double()
if (t > 0).any():
    names = ["wait own DMA pieces + barrier + issue next stage", "K fragment reads + QK^T MFMAs issued", "softmax (both q-tiles)", "V^T reads + PV MFMAs issued", "end -> next stage top"]
    iv = [t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], t[:, :, 1:, 0] - t[:, :, :-1, 4]]
    tot = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    print(f"  stage period median {tot.median().item():.0f} p90 {tot.flatten().quantile(0.9).item():.0f} cycles")
    for n, d in zip(names, iv):
        f = d.flatten()
        print(f"  {n:52s} median {f.median().item():7.0f}  p90 {f.quantile(0.9).item():7.0f}  max {f.max().item():7.0f}")
else:
    print("  (no trace: not a UMV_ATTN_TRACE build)")
