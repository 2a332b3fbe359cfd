#!/usr/bin/env python
"""Debug aid: is one prefill-attention kernel deterministic?  Runs the same call N times and counts outputs that differ from the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimedvl_amd import ops
nq, nkv, hd, B = [int(x) for x in os.environ.get('SHAPE', '28,4,128,8').split(',')]
L = int(os.environ.get('L', '1026'))
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(B * L, nq, hd, device='cuda', generator=g).to(torch.bfloat16)
slab = ops.KVSlab(B, nkv, (L + 31) // 32 * 32, hd, 'cuda')
slab.k.copy_(torch.randn(slab.k.shape, device='cuda', generator=g).to(torch.bfloat16))
slab.vt.copy_(torch.randn(slab.vt.shape, device='cuda', generator=g).to(torch.bfloat16))
cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device='cuda')
kvl = torch.full((B,), L, dtype=torch.int32, device='cuda')
outs = []
for i in range(6):
    out = torch.zeros_like(q)
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, L, L, 1, None)
    torch.cuda.synchronize()
    outs.append(out.float())
for i in range(1, 6):
    d = (outs[i] - outs[0]).abs()
    bad = (d > 0).any(-1).any(-1).nonzero().flatten().tolist()
    print(f"run {i} vs run 0: max {d.max().item():.4f}, tokens differing {len(bad)}: {bad[:16]}")
