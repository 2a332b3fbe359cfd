#!/usr/bin/env python
"""Debug aid: the per-wave kernel (UMV_ATTN_SHARED=0) against the LDS-shared prefill kernel on one shape - where and by how much do they differ?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from unimedvl_amd import ops
import os
nq, nkv, hd, B = 28, 4, 128, 2
L = int(os.environ.get('L', '70')); causal = os.environ.get('CAUSAL', '1') == '1'
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(B * L, nq, hd, device='cuda', generator=g).to(torch.bfloat16)
slab = ops.KVSlab(B, nkv, 96, hd, 'cuda')
slab.k.copy_(torch.randn(slab.k.shape, device='cuda', generator=g).to(torch.bfloat16))
slab.vt.copy_(torch.randn(slab.vt.shape, device='cuda', generator=g).to(torch.bfloat16))
cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device='cuda')
kvl = torch.full((B,), L, dtype=torch.int32, device='cuda')
out = torch.zeros_like(q)
ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, causal, L, L, 1, None)
torch.save(out.cpu(), sys.argv[1])
"""
outs = {}
for name, env in (("wave", {"UMV_ATTN_SHARED": "0"}), ("tq1", {"UMV_ATTN_TQ": "1"}), ("tq2", {"UMV_ATTN_TQ": "2"})):
    path = f"/tmp/attn_{name}.pt"
    r = subprocess.run([sys.executable, "-c", CODE, path], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch
    outs[name] = torch.load(path).float()
for n in ("tq1", "tq2"):
    d = (outs[n] - outs["wave"]).abs()
    idx = d.nonzero()
    print(n, "max diff", d.max().item(), "differing", int((d > 0).sum()), "of", d.numel(), "first", idx[:5].tolist())
    if len(idx):
        t, h, dd = idx[0].tolist()
        print("  wave", outs["wave"][t, h, dd - 2:dd + 3].tolist(), n, outs[n][t, h, dd - 2:dd + 3].tolist())
        ratio = (outs[n][t, h] / outs["wave"][t, h])
        print("  ratio of the row: min %.5f max %.5f" % (ratio.min().item(), ratio.max().item()), "elements differing in the row:", int((outs[n][t, h] != outs["wave"][t, h]).sum()))
        rows = sorted(set(i[0] for i in idx.tolist()))
        print("  rows differing:", rows[:40])
