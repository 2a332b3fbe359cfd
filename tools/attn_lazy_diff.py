#!/usr/bin/env python
"""Debug aid: the lazy-softmax kernels against each other (TQ = 1, TQ = 2) and the exact kernels on one shape: which rows differ, by how much."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = f"""
import sys, torch, os
sys.path.insert(0, {ROOT!r})
from unimedvl_amd import ops
nq, nkv, hd, B = [int(x) for x in os.environ.get('SHAPE', '16,16,72,8').split(',')]
L = int(os.environ.get('L', '1024')); causal = os.environ.get('CAUSAL', '0') == '1'
g = torch.Generator(device='cuda').manual_seed(1)
q = torch.randn(B * L, nq, hd, device='cuda', generator=g).to(torch.bfloat16)
slab = ops.KVSlab(B, nkv, (L + 31) // 32 * 32, hd, 'cuda')
slab.k.copy_(torch.randn(slab.k.shape, device='cuda', generator=g).to(torch.bfloat16))
slab.vt.copy_(torch.randn(slab.vt.shape, device='cuda', generator=g).to(torch.bfloat16))
cu = torch.arange(0, (B + 1) * L, L, dtype=torch.int32, device='cuda')
kvl = torch.full((B,), L, dtype=torch.int32, device='cuda')
out = torch.zeros_like(q)
ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, causal, L, L, 1, None)
torch.save(out.cpu(), sys.argv[1])
"""
outs = {}
for name, env in (("tq1", {"UMV_ATTN_TQ": "1"}), ("tq2", {"UMV_ATTN_TQ": "2"}), ("exact", {"UMV_ATTN_LAZY": "0"})):
    path = f"/tmp/attn_{name}.pt"
    r = subprocess.run([sys.executable, "-c", CODE, path], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch
    outs[name] = torch.load(path).float()
for n in ("tq2", "exact"):
    d = (outs[n] - outs["tq1"]).abs()
    idx = (d > 0.01).nonzero()
    print(n, "vs tq1: max diff", d.max().item(), "differing", int((d > 0).sum()), "of", d.numel(), "; > 0.01:", len(idx), "first", idx[:6].tolist())
    if len(idx):
        toks = sorted(set(i[0] for i in idx.tolist()))
        heads = sorted(set(i[1] for i in idx.tolist()))
        print("   tokens:", toks[:48], "heads:", heads[:32])
        for t, h, _ in idx[:1].tolist() + idx[len(idx) // 2:len(idx) // 2 + 1].tolist():
            r = outs[n][t, h] / outs["tq1"][t, h]
            print(f"   token {t} head {h}: ratio to tq1 min {r.min().item():.4f} max {r.max().item():.4f} median {r.median().item():.4f}; |tq1| max {outs['tq1'][t, h].abs().max().item():.4f} |{n}| max {outs[n][t, h].abs().max().item():.4f}; elements differing {int((outs[n][t, h] != outs['tq1'][t, h]).sum())}")
