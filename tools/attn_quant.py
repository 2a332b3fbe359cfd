import os, sys
sys.path.insert(0, "/root/repo")
import torch
from unimedvl_amd import ops
BF16 = torch.bfloat16
def run(nseg, L=1026, nq=28, nkv=4, hd=128):
    g = torch.Generator().manual_seed(nseg)
    cap = (L + 31) // 32 * 32
    slab = ops.KVSlab(nseg, nkv, cap, hd, "cuda")
    slab.k.copy_(torch.randn(slab.k.shape, generator=g).to(BF16)); slab.vt.copy_(torch.randn(slab.vt.shape, generator=g).to(BF16))
    q = torch.randn(nseg * L, nq, hd, generator=g).to(BF16).cuda(); out = torch.zeros_like(q)
    cu = torch.arange(0, (nseg + 1) * L, L, dtype=torch.int32).cuda(); kvl = torch.full((nseg,), L, dtype=torch.int32).cuda()
    for _ in range(3): ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, L, L)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, L, L)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    wgs = 57 * nkv * nseg
    print(f"nseg={nseg:2d} workgroups={wgs:5d} = {wgs / 768:5.2f} x 768  {us:7.1f} us  {us / wgs * 768:6.1f} us per 768 workgroups")
for n in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14):
    run(n)
