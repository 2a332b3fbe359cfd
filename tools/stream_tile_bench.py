#!/usr/bin/env python
"""gate/up (N = 37888, SwiGLU) and lm_head-like GEMMs at M rows through the tiled kernel's experimental weight-streaming
shapes (UMV_GEMM_TILE=<id> UMV_GEMM_SKINNY_MAX=16) vs the skinny kernel: us per launch, cold weights.  Usage: ... M"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

BF16 = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H, I = 3584, 18944
lins = [ops.PackedLinear.from_gate_up(torch.randn(I, H, device="cuda").to(BF16) * 0.02, torch.randn(I, H, device="cuda").to(BF16) * 0.02)
        for _ in range(3)]
x = torch.randn(M, H, device="cuda").to(BF16)
out = torch.empty(M, I, device="cuda", dtype=BF16)
for l in lins:
    ops.gemm(x, l, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(30):
    ops.gemm(x, lins[i % 3], out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
ref = torch.nn.functional.silu((x.float() @ torch.zeros(1, device="cuda")).sum() * 0 + 0) if False else None
print(f"M={M} tile={os.environ.get('UMV_GEMM_TILE', '-')} skinny_max={os.environ.get('UMV_GEMM_SKINNY_MAX', '-')}: gate_up {us:7.2f} us "
      f"{2 * I * H * 2 / us / 1e3:7.1f} GB/s  checksum {out.float().abs().sum().item():.6e}")
