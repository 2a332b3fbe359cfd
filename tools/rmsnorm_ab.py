#!/usr/bin/env python
"""A/B of umv_rmsnorm_bf16's two kernels (one wave per row / one workgroup per row; UMV_RMSNORM_ROWBLOCK_MAX = rows up to which the second is
used; needs a build with that knob, see profiles/r05_rmsnorm_rowblock_ab.txt) at the row counts of the legs: time per call and a sha of the output."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimedvl_amd import ops
H = 3584
g = torch.Generator().manual_seed(5)
w = torch.randn(H, generator=g).to(torch.bfloat16).cuda()
wg = torch.randn(H, generator=g).to(torch.bfloat16).cuda()
for T in (64, 272, 1040, 2064, 8208, 12288):
    x = torch.randn(T, H, generator=g).to(torch.bfloat16).cuda()
    expert = (torch.arange(T) % 129 != 0).int().cuda()
    out = torch.empty_like(x)
    for with_expert in (False, True):
        f = (lambda: ops.rmsnorm(x, w, 1e-6, out=out, w_gen=wg, expert=expert)) if with_expert else (lambda: ops.rmsnorm(x, w, 1e-6, out=out))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        h = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
        print(f"T={T:6d} expert={int(with_expert)}  {e0.elapsed_time(e1) * 20:7.2f} us  sha {h}")
