#!/usr/bin/env python
"""The four GEMMs of one SigLIP layer at 8 x 1024 patches WITH their epilogues (bias; bias + GELU-tanh; bias + residual), timed
one by one on random data, next to the same shapes without an epilogue: what the epilogue costs at K = 1152.
UMV_GEMM_TILE forces a tile configuration."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def t(fn, reps=20):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


M = int(os.environ.get("ROWS", "8192"))
tot = [0.0, 0.0]
for name, N, K, act, res in [("qkv", 3456, 1152, None, False), ("out", 1152, 1152, None, True), ("fc1", 4304, 1152, "gelu_tanh", False),
                             ("fc2", 1152, 4304, None, True)]:
    x = torch.randn(M, K, device="cuda").to(BF16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
    b = torch.randn(N, device="cuda").to(BF16)
    lin, lin0 = ops.PackedLinear.from_weight(w, b), ops.PackedLinear.from_weight(w)
    out = torch.empty(M, N, device="cuda", dtype=BF16)
    h = torch.randn(M, N, device="cuda").to(BF16)
    us0 = t(lambda: ops.gemm(x, lin0, out=out))
    us1 = t(lambda: ops.gemm(x, lin, out=out, act=act, residual=h if res else None))
    fl = 2.0 * M * N * K
    tot[0] += us0
    tot[1] += us1
    print(f"{name} M={M} N={N:5d} K={K:5d}  plain {us0:7.1f} us {fl / us0 / 1e6:7.1f} TF/s   with epilogue {us1:7.1f} us {fl / us1 / 1e6:7.1f} TF/s")
print(f"layer GEMMs: plain {tot[0]:.1f} us, with epilogues {tot[1]:.1f} us")
