#!/usr/bin/env python
"""Yardstick only (never on the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the prefill / flow / ViT GEMM
shapes on this box, next to umv_gemm_bf16 on the same shapes and data.  TF/s, random N(0,1) data (DVFS-limited)."""
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


for M, N, K in [(8208, 4608, 3584), (8208, 3584, 3584), (8208, 18944, 3584), (8208, 3584, 18944), (4096, 4096, 4096), (8192, 8192, 8192),
                (2048, 18944, 3584), (2048, 3584, 18944), (8192, 3456, 1152), (8192, 1152, 4304), (8192, 4304, 1152)]:
    x = torch.randn(M, K, device="cuda").to(BF16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF16)
    lin = ops.PackedLinear.from_weight(w)
    out = torch.empty(M, N, device="cuda", dtype=BF16)
    us_lib = t(lambda: torch.matmul(x, w.t(), out=out)) if not os.environ.get("UMV_NO_LIB") else float("nan")
    us_umv = t(lambda: ops.gemm(x, lin, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:6d} K={K:6d}  torch.matmul {us_lib:8.1f} us {fl / us_lib / 1e6:7.1f} TF/s   umv_gemm_bf16 {us_umv:8.1f} us {fl / us_umv / 1e6:7.1f} TF/s")
