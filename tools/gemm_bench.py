#!/usr/bin/env python
"""Micro-benchmark of umv_gemm_bf16 on the model's GEMM shapes (TFLOP/s, random data).
UMV_GEMM_TILE=<256|128|129|130|64|...> forces a tile configuration.  `--flow` sweeps the
guided-flow shapes (M = 3 x 1024 latent rows) instead of the prefill / ViT ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402


FP8 = "--fp8" in sys.argv   # W8A8 on the fp8 matrix instruction (the time includes the per-row activation quantisation)


def bench(M, N, K, swiglu=False, reps=20):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    if swiglu:
        mk = ops.PackedLinear.from_gate_up_fp8 if FP8 else ops.PackedLinear.from_gate_up
        lin = mk(torch.randn(N // 2, K, device="cuda").to(torch.bfloat16) * 0.02, torch.randn(N // 2, K, device="cuda").to(torch.bfloat16) * 0.02)
    else:
        mk = ops.PackedLinear.from_weight_fp8 if FP8 else ops.PackedLinear.from_weight
        lin = mk(torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02)
    if FP8:
        lin.enable_fp8_mfma()
    out = torch.empty(M, N // 2 if swiglu else N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(x, lin, out=out, act8=FP8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(x, lin, out=out, act8=FP8)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


PREFILL = [(1024, 4608, 3584, False), (1024, 3584, 3584, False), (1024, 37888, 3584, True), (1024, 3584, 18944, False),
           (8208, 4608, 3584, False), (8208, 3584, 3584, False), (8208, 37888, 3584, True), (8208, 3584, 18944, False),
           (8192, 3456, 1152, False), (8192, 1152, 1152, False), (8192, 4304, 1152, False), (8192, 1152, 4304, False),
           (4096, 4096, 4096, False), (8192, 8192, 8192, False)]
FLOW = [(3072, 4608, 3584, False), (3072, 3584, 3584, False), (3072, 37888, 3584, True), (3072, 3584, 18944, False),
        (2048, 3584, 18944, False), (2048, 4608, 3584, False), (2048, 3584, 3584, False)]

if __name__ == "__main__":
    tag = os.environ.get("UMV_GEMM8_TILE" if FP8 else "UMV_GEMM_TILE", "auto")
    shapes = FLOW if "--flow" in sys.argv else PREFILL
    if os.environ.get("SHAPE"):   # SHAPE=M,N,K[,swiglu]: one shape only (counter collection)
        v = os.environ["SHAPE"].split(",")
        shapes = [(int(v[0]), int(v[1]), int(v[2]), len(v) > 3)]
    for M, N, K, sw in shapes:
        us, tf = bench(M, N, K, sw)
        print(f"tile={tag:>4s} M={M:5d} N={N:6d} K={K:6d} {'swiglu' if sw else '      '} {us:9.1f} us {tf:8.1f} TF/s", flush=True)
