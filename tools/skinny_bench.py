#!/usr/bin/env python
"""Micro-benchmark of the decode (M = batch) weight-streaming GEMMs: GB/s per shape.
cold  = rotate through enough distinct weight copies that nothing is cache resident (> 256 MiB MALL);
warm  = the same weight every launch (fits the 256 MiB memory-side cache; does not fit the 32 MiB of L2).
Usage: python tools/skinny_bench.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402
from experimental import ops as xops  # noqa: E402  (DEC=1: the experimental persistent decode GEMM)

H, I, QKV, V = 3584, 18944, 4608, 152064
BF16 = torch.bfloat16


def timed(fn, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    exact = os.environ.get("EXACT", "1") != "0" and int(os.environ.get("SPLIT", "0")) <= 1   # split-K uses the 16-row tiles
    fp8 = os.environ.get("FP8", "0") != "0"
    norm = os.environ.get("NORM", "0") != "0"   # fused RMSNorm prologue (K <= 4096 shapes only)
    dec = os.environ.get("DEC", "0") != "0"     # persistent decode GEMM (umv_gemm_decode) on the decode image
    shapes = [("qkv", QKV, H, False), ("o", H, H, False), ("gate_up", 2 * I, H, True), ("down", H, I, False)]
    for name, N, K, swiglu in shapes:
        nbytes = N * K * (1 if fp8 else 2)
        ncopies = max(3, int(600e6 // nbytes) + 1)
        lins = []
        for c in range(ncopies):
            if swiglu:
                mk = ops.PackedLinear.from_gate_up_fp8 if fp8 else ops.PackedLinear.from_gate_up
                lin = mk(torch.randn(N // 2, K, device="cuda").to(BF16) * 0.02, torch.randn(N // 2, K, device="cuda").to(BF16) * 0.02)
            else:
                mk = ops.PackedLinear.from_weight_fp8 if fp8 else ops.PackedLinear.from_weight
                lin = mk(torch.randn(N, K, device="cuda").to(BF16) * 0.02)
                if exact:
                    lin = lin.for_decode()
            if dec:
                lin = xops.DecodeLinear(lin)
            elif fp8:
                lin.wp = None   # only the e4m3 image is streamed at M <= 64
            lins.append(lin)
        x = torch.randn(B, K, device="cuda").to(BF16)
        out = torch.empty(B, N // 2 if swiglu else N, device="cuda", dtype=BF16)
        nw = torch.ones(K, device="cuda", dtype=BF16) if (norm and K <= 4096) else None
        mm = xops.gemm_decode if dec else ops.gemm
        split = int(os.environ.get("SPLIT", "0"))   # split-K mode (fp32 partials, finished by the consumer kernel)
        if split > 1 and not swiglu:
            part = torch.empty(split, B, N, device="cuda", dtype=torch.float32)
            mm = lambda x, lin, out=None, norm_w=None: ops.gemm_splitk(x, lin, part, split)   # noqa: E731
        for lin in lins:
            mm(x, lin, out=out, norm_w=nw)
        cold = timed(lambda i: mm(x, lins[i % ncopies], out=out, norm_w=nw), 10 * ncopies)
        warm = timed(lambda i: mm(x, lins[0], out=out, norm_w=nw), 50)
        th = lins[0].layout.th if dec else lins[0].th
        print(f"{name:8s} N={N:6d} K={K:6d} th={th:2d} copies={ncopies:3d}  cold {cold:7.2f} us {nbytes / cold / 1e3:7.1f} GB/s"
              f"   warm {warm:7.2f} us {nbytes / warm / 1e3:7.1f} GB/s   ideal@6.3TB/s {nbytes / 6.3e6:6.2f} us")


if __name__ == "__main__":
    main()
