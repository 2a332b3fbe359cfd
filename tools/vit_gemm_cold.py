#!/usr/bin/env python
"""ViT-shaped tiled GEMMs with WARM weights (the same packed image every launch: resident in L2 / the memory-side cache) against COLD
weights (rotating through > 600 MB of copies, as the 26 layers of the tower do), and cold + an in-stream prefetch of the weights
(experimental umv_prefetch) right before the GEMM.  Answers: is the in-tower GEMM time (r04_vit_kernel_stats_by_grid.csv: qkv 82.6,
fc1 107, out-proj / fc2 34 / 103 us) against 69 / 88 / 26 / 74 us in a same-weight loop a cold-weight effect?
Usage: python tools/vit_gemm_cold.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    try:
        from experimental import ops as xops
        xops._lib.load()
    except Exception as e:  # noqa: BLE001
        xops = None
        print("no experimental library (prefetch arm skipped):", e)
    shapes = [("qkv", 3456, 1152), ("out", 1152, 1152), ("fc1", 4304, 1152), ("fc2", 1152, 4304),
              ("llm_qkv", 4608, 3584), ("llm_o", 3584, 3584), ("llm_down", 3584, 18944)]
    for name, N, K in shapes:
        nbytes = N * K * 2
        ncopies = int(700e6 // nbytes) + 1
        if name.startswith("llm"):
            ncopies = min(ncopies, 12)
        lins = [ops.PackedLinear.from_weight(torch.randn(N, K, device="cuda").to(BF16) * 0.02, torch.zeros(N, device="cuda", dtype=BF16))
                for _ in range(ncopies)]
        x = torch.randn(M, K, device="cuda").to(BF16)
        out = torch.empty(M, N, device="cuda", dtype=BF16)
        for lin in lins[:3]:
            ops.gemm(x, lin, out=out)
        reps = max(3 * ncopies, 200)
        # sustained warm-up so that all arms run at the clocks of a loaded chip
        timed(lambda i: ops.gemm(x, lins[0], out=out), 300)
        warm = timed(lambda i: ops.gemm(x, lins[0], out=out), reps)
        cold = timed(lambda i: ops.gemm(x, lins[i % ncopies], out=out), reps)
        line = f"{name:8s} M={M} N={N:5d} K={K:5d} copies={ncopies:3d}  warm {warm:7.1f} us   cold {cold:7.1f} us ({cold / warm - 1:+.1%})"
        if xops is not None:
            def pf(i):
                xops.prefetch(lins[i % ncopies].wp, blocks=256)
                ops.gemm(x, lins[i % ncopies], out=out)
            cp = timed(pf, reps)
            po = timed(lambda i: xops.prefetch(lins[i % ncopies].wp, blocks=256), reps)
            line += f"   cold+prefetch {cp:7.1f} us (prefetch alone {po:5.1f})"
        print(line, flush=True)
        del lins


if __name__ == "__main__":
    main()
