"""A/B of tiled-GEMM tile configurations (UMV_GEMM_TILE=<cfg>, tuning only): time + max error against torch fp32 matmul + sha."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


shapes = [(8192, 1152, 4304), (8192, 1152, 1152), (2048, 4608, 3584), (2048, 3584, 18944), (2048, 3584, 3584), (3072, 4608, 3584), (3072, 3584, 18944),
          (1024, 3584, 3584), (1000, 1152, 1160)]
if os.environ.get("SHAPES"):
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["SHAPES"].split(";")]
g = torch.Generator(device="cuda").manual_seed(1)
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda", generator=g).to(BF16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(BF16)
    b = torch.randn(N, device="cuda", generator=g).to(BF16)
    lin = ops.PackedLinear.from_weight(w, b)
    res = torch.randn(M, N, device="cuda", generator=g).to(BF16)
    out = torch.empty(M, N, device="cuda", dtype=BF16)
    us = t(lambda: ops.gemm(x, lin, out=out, residual=res))
    ref = ((x.float() @ w.float().t() + b.float()).to(BF16).float() + res.float()).to(BF16)
    err = (out.float() - ref.float()).abs().max().item()
    exact = (out == ref).float().mean().item()
    h = hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"M={M:5d} N={N:6d} K={K:6d} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  max err {err:.4f} exact {exact:.4f} sha {h}", flush=True)
