#!/usr/bin/env python
"""Host-side cost of one op through ctypes (enqueue rate, no sync in the loop): what bounds single-request prefill latency."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

x = torch.randn(8, 3584, device="cuda").to(torch.bfloat16)
w = torch.ones(3584, device="cuda", dtype=torch.bfloat16)
out = torch.empty_like(x)
lin = ops.PackedLinear.from_weight(torch.randn(256, 3584, device="cuda").to(torch.bfloat16))
y = torch.empty(8, 256, device="cuda", dtype=torch.bfloat16)
for name, fn in (("rmsnorm", lambda: ops.rmsnorm(x, w, 1e-6, out=out)), ("gemm", lambda: ops.gemm(x, lin, out=y)),
                 ("_stream", lambda: ops._stream()), ("torch add_", lambda: out.add_(1))):
    fn()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:10s} {1e6 * (t1 - t0) / n:6.2f} us per call (host side)")
