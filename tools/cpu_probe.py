#!/usr/bin/env python
"""Host-CPU probe for the GPU box: bf16 F.linear throughput of the oracle's shapes at a few thread counts (sizes the
full-width oracle tests and bench.py's cpu_baseline sample), plus lscpu."""
import subprocess
import time

import torch

print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout[:1500])
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    for M in (8, 32, 2064, 8208):
        x = torch.randn(M, 3584).to(torch.bfloat16)
        w = torch.randn(18944, 3584).to(torch.bfloat16)
        torch.nn.functional.linear(x, w)
        t = time.time()
        n = 3 if M < 100 else 1
        for _ in range(n):
            torch.nn.functional.linear(x, w)
        dt = (time.time() - t) / n
        print(f"threads {nt:4d} M {M:5d}: {dt * 1e3:9.1f} ms  {2 * M * 3584 * 18944 / dt / 1e12:7.3f} TF/s", flush=True)
x = torch.randn(1, 128, 256, 256).to(torch.bfloat16)
w = torch.randn(128, 128, 3, 3).to(torch.bfloat16)
for nt in (32, 128):
    torch.set_num_threads(nt)
    torch.nn.functional.conv2d(x, w, padding=1)
    t = time.time()
    torch.nn.functional.conv2d(x, w, padding=1)
    dt = time.time() - t
    print(f"threads {nt}: conv 128->128 256x256 {dt * 1e3:.1f} ms {2 * 128 * 128 * 9 * 65536 / dt / 1e12:.3f} TF/s")
