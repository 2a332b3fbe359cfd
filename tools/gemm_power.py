#!/usr/bin/env python
"""Sustained tiled-GEMM loop with rocm-smi samples of the shader clock and the package power next to the TFLOP/s
(the chip clocks to its power budget: a denser loop can be paid back in clock).  UMV_GEMM_TILE=<cfg> picks the tile,
SHAPE=M,N,K[,swiglu], SECONDS=<loop length>."""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402

v = os.environ.get("SHAPE", "8192,8192,8192").split(",")
M, N, K, sw = int(v[0]), int(v[1]), int(v[2]), len(v) > 3
secs = float(os.environ.get("SECONDS", "4"))
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
if sw:
    lin = ops.PackedLinear.from_gate_up(torch.randn(N // 2, K, device="cuda").to(torch.bfloat16) * 0.02,
                                        torch.randn(N // 2, K, device="cuda").to(torch.bfloat16) * 0.02)
else:
    lin = ops.PackedLinear.from_weight(torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02)
out = torch.empty(M, N // 2 if sw else N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(x, lin, out=out)
torch.cuda.synchronize()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
            samples.append((int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:  # noqa: BLE001
            samples.append((-2, -2.0))
        time.sleep(0.2)


th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        ops.gemm(x, lin, out=out)
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
us = e0.elapsed_time(e1) * 1e3 / n
mid = samples[len(samples) // 3:] or samples
clk = sorted(s[0] for s in mid)[len(mid) // 2] if mid else -1
pw = sorted(s[1] for s in mid)[len(mid) // 2] if mid else -1
print(f"tile={os.environ.get('UMV_GEMM_TILE', 'auto'):>5s} M={M} N={N} K={K} {us:9.1f} us {2.0 * M * N * K / us / 1e6:8.1f} TF/s  "
      f"sclk median {clk} MHz  power median {pw} W  ({len(samples)} samples)", flush=True)
