#!/usr/bin/env python
"""Where a k-step of the 4-wave GEMM tile goes (needs a UMV_GEMM_ABLATIONS=1 build): runs cfg 94662, whose first 8 workgroups log
five s_memtime stamps per k-step for steps 16..79 - body entry, own pieces landed (vmcnt), barrier passed, last MFMA issued,
fragments of the next tile landed (lgkmcnt) - and prints the median / p90 of each interval in shader cycles.
    UMV_GEMM_TILE=94662 SHAPE=8192,8192,8192 python tools/w4_trace.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("UMV_GEMM_TILE", "94662")
import torch  # noqa: E402

from unimedvl_amd import _lib, ops  # noqa: E402

v = os.environ.get("SHAPE", "8192,8192,8192").split(",")
M, N, K = int(v[0]), int(v[1]), int(v[2])
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
lin = ops.PackedLinear.from_weight(torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
trace = torch.zeros(8 * 4 * 64 * 5 + 128 * 4 * 6, dtype=torch.int64, device="cuda")
lib = _lib.load()
a = ops.GemmArgs(x=x.data_ptr(), ldx=x.stride(0), wp=lin.wp.data_ptr(), bias=None, residual=None, ldr=0, out=out.data_ptr(), ldo=out.stride(0),
                 row_idx=None, M=M, N=N, K=K, epilogue=0, norm_w=None, norm_eps=1e-6, tile_rows=lin.th, w_scale=trace.data_ptr(), argmax_partial=None)
for _ in range(20):      # warm clocks
    ops.check(lib.umv_gemm_bf16(C.byref(a), ops._stream()), "umv_gemm_bf16")
torch.cuda.synchronize()
tl = trace.cpu()[8 * 4 * 64 * 5:].view(128, 4, 6).double()
t = trace.cpu()[:8 * 4 * 64 * 5].view(8, 4, 64, 5).double()
names = ["wait own pieces (vmcnt)", "barrier", "64 MFMAs + 16 reads + 8 pieces issued", "fragments landed (lgkmcnt)", "tail -> next body entry"]
iv = [t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], t[:, :, 1:, 0] - t[:, :, :-1, 4]]
tot = t[:, :, 1:, 0] - t[:, :, :-1, 0]
print(f"shape {M}x{N}x{K}: k-step period median {tot.median().item():.0f} p90 {tot.flatten().quantile(0.9).item():.0f} cycles (MFMA floor 1024)")
for n, d in zip(names, iv):
    f = d.flatten()
    print(f"  {n:44s} median {f.median().item():7.0f}  p90 {f.quantile(0.9).item():7.0f}  max {f.max().item():7.0f}")
for w in range(4):
    f = (t[:, w, :, 1] - t[:, w, :, 0]).flatten()
    g = (t[:, w, :, 2] - t[:, w, :, 1]).flatten()
    print(f"  wave {w}: vmcnt wait median {f.median().item():.0f}, barrier wait median {g.median().item():.0f}")

# tile-level stamps (first 64 and last 64 workgroups): entry -> first body -> loop end -> first epilogue chunk stored -> all stores accepted
ok = tl[..., 0] > 0
if ok.any():
    names2 = ["prologue (zero AGPRs, first pieces landed, first fragments)", "k loop", "epilogue chunk 0 (issue)", "epilogue chunk 1 + stores accepted", "whole tile"]
    iv2 = [tl[..., 1] - tl[..., 0], tl[..., 2] - tl[..., 1], tl[..., 3] - tl[..., 2], tl[..., 4] - tl[..., 3], tl[..., 4] - tl[..., 0]]
    for half, sl in (("first 64 workgroups", slice(0, 64)), ("last 64 workgroups", slice(64, 128))):
        if not ok[sl].any():
            continue
        print(f"tile level, {half} (shader cycles):")
        for n, d in zip(names2, iv2):
            f = d[sl][ok[sl]].flatten()
            print(f"  {n:62s} median {f.median().item():8.0f}  p90 {f.quantile(0.9).item():8.0f}  max {f.max().item():8.0f}")
