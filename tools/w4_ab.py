#!/usr/bin/env python
"""4-wave AGPR tiles (gemm_w4.hip) against the 8-wave tiles they replace: bit identity (sha of outputs on a set of shapes
incl. ragged M / N / K, row-indexed x / out, SwiGLU, bias + residual + GELU) and sustained timing (SECONDS-long loops, so that
both arms run at the power-limited clock).  Each arm is its own process (the dispatch reads its env knobs once).

    python tools/w4_ab.py [sha] [time]        (default: both)
"""
import os
import subprocess as sp
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHA_CODE = f"""
import hashlib, sys, math, torch
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
from unimedvl_amd import ops
from test_kernel_branches_gpu import rnd, BF16
def sha(t): return hashlib.sha256(t.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:20]
for M, N, K in ((2048, 4608, 3584), (8192, 1152, 4304), (1000, 1152, 1160), (300, 520, 1096), (700, 3584, 96), (515, 1152, 4304), (260, 300, 40), (4099, 777, 2080), (8208, 3584, 3584)):
    x = rnd((M, K), 1); w = rnd((N, K), 2, 1 / math.sqrt(K)); b = rnd((N,), 3)
    lin = ops.PackedLinear.from_weight(w, b)
    res = rnd((M, N), 4)
    print('sha', M, N, K, sha(ops.gemm(x, lin, residual=res)))
    print('sha gelu', M, N, K, sha(ops.gemm(x, lin, act='gelu_tanh')))
    T = M + 9
    rows = torch.randperm(T, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))[:M].sort().values.to(torch.int32)
    xs = torch.zeros((T, K), dtype=BF16, device='cuda'); xs[rows.long()] = x
    o2 = torch.zeros((T, N), dtype=BF16, device='cuda')
    ops.gemm(xs, lin, out=o2, M=M, row_idx=rows)
    print('sha rows', M, N, K, sha(o2))
g, u = rnd((1024, 2048), 6, 0.02), rnd((1024, 2048), 7, 0.02)
lin = ops.PackedLinear.from_gate_up(g, u)
print('sha swiglu', sha(ops.gemm(rnd((2050, 2048), 8), lin)))
g, u = rnd((18944, 3584), 6, 0.02), rnd((18944, 3584), 7, 0.02)
lin = ops.PackedLinear.from_gate_up(g, u)
print('sha swiglu big', sha(ops.gemm(rnd((2064, 3584), 8), lin)))
"""

TIME_SHAPES = ["8192,8192,8192", "4096,4096,4096", "2064,37888,3584,swiglu", "8208,37888,3584,swiglu", "2064,3584,18944", "8208,3584,18944",
               "2064,4608,3584", "8208,4608,3584", "8192,3456,1152", "8192,4304,1152", "8192,1152,4304", "8192,1152,1152"]


def run(code_or_args, env, timeout=900):
    e = dict(os.environ)
    e.update(env)
    r = sp.run(code_or_args, capture_output=True, text=True, timeout=timeout, env=e)
    return r


def main():
    what = [a for a in sys.argv[1:] if a in ("sha", "time")] or ["sha", "time"]
    secs = os.environ.get("SECONDS", "2")
    if "sha" in what:
        for tile in ("266", "268", "384"):
            shas = {}
            for w4 in ("0", "2"):
                r = run([sys.executable, "-c", SHA_CODE], {"UMV_GEMM_TILE": tile, "UMV_GEMM_W4": w4})
                if r.returncode != 0:
                    print(f"tile {tile} w4={w4} FAILED rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}", flush=True)
                    shas[w4] = None
                    continue
                shas[w4] = [ln for ln in r.stdout.splitlines() if ln.startswith("sha")]
            if shas.get("0") and shas.get("2"):
                bad = [(a, b) for a, b in zip(shas["0"], shas["2"]) if a != b]
                print(f"tile {tile}: {len(shas['0'])} outputs, {'BIT-IDENTICAL' if not bad else f'{len(bad)} DIFFER'}", flush=True)
                for a, b in bad[:12]:
                    print("   8-wave", a, "| 4-wave", b, flush=True)
    if "time" in what:
        shapes = os.environ.get("SHAPES", ";".join(TIME_SHAPES)).split(";")
        for shape in shapes:
            for w4 in os.environ.get("W4_ARMS", "0,2").split(","):
                r = run([sys.executable, os.path.join(ROOT, "tools", "gemm_power.py")], {"UMV_GEMM_W4": w4, "SHAPE": shape, "SECONDS": secs})
                line = (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]
                print(f"w4={w4} {line}", flush=True)


if __name__ == "__main__":
    main()
