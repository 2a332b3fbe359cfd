import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F
from test_kernel_branches_gpu import rnd, _mm, ulp_diff, BF16
from unimedvl_amd import ops
for (M, N, K, epi) in [(8208, 3584, 18944, "residual"), (8192, 4304, 1152, "gelu")]:
    x = rnd((M, K), 1)
    w, b = rnd((N, K), 2, 1 / math.sqrt(K)), rnd((N,), 3)
    base = _mm(x, w)
    if epi == "gelu":
        out = ops.gemm(x, ops.PackedLinear.from_weight(w, b), act="gelu_tanh")
        pre = (base + b.float())
        ref = F.gelu(pre.to(BF16), approximate="tanh")
        res = None
    else:
        res = rnd((M, N), 4)
        out = ops.gemm(x, ops.PackedLinear.from_weight(w), residual=res)
        pre = base
        ref = res + base.to(BF16)
    d = ulp_diff(out, ref)
    absd = (out.float() - ref.float()).abs()
    bad = (d > 1) & (absd > ref.float().abs().max() * 2 ** -8)
    idx = bad.nonzero()
    print(epi, M, N, K, "bad", idx.shape[0])
    for i, j in idx[:12].tolist():
        print("  ", i, j, "got", out[i, j].item(), "ref", ref[i, j].item(), "pre(fp32)", pre[i, j].item(), "pre bf16", pre[i, j].to(BF16).item(),
              "res", None if res is None else res[i, j].item(), "ulp", d[i, j].item())
    # fp64 truth for those elements
    for i, j in idx[:12].tolist():
        t = (x[i].double() * w[j].double()).sum().item()
        print("     fp64 acc", t, "+bias" if epi == "gelu" else "", (t + b[j].double().item()) if epi == "gelu" else "")
