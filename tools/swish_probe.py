import sys, torch
sys.path.insert(0, '/root/repo')
from unimedvl_amd import _lib
from unimedvl_amd.vae import _stream
BF16 = torch.bfloat16
lib = _lib.load()
C, hw = 2048, 4
bits = torch.arange(65536, dtype=torch.int32)
allv = bits.to(torch.int16).view(BF16)
x = torch.tensor([1.0, -1.0, 1.0, -1.0]).view(1, hw, 1).expand(1, hw, C).contiguous().to(BF16).cuda()
ws = torch.empty(lib.umv_groupnorm_workspace_bytes(1, hw) // 4 + 16, dtype=torch.float32, device="cuda")
gamma = torch.zeros(C, dtype=BF16, device="cuda")
out = torch.empty_like(x)
gots = []
for i in range(65536 // C):
    beta = allv[i * C:(i + 1) * C].cuda()
    _lib.check(lib.umv_groupnorm_nhwc_bf16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), 1, hw, C, 1e-6, 1, _stream()), "gn")
    torch.cuda.synchronize()
    gots.append(out[0, 0].cpu().clone())
got = torch.cat(gots)
v = allv
fin = torch.isfinite(v.float())
ref = v * torch.sigmoid(v)
gi, ri = got.view(torch.int16).int(), ref.view(torch.int16).int()
ne = (gi != ri) & fin & ~((got.float() == 0) & (ref.float() == 0))
print("differ:", int(ne.sum()))
vf = v.float()
ulp1 = ne & ((gi - ri).abs() == 1)
print("  exactly 1 bf16 step apart:", int(ulp1.sum()))
rest = ne & ~ulp1
print("  others:", int(rest.sum()))
if rest.any():
    rv = vf[rest]
    print("   v range of others: min", float(rv.min()), "max", float(rv.max()), " |v| min", float(rv.abs().min()))
    idx = rest.nonzero().flatten()[:12]
    for k in idx.tolist():
        print("   v=%g got=%g ref=%g" % (float(vf[k]), float(got[k].float()), float(ref[k].float())))
# where do the 1-step ones live
if ulp1.any():
    a = vf[ulp1].abs()
    for lo, hi in ((0, 1e-30), (1e-30, 1e-3), (1e-3, 0.1), (0.1, 1), (1, 10), (10, 100), (100, 1e38)):
        print("   1-step, |v| in [%g,%g): %d" % (lo, hi, int(((a >= lo) & (a < hi)).sum())))
