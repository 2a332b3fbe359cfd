"""EXPERIMENTAL prefill attention on the 32x32x16 matrix instruction (experimental/csrc/attention_prefill32.hip, umv_attn_prefill32 of
experimental/lib/libunimedvl_hip_experimental.so): the hd 128 shapes of the product path - image-span prefill, ragged causal prefill on cached
context, a guided flow pass, a single segment (flash_attn_varlen_func at qwen2_navit.py:605-614) - against the fp32 flash model of
tests/test_kernel_branches_gpu.py and against the shipped kernel, plus what the kernel promises about itself: 4 and 8 waves
per workgroup give the same bits, and a segment computed alone equals the same segment inside a batch."""
import os
import subprocess
import sys

import pytest
import torch

from test_kernel_branches_gpu import BF16, _attn_ref, check_bf16, rnd  # noqa: E402

pytestmark = pytest.mark.gpu


def _xops():
    """the experimental package (experimental/): skip when its library has not been built (python -m experimental.build)"""
    from experimental import _lib as xlib
    if not xlib.available():
        pytest.skip("experimental library not built (python -m experimental.build)")
    from experimental import ops as xops
    return xops
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    from unimedvl_amd import ops as o
    return o


def _attn32(ops, *a):
    return _xops().attn_prefill32(*a)


def _run(ops, nq, nkv, hd, q_lens, k_lens, causal, seed):
    nseg = len(q_lens)
    cap = (max(k_lens) + 31) // 32 * 32
    slab = ops.KVSlab(nseg, nkv, cap, hd, "cuda")
    slab.k.fill_(1e4)          # keys / values beyond kv_len must not leak into the output
    slab.vt.fill_(float("nan"))   # ... not even as 0 x NaN
    T = sum(q_lens)
    q = rnd((T, nq, hd), seed)
    ks = [rnd((lk, nkv, hd), seed + 1 + i) for i, lk in enumerate(k_lens)]
    vs = [rnd((lk, nkv, hd), seed + 100 + i) for i, lk in enumerate(k_lens)]
    for i, lk in enumerate(k_lens):
        slab.k[i, :, :lk] = ks[i].transpose(0, 1)
        slab.vt[i, :, :, :lk] = vs[i].permute(1, 2, 0)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32).cuda()
    out = torch.zeros((T, nq, hd), dtype=BF16, device="cuda")
    _attn32(ops, q, out, slab, cu, torch.tensor(k_lens, dtype=torch.int32).cuda(), nq, nkv, hd, causal, max(q_lens), max(k_lens))
    return out, _attn_ref(q, ks, vs, q_lens, causal)


CASES = [
    ("image span 8 x 1026", [1026] * 8, [1026] * 8, False),
    ("ragged causal on context", [700, 513, 640, 1000], [700 + 1026, 513 + 40, 640, 1000 + 7], True),
    ("flow pass 12 x 258", [258] * 12, [130 + 258] * 4 + [258] * 4 + [130 + 258] * 4, False),
    ("single segment", [1026], [1026], False),
    ("short text prefill", [34] * 8, [1060] * 8, True),
    ("tiny", [5, 3], [5, 3], True),
]


@pytest.mark.parametrize("name,q_lens,k_lens,causal", CASES, ids=[c[0] for c in CASES])
def test_prefill32_vs_flash_model(ops, name, q_lens, k_lens, causal):
    out, ref = _run(ops, 28, 4, 128, q_lens, k_lens, causal, 50)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref.float()).abs().max().item() < 0.03
    check_bf16(out, ref, 4, 0.5, f"attn_prefill32 {name}")


def test_prefill32_segment_alone_equals_segment_in_batch(ops):
    """per q column the arithmetic depends on nothing but that column's own segment: same bits alone and inside a batch"""
    q_lens, k_lens = [700, 513, 640, 1000], [700 + 1026, 513 + 40, 640, 1000 + 7]
    nseg = len(q_lens)
    cap = (max(k_lens) + 31) // 32 * 32
    slab = ops.KVSlab(nseg, 4, cap, 128, "cuda")
    slab.k.copy_(rnd(tuple(slab.k.shape), 7))
    slab.vt.copy_(rnd(tuple(slab.vt.shape), 8))
    q = rnd((sum(q_lens), 28, 128), 9)
    cu = torch.tensor([0] + list(torch.tensor(q_lens).cumsum(0)), dtype=torch.int32).cuda()
    kvl = torch.tensor(k_lens, dtype=torch.int32).cuda()
    out = torch.zeros_like(q)
    _attn32(ops, q, out, slab, cu, kvl, 28, 4, 128, True, max(q_lens), max(k_lens))
    for i in (1, 3):
        s1 = ops.KVSlab(1, 4, cap, 128, "cuda")
        s1.k.copy_(slab.k[i:i + 1])
        s1.vt.copy_(slab.vt[i:i + 1])
        qi = q[int(cu[i]):int(cu[i + 1])].contiguous()
        oi = torch.zeros_like(qi)
        _attn32(ops, qi, oi, s1, torch.tensor([0, q_lens[i]], dtype=torch.int32).cuda(), kvl[i:i + 1].contiguous(), 28, 4, 128, True,
                q_lens[i], k_lens[i])
        assert torch.equal(oi, out[int(cu[i]):int(cu[i + 1])])


def test_prefill32_four_and_eight_waves_same_bits():
    _xops()          # (skips here, in the parent, when the experimental library has not been built)
    code = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from unimedvl_amd import ops
from test_attn_prefill32_gpu import _run
for q_lens, k_lens, causal in (([1026] * 8, [1026] * 8, False), ([700, 513, 640, 1000], [1726, 553, 640, 1007], True)):
    out, _ = _run(ops, 28, 4, 128, q_lens, k_lens, causal, 50)
    print('sha', hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
""" % (ROOT, os.path.join(ROOT, "tests"))
    shas = {}
    for nw in ("4", "8"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, UMV_ATTN32_NW=nw))
        assert r.returncode == 0, r.stderr[-2000:]
        shas[nw] = [ln for ln in r.stdout.splitlines() if ln.startswith("sha")]
        assert len(shas[nw]) == 2
    assert shas["4"] == shas["8"], shas
