"""CPU checks of the fp8 weight format restatement (oracle/fp8.py): power-of-two channel scales, e4m3fn codes,
exactness of the dequantised weights in bf16.  No reference counterpart exists (see oracle/fp8.py header)."""
import numpy as np
import torch

from oracle import fp8


def test_pow2_scale_properties():
    rng = np.random.default_rng(0)
    amax = np.concatenate([np.exp(rng.uniform(-20, 20, 4000)).astype(np.float32),
                           np.float32(448.0) * np.ldexp(np.float32(1.0), np.arange(-12, 12)).astype(np.float32),
                           np.nextafter(np.float32(448.0), np.float32(1e9), dtype=np.float32)[None],
                           np.float32([0.0, 1e-30, 3.0e38 / 4])])
    s = fp8.pow2_scale(amax)
    m, e = np.frexp(s)
    assert np.all(m == 0.5), "scales are powers of two"
    pos = amax > 0
    assert np.all(448.0 * s[pos].astype(np.float64) >= amax[pos]), "no channel maximum overflows e4m3"
    assert np.all(448.0 * s[pos].astype(np.float64) / 2 < amax[pos]), "and the scale is the smallest such power of two"
    assert s[amax == 0] == 1.0


def test_all_codes_roundtrip():
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(torch.float8_e4m3fn).to(torch.float32)
    finite = ~torch.isnan(vals)
    w = vals[finite].to(torch.bfloat16)
    assert torch.equal(w.to(torch.float32), vals[finite]), "every e4m3 value is exact in bf16"
    row = torch.cat([w, torch.tensor([448.0], dtype=torch.bfloat16)])[None, :]   # channel max 448 -> scale 1
    q, scale, deq = fp8.quantize_rows(row)
    assert float(scale[0]) == 1.0
    assert torch.equal(deq, row)
    back = q[0, :-1].view(torch.float8_e4m3fn).to(torch.float32)
    assert torch.equal(back, vals[finite])


def test_quantize_error_bound_and_idempotence():
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 512, generator=g) * 0.02).to(torch.bfloat16)
    w[5] = 0
    q, scale, deq = fp8.quantize_rows(w)
    wf, df = w.float(), deq.float()
    # e4m3: 3 mantissa bits -> half-ulp relative error 2^-4 in the normal range, absolute 2^-10 * scale below it
    bound = torch.maximum(wf.abs() * 2.0 ** -4, scale[:, None] * 2.0 ** -10)
    assert torch.all((wf - df).abs() <= bound)
    assert torch.all(deq[5] == 0) and float(scale[5]) == 1.0
    # W' is a fixed point (codes / scale may shift by one binade when a channel maximum rounded down to 224 * s)
    q2, scale2, deq2 = fp8.quantize_rows(deq)
    assert torch.equal(deq2, deq)
    assert torch.equal(q2.view(torch.float8_e4m3fn).float() * scale2[:, None], q.view(torch.float8_e4m3fn).float() * scale[:, None])


def test_dequantised_weights_touch_only_llm_linears(tiny_weights):
    cfg, sd, _, _ = tiny_weights
    out = fp8.dequantised_weights(sd)
    changed = {k for k in sd if not torch.equal(sd[k], out[k])}
    assert changed, "tiny weights are not already on the e4m3 grid"
    for k in changed:
        assert k.startswith("language_model.") and k.endswith(".weight")
        assert "norm" not in k and "embed" not in k
    n_lin = sum(1 for k in sd if k.startswith("language_model.") and k.endswith("_proj.weight") or k.endswith("_proj_moe_gen.weight"))
    assert len(changed) == n_lin + 1   # + lm_head
