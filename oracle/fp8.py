"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

CPU restatement of the weight-only fp8 format of BASELINE.json configs[4] ("fp8 weights").

PARITY UNPINNED BY THE REFERENCE: uni-medical/UniMedVL has no fp8 path (its linears are bf16 nn.Linear,
codes/modeling/unimedvl/qwen2_navit.py:541-562, codes/modeling/qwen2/modeling_qwen2.py:229-235), so there is no
reference behaviour to match.  The format is this project's (include/unimedvl_hip.h, "fp8 weights") and this
restatement pins it against torch's own OCP float8_e4m3fn conversion:

    s[n]  = the smallest power of two with 448 * s >= max_k |W[n, k]|      (448 = largest finite e4m3 value)
    q     = round-to-nearest-even e4m3fn(W / s)                              (W / s is exact: s is a power of two)
    W'    = q * s                                                            (exact in bf16)

The model with fp8 weights IS the bf16 model run on W' (the HIP decode GEMM converts q * s in registers and feeds the
same bf16 MFMAs), so the oracle for an fp8 run is OracleBagel on `dequantised_weights(...)`.
"""
import numpy as np
import torch

E4M3_MAX = 448.0


def pow2_scale(amax: np.ndarray) -> np.ndarray:
    """smallest 2^e with 448 * 2^e >= amax (1.0 for an all-zero channel); 448 = 0.875 * 2^9"""
    amax = np.asarray(amax, dtype=np.float32)
    ma, ea = np.frexp(amax)   # amax = ma * 2^ea, ma in [0.5, 1)
    e = np.where(ma <= np.float32(0.875), ea - 9, ea - 8)
    s = np.ldexp(np.float32(1.0), e).astype(np.float32)
    return np.where(amax > 0, s, np.float32(1.0)).astype(np.float32)


def quantize_rows(w: torch.Tensor):
    """w [N, K] (bf16 or fp32) -> (q uint8 [N, K] e4m3fn codes, scale fp32 [N], deq bf16 [N, K])"""
    wf = w.detach().to(torch.float32)
    scale = torch.from_numpy(pow2_scale(wf.abs().amax(dim=1).numpy()))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    deq = (q.to(torch.float32) * scale[:, None]).to(torch.bfloat16)
    assert torch.equal(deq.to(torch.float32), q.to(torch.float32) * scale[:, None]), "W' must be exact in bf16"
    return q.view(torch.uint8), scale, deq


def quantize_act_rows(x: torch.Tensor):
    """W8A8 mode (umv_quantize_act_fp8): activations get the weights' treatment per ROW.  Same arithmetic as quantize_rows."""
    return quantize_rows(x)


def w8a8_linear(x: torch.Tensor, w_deq: torch.Tensor, b=None) -> torch.Tensor:
    """What umv_gemm_fp8a8w computes, restated: a bf16 linear (fp32 accumulate, one rounding of the sum [+ bias]) on the
    per-row-dequantised activations and the per-channel-dequantised weights - both exactly representable in bf16."""
    xd = quantize_act_rows(x)[2]
    return torch.nn.functional.linear(xd, w_deq.to(torch.bfloat16), None if b is None else b.to(torch.bfloat16))


def unpack_image(img: torch.Tensor, N: int, K: int, swiglu_I: int = 0) -> torch.Tensor:
    """Invert the P8[nt][kt8][lane][16] image of umv_quantize_pack_weight_fp8 -> uint8 [rows, K]
    (rows = N, or [2, I, K] stacked gate/up when swiglu_I > 0)."""
    ntt, kt8 = (N + 15) // 16, (K + 63) // 64
    t = img.cpu().view(ntt, kt8, 4, 16, 2, 8)              # [nt][kt8][g][r][h][j]
    t = t.permute(0, 3, 1, 4, 2, 5).reshape(ntt * 16, kt8 * 64)   # row = nt*16 + r, k = kt8*64 + h*32 + g*8 + j
    t = t[:, :K]
    if swiglu_I:
        t = t.view(ntt // 2, 2, 16, K)
        return torch.stack([t[:, 0].reshape(-1, K)[:swiglu_I], t[:, 1].reshape(-1, K)[:swiglu_I]])
    return t[:N]


LLM_LINEAR_SUFFIXES = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def dequantised_weights(weights: dict) -> dict:
    """state dict -> the same dict with every LLM linear weight (both experts) and lm_head replaced by W'"""
    out = {}
    for name, t in weights.items():
        base = name.rsplit(".", 1)[0]
        is_lin = name.endswith(".weight") and name.startswith("language_model.") and (
            base.endswith("lm_head") or any(base.endswith(s) or base.endswith(s + "_moe_gen") for s in LLM_LINEAR_SUFFIXES))
        out[name] = quantize_rows(t)[2].to(t.dtype) if is_lin else t
    return out
