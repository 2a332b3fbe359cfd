#!/usr/bin/env python
"""Debug aid (round 6): the paired-call lazy-softmax kernel (UMV_ATTN_VARIANT_PAIR) is not deterministic on large grids.  Runs it N times on
the bench shape with peaked scores, reports which elements differ from the per-tile kernel's (deterministic) output - token, head, q-tile
parity, lane row - and, in a UMV_ATTN_PAIR_DEBUG=1 build, the same for the bisecting forms (variant bits 8..10):
  1 = s_nop guard on the QK^T accumulators before the softmax   2 = ds_bpermute instead of v_permlane*_swap
  3 = 16 wait states between the swaps and their consumers        4 = rare path unconditional (no branch)
  5 = 32 wait states between the QK^T MFMAs of a K fragment and the next fragment's ds_read
  6 = one v_cmp for both tiles' triggers (no v_cmp_e64 -> s_or_b64 -> s_cbranch_vccz chain)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimedvl_amd import _lib as L
if os.environ.get("UMV_LIB_PATH"):        # another build of the library (e.g. the v_mov_b64-free object of profiles/r06_attn_pair_nondeterminism.txt)
    L.LIB_PATH = os.environ["UMV_LIB_PATH"]
from unimedvl_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_attn_lazy_gpu import make_case, fill_slab, HEADS

nq, nkv, hd = HEADS[128]
Lq = int(os.environ.get("L", "1026"))
B = int(os.environ.get("B", "8"))
q, ks, vs = make_case(os.environ.get("DIST", "scale8"), nq, nkv, hd, [Lq] * B, [Lq] * B, seed=5)
slab = fill_slab(ops, ks, vs, nkv, hd)
cu = torch.arange(0, (B + 1) * Lq, Lq, dtype=torch.int32).cuda()
kvl = torch.full((B,), Lq, dtype=torch.int32).cuda()
F = L.ATTN_FORCE


NWG = ((Lq * (nq // nkv) + 15) // 16 + 7) // 8 * nkv * B
POISON = int(os.environ.get("POISON", "0"))        # 1 = LDS, 2 = VGPRs, 3 = both: tools/poison.hip in front of every call
if POISON:
    import ctypes
    _pz = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libpoison.so"))
    _sink = torch.zeros(4, dtype=torch.int32, device="cuda")
def run(variant, state=None):
    out = torch.zeros_like(q)
    if POISON:
        _pz.poison(POISON, 0x7FC07FC0, ctypes.c_void_p(_sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    ops.attention(q, out, slab, cu, kvl, nq, nkv, hd, False, Lq, Lq, variant=variant, workspace=state)
    torch.cuda.synchronize()
    return out


if os.environ.get("STATE"):
    st_ref = torch.zeros(NWG * 4 * 2 * 64 * 4, dtype=torch.float32, device="cuda")
    ref0 = run(F | L.ATTN_TQ2, st_ref)
    st_ref = st_ref.view(NWG, 4, 2, 64, 4).clone()
    for rep in range(int(os.environ.get("REPS", "6"))):
        st = torch.zeros(NWG * 4 * 2 * 64 * 4, dtype=torch.float32, device="cuda")
        out = run(F | L.ATTN_TQ2 | L.ATTN_PAIR, st)
        st = st.view(NWG, 4, 2, 64, 4)
        bad = (st != st_ref) & ~(torch.isnan(st) & torch.isnan(st_ref))
        tiles = bad.flatten(3).any(-1).nonzero()
        print(f"rep {rep}: output differs: {bool((out != ref0).any())}; {len(tiles)} (wg, wave, u) tiles with a differing state word")
        for wg, wave, u in tiles[:3].tolist():
            for name, k in (("nm", 0), ("thr", 1), ("l", 2), ("o0", 3)):
                lanes = bad[wg, wave, u, :, k].nonzero().flatten().tolist()
                if lanes:
                    l0 = lanes[0]
                    row = l0 & 15
                    grp = [row, row + 16, row + 32, row + 48]
                    print(f"   wg {wg} wave {wave} u {u}: {name} differs in {len(lanes)} lanes {lanes[:20]}; row {row}: pair {[round(float(st[wg, wave, u, x, k]), 4) for x in grp]} "
                          f"per-tile {[round(float(st_ref[wg, wave, u, x, k]), 4) for x in grp]}")
    sys.exit(0)


ref = run(F | L.ATTN_TQ2)
assert torch.equal(ref, run(F | L.ATTN_TQ2)) and torch.equal(ref, run(F | L.ATTN_TQ1))
G = nq // nkv
for dbg in range(0, 7 if os.environ.get("UMV_ATTN_PAIR_DEBUG") else 1):
    v = F | L.ATTN_TQ2 | L.ATTN_PAIR | (dbg << 8)
    nbad, shown = 0, 0
    for rep in range(int(os.environ.get("REPS", "20"))):
        out = run(v)
        bad = (out != ref)
        if bad.any():
            nbad += 1
            if shown < 4 and POISON:
                print(f"  dbg {dbg} rep {rep}: non-finite outputs: {int((~torch.isfinite(out.float())).sum())}")
            rows = bad.any(-1).nonzero()                      # (token, head)
            if shown < 4:
                shown += 1
                desc = []
                for t, h in rows[:6].tolist():
                    seg, tok = divmod(t, Lq)
                    kh, hg = divmod(h, G)
                    pair = tok * G + hg                       # dense packing: pair index within (segment, kv head)
                    qt, j = divmod(pair, 16)
                    wg, r = divmod(qt, 8)
                    wave, u = divmod(r, 2)
                    nd = int(bad[t, h].sum())
                    err = float((out[t, h].float() - ref[t, h].float()).abs().max())
                    ratio = (out[t, h].float() / ref[t, h].float().clamp_min(1e-30).where(ref[t, h].float().abs() > 0.05, torch.tensor(float("nan"), device="cuda")))
                    ratio = ratio[torch.isfinite(ratio)]
                    rs = f"ratio out/ref median {float(ratio.median()):.4f} min {float(ratio.min()):.4f} max {float(ratio.max()):.4f}" if ratio.numel() else "no ratio"
                    desc.append(f"(seg {seg} tok {tok} head {h}: wg {wg} wave {wave} tile u={u} row j={j}, {nd}/128 dims, max err {err:.3g}, {rs})")
                print(f"  dbg {dbg} rep {rep}: {len(rows)} rows differ: " + " ".join(desc))
    print(f"pair dbg={dbg}: {nbad} of {os.environ.get('REPS', '20')} runs differ from the per-tile kernel")
