#!/usr/bin/env python
"""Decode layer engine (experimental/engine.py, experimental/csrc/decode_engine.hip) against the kernel chain it replaces:
bit-level comparison of every intermediate of one decoder layer after its attention, then timing of N_LAYERS
distinct layers' worth of weights back to back (nothing cache resident) under a HIP graph.

    python tools/engine_bench.py [batch] [stage]
stage: gu (gate/up only) | ogu (o_proj -> gate/up) | mlp (o_proj -> gate/up -> down partials) | full (... -> reduce -> qkv)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from unimedvl_amd import ops  # noqa: E402
from experimental import engine  # noqa: E402

H, I, QKV = 3584, 18944, 4608
BF16 = torch.bfloat16
EPS = 1e-6


class LW:
    pass


def make_layer(dev, gen):
    def rn(*shape, std=0.02):
        return (torch.randn(*shape, device=dev, generator=gen) * std).to(BF16)
    lw = LW()
    lw.o = ops.PackedLinear.from_weight(rn(H, H))
    lw.gate_up = ops.PackedLinear.from_gate_up(rn(I, H), rn(I, H))
    lw.down = ops.PackedLinear.from_weight(rn(H, I))
    lw.qkv = ops.PackedLinear.from_weight(rn(QKV, H), rn(QKV))
    lw.post_norm = (1.0 + 0.1 * torch.randn(H, device=dev, generator=gen)).to(BF16)
    lw.in_norm = (1.0 + 0.1 * torch.randn(H, device=dev, generator=gen)).to(BF16)
    return lw


class Bufs:
    def __init__(self, B, dev, gen):
        self.attn = torch.randn(B, H, device=dev, generator=gen).to(BF16)
        self.seq0 = torch.randn(B, H, device=dev, generator=gen).to(BF16)
        self.seq = self.seq0.clone()
        self.x = torch.empty(B, H, dtype=BF16, device=dev)
        self.act = torch.empty(B, I, dtype=BF16, device=dev)
        self.p_h = torch.empty(8, B, H, dtype=torch.float32, device=dev)
        self.qkv = torch.empty(B, QKV, dtype=BF16, device=dev)
        self.ss = torch.zeros(2, H // 16, 8, dtype=torch.float32, device=dev)


def classic(lw, b, stage, sd=8):
    """the kernel chain of decode.py::_step for the same ops (o_proj unsplit, down split `sd` ways)"""
    if stage != "gu":
        ops.gemm(b.attn, lw.o, out=b.seq, residual=b.seq)
    ops.rmsnorm(b.seq, lw.post_norm, EPS, out=b.x)
    ops.gemm(b.x, lw.gate_up, out=b.act)
    if stage in ("mlp", "full"):
        ops.gemm_splitk(b.act, lw.down, b.p_h[:sd], sd)
    if stage == "full":
        ops.residual_rmsnorm(b.p_h[:sd], b.seq, lw.in_norm, EPS, out=b.x)
        ops.gemm(b.x, lw.qkv, out=b.qkv)


def chain(lw, b, stage, counters, B):
    full = engine.layer_chain(lw, lw.in_norm, lw.qkv, attn_out=b.attn, seq=b.seq, act=b.act, p_h=b.p_h, qkv_out=b.qkv, eps=EPS,
                              device=b.seq.device, counters=counters, ss=b.ss)
    if stage == "gu":
        c = [full[1]]
        c[0].wait_cnt = 0
        c[0].sig_cnt = 0
        c[0].publish = 0
        c[0].ss_in = 0
        return c
    if stage == "ogu":
        full[1].sig_cnt = 0
        return full[:2]
    if stage == "mlp":
        full[2].sig_cnt = 0
        return full[:3]
    return full


def timed_graph(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    stage = sys.argv[2] if len(sys.argv) > 2 else "full"
    nl = int(os.environ.get("N_LAYERS", "6"))
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(7)
    layers = [make_layer(dev, gen) for _ in range(nl)]
    # ---------------- correctness: engine vs the classic chain on the same inputs, every intermediate
    lw = layers[0]
    bc, be = Bufs(B, dev, gen), None
    be = Bufs(B, dev, gen)
    for name in ("attn", "seq0"):
        getattr(be, name).copy_(getattr(bc, name))
    be.seq.copy_(be.seq0)
    classic(lw, bc, stage)
    cnt = engine.Counters(dev, 128)
    prog = engine.EngineProgram(chain(lw, be, stage, cnt, B), B, dev, cnt)
    prog.launch()
    torch.cuda.synchronize()
    prog.check_error()

    def cmp(what, a, b):
        a, b = a.float(), b.float()
        same = bool((a == b).all())
        d = (a - b).abs().max().item()
        frac = (a == b).float().mean().item()
        print(f"  {what:10s} bit-identical={same}  equal {frac * 100:.3f}%  max|diff|={d:.4g}  (ref absmax {b.abs().max().item():.4g})")
        return same or (frac > 0.98 and d <= 0.02 * b.abs().max().item())
    ok = True
    if stage != "gu":
        pass
    if stage in ("gu", "ogu"):
        if stage == "ogu":
            ok &= cmp("seq(o)", be.seq, bc.seq)
        ok &= cmp("act", be.act, bc.act) or stage == "ogu"
    if stage == "mlp":
        ok &= cmp("seq(o)", be.seq, bc.seq)
        cmp("act", be.act, bc.act)
        cmp("down part", be.p_h, bc.p_h)
    if stage == "full":
        cmp("seq", be.seq, bc.seq)
        cmp("act", be.act, bc.act)
        cmp("down part", be.p_h, bc.p_h)
        cmp("qkv", be.qkv, bc.qkv)
    print("correctness:", "OK" if ok else "MISMATCH (see above; norm statistics may differ in the last bit)")

    # ---------------- timing: nl layers' weights back to back
    bufs = Bufs(B, dev, gen)

    def run_classic():
        for l in layers:
            classic(l, bufs, stage, sd=int(os.environ.get("SD", "4")))
    cnts = engine.Counters(dev, 128 * nl)
    progs = [engine.EngineProgram(chain(l, bufs, stage, cnts, B), B, dev, cnts) for l in layers]
    for p in progs:           # one counter pool, zeroed by every launch: give each program its own
        p.counters = None
    zero = cnts.buf

    def run_engine():
        zero.zero_()
        for p in progs:
            p.launch()
    if stage == "gu" and os.environ.get("ONE_LAUNCH", "0") != "0":     # all nl gate/up GEMMs as ONE launch: steady-state streaming rate
        ops_all = []
        for l in layers:
            ops_all += chain(l, bufs, "gu", cnts, B)
        big = engine.EngineProgram(ops_all, B, dev, None)

        def run_engine():   # noqa: F811
            big.launch()
        progs = [big]
    tc = timed_graph(run_classic) / nl
    te = timed_graph(run_engine) / nl
    for p in progs:
        p.check_error()
    wbytes = {"gu": 2 * I * H, "ogu": 2 * I * H + H * H, "mlp": 3 * I * H + H * H, "full": 3 * I * H + H * H + QKV * H}[stage] * 2
    if os.environ.get("TRACE", "0") != "0":     # timeline of one launch (lead service wave of a few workgroups), us from the earliest stamp
        tr = torch.zeros(256, 64, dtype=torch.int64, device=dev)
        p0 = progs[min(1, len(progs) - 1)]
        zero.zero_()
        p0.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        zero.zero_()
        e0.record()
        p0.launch(trace=tr)
        e1.record()
        torch.cuda.synchronize()
        t = tr.cpu().double()
        nev = int((t[0] > 0).sum())
        base = t[:, 0].min()
        span_ticks = float(t[:, :nev].max() - base)
        print(f"  trace: {nev} events, kernel {e0.elapsed_time(e1) * 1e3:.1f} us by events, {span_ticks:.0f} ticks first..last stamp")
        names = ["start"]
        for o in chain(layers[0], bufs, stage, engine.Counters(dev, 128), B):
            names += (["r.in", "r.seen", "r.pub"] if o.kind == 1 else ["g.in", "g.seen", "g.x", "g.done"])
        tick_us = 0.01     # s_memtime runs at 100 MHz
        for cu in (0, 1, 100, 223, 224, 255):
            print(f"  cu{cu:3d}: " + "  ".join(f"{names[i] if i < len(names) else i}={(t[cu, i] - base) * tick_us:6.1f}" for i in range(nev)))
        tt = (t[:, :nev] - base) * tick_us
        print("  min   : " + "  ".join(f"{names[i] if i < len(names) else i}={tt[:, i].min():6.1f}" for i in range(nev)))
        print("  max   : " + "  ".join(f"{names[i] if i < len(names) else i}={tt[:, i].max():6.1f}" for i in range(nev)))
    print(f"B={B} stage={stage}: classic {tc:7.2f} us/layer ({wbytes / tc / 1e6:5.2f} TB/s)   engine {te:7.2f} us/layer "
          f"({wbytes / te / 1e6:5.2f} TB/s)   ideal@6.4TB/s {wbytes / 6.4e6:6.2f} us")


if __name__ == "__main__":
    main()
