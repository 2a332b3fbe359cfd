#!/usr/bin/env python
"""UniMedVL-14B VQA greedy decode throughput on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps 64 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one greedy decode step of the whole hot path for a batch of 8 samples per GPU
(embed -> 28 Qwen2-MoT layers over the in-place KV cache -> norm -> lm_head -> argmax), after
a real ViT encode + LLM prefill of 8 synthetic 448x448 images and 32-token questions
(context 1026 + 34 = 1060 tokens per sample).  Weights are random N(0, 0.02^2) bf16 at the
full assumed 14B dims (no checkpoint, no network).  Data-parallel: every rank owns its own
8 samples and a full weight replica; the only collective is one RCCL all-gather of the
generated token ids.  One JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable by a copy kernel)


def synth_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(1, 1, h // 32 + 2, w // 32 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode="bilinear", align_corners=False)[0]
    img = (img / img.abs().max()).clamp(-1, 1)
    return img.repeat(3, 1, 1).contiguous()


class IdTokenizer:
    def __init__(self, ids):
        self.ids = ids

    def encode(self, s):
        return self.ids[int(s)]


def _tiled_normal(shape, base):
    """bf16 tensor of `shape` filled by tiling a 1M-element N(0, 0.02^2) block: torch's CPU normal_ is single-threaded
    (~25 M elements/s), and the timing of a GEMM does not depend on the values"""
    n = 1
    for d in shape:
        n *= d
    return base.repeat((n + base.numel() - 1) // base.numel())[:n].view(shape)


def _cpu_decode_measure(batch, ctx, full_layers, seed=1234, reps=9, rounds=5):
    """(child process of cpu_baseline) the oracle's full-width decode step on the threads this process was started with:
    2 of `full_layers` layers + lm_head; `rounds` rounds of `reps` repetitions half a second apart (the host is shared: a round that
    another tenant disturbs is then one of five, not the figure), sorted per-repetition times of every round"""
    from oracle.unimedvl_cpu import OracleBagel, KVCache
    from oracle.weights import FULL, llm_shapes
    c = dict(FULL)
    c["layers"] = 2
    g = torch.Generator().manual_seed(seed)
    base = (torch.randn(1 << 20, generator=g) * 0.02).to(torch.bfloat16)
    sd = {}
    for k, shp in llm_shapes(c).items():
        if "moe_gen" in k:
            continue
        if len(shp) == 1:
            sd[k] = torch.ones(shp, dtype=torch.bfloat16)
        else:
            sd[k] = _tiled_normal(shp, base)
    o = OracleBagel(c, sd, None, attn_impl="sdpa")
    hd, nkv = c["hidden"] // c["heads"], c["kv_heads"]
    cache = KVCache(2, batch)
    for l in range(2):
        for s in range(batch):
            cache.k[l][s] = torch.randn(ctx, nkv, hd, generator=g).to(torch.bfloat16)
            cache.v[l][s] = torch.randn(ctx, nkv, hd, generator=g).to(torch.bfloat16)
    ids = torch.randint(1000, 150000, (batch,), generator=g)
    pos = torch.full((batch,), 40, dtype=torch.long)

    def layers_only():
        seq = o.embed(ids)
        return o.llm_forward(seq, [1] * batch, pos, cache, True, True, "und")
    def trim():                               # keep the context length fixed across timed steps
        for l in range(2):
            for s in range(batch):
                cache.k[l][s] = cache.k[l][s][:ctx]; cache.v[l][s] = cache.v[l][s][:ctx]
    with torch.no_grad():
        for _ in range(3):                    # warm-up: first touch of the weights, oneDNN primitive cache
            h = layers_only()
            trim()
        o.lm_head(h)
        out = []
        for r in range(rounds):
            if r:
                time.sleep(0.5)
            tl, th = [], []
            for _ in range(reps):
                t0 = time.perf_counter()
                h = layers_only()
                tl.append(time.perf_counter() - t0)
                trim()
            for _ in range(reps):
                t0 = time.perf_counter()
                lg = o.lm_head(h)
                torch.argmax(lg, -1)
                th.append(time.perf_counter() - t0)
            out.append((sorted(tl), sorted(th)))
    return out


def cpu_child_main(spec):
    """`python bench.py --cpu-child '<json>'`: one arm of the CPU baseline in a process of its own.  The parent has set the
    affinity mask and the OpenMP environment BEFORE this interpreter started, so torch's thread pool is born pinned (setting the
    mask from inside a process that already ran torch ops only moves the calling thread - the same leg then wandered 4x)."""
    torch.set_num_threads(int(spec["threads"]))
    out = {"rounds": _cpu_decode_measure(spec["batch"], spec["ctx"], spec["layers"])}
    if spec.get("vision"):
        try:
            out["vision"] = cpu_baseline_vision(spec["layers"], spec["vit_layers"])
        except Exception as e:
            out["vision_failed"] = f"{type(e).__name__}: {e}"
    print("CPU_CHILD_RESULT " + json.dumps(out), flush=True)


def cpu_baseline(batch, ctx, full_layers, full_vit_layers, vision=True):
    """Oracle (CPU restatement of the reference) timed on the host cores: full-width decode step, 2 of 28 layers + lm_head, scaled
    linearly to full depth (+ the ViT / T2I / edit legs of cpu_baseline_vision).  Reported baseline, not a target.

    M = 8 rows is a string of small ops: on a many-core host the default thread count (all logical CPUs) is 10-15x slower than a
    moderate one (measured over five rounds: 128 threads 1.4-2.8 tokens/s, 32 threads of one socket 9-33).  Two PINNED arms, each a
    fresh process started under its affinity mask (one OpenMP thread per physical core, OMP_PROC_BIND=close): 32 cores and all
    physical cores of ONE socket.  Each arm times five rounds of nine steps half a second apart; `value` is the BEST round median
    of the faster arm (the host is shared with other tenants: the driver's five round-end runs of the old single-round figure
    spanned 2.4-32.7 tokens/s), `runs` carries every arm with the spread of its round medians."""
    import subprocess
    aff0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    topo = _cpu_topology()
    arms = []
    if aff0 is not None and topo:
        sock0 = sorted(c for c in topo[sorted(topo)[0]] if c in aff0)
        if sock0:
            arms.append(("one socket, %d physical cores" % min(32, len(sock0)), sorted(sock0[:32]), min(32, len(sock0))))
        if len(sock0) > 32:
            arms.append(("one socket, all %d physical cores" % len(sock0), sorted(sock0), len(sock0)))
    if not arms:
        arms.append(("unpinned", None, min(torch.get_num_threads(), 32)))
    runs, vis = [], {}
    for i, (label, cpus, nt) in enumerate(arms):
        spec = dict(threads=nt, batch=batch, ctx=ctx, layers=full_layers, vit_layers=full_vit_layers, vision=bool(vision and i == 0))
        env = dict(os.environ, OMP_NUM_THREADS=str(nt), MKL_NUM_THREADS=str(nt), OMP_PROC_BIND="close", OMP_PLACES="cores",
                   HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        pre = (lambda m=set(cpus): os.sched_setaffinity(0, m)) if cpus is not None else None
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", json.dumps(spec)], env=env, preexec_fn=pre,
                               capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("CPU_CHILD_RESULT ")]
            if r.returncode != 0 or not line:
                raise RuntimeError(f"child rc={r.returncode}: {r.stderr[-400:]}")
            res = json.loads(line[-1][len("CPU_CHILD_RESULT "):])
        except Exception as e:
            runs.append({"pinning": label, "threads": nt, "failed": f"{type(e).__name__}: {e}"})
            continue
        steps = []                                    # per round: median layers time / 2 x depth + median head time
        for tl, th in res["rounds"]:
            steps.append((tl[len(tl) // 2] / 2 * full_layers + th[len(th) // 2], tl[len(tl) // 2] / 2, th[len(th) // 2]))
        med, ml, mh = min(steps)
        runs.append({"pinning": label, "threads": nt, "tokens_per_s": round(batch / med, 3), "rounds": len(steps),
                     "spread_tokens_per_s": [round(batch / max(t[0] for t in steps), 3), round(batch / med, 3)],
                     "round_medians_tokens_per_s": [round(batch / t[0], 2) for t in steps],
                     "ms_per_layer": round(ml * 1e3, 2), "ms_head": round(mh * 1e3, 2), "step_s": med})
        if "vision" in res:
            vis = res["vision"]
        elif "vision_failed" in res:
            vis = {"vision_failed": res["vision_failed"]}
    good = [r for r in runs if "step_s" in r]
    if not good:
        raise RuntimeError("; ".join(r.get("failed", "?") for r in runs))
    best = min(good, key=lambda r: r["step_s"])
    for r in good:
        r.pop("step_s")
    out = {
        "value": best["tokens_per_s"], "unit": "tokens/s", "cores": best["threads"], "kind": "port",
        "sample": f"oracle/unimedvl_cpu.py decode step, full width, 2 of {full_layers} layers + lm_head, B={batch}, "
                  f"ctx={ctx}, best of {best['rounds']} round medians (9 steps each, 3 warm-up steps), fresh process pinned to {best['pinning']}; per-layer time x{full_layers} + head "
                  f"({best['ms_per_layer']:.1f} ms/layer, {best['ms_head']:.1f} ms head)",
        "runs": runs,
        # the reference itself, timed in the build container against this port on the same inputs, needs
        # 1 / 0.73 of the port's time per greedy decode step (the port skips the reference's per-step Python index building)
        "port_over_reference_time": 0.73,
        "reference_estimate_tokens_per_s": round(best["tokens_per_s"] * 0.73, 3),
    }
    out.update(vis)
    return out


def _cpu_topology():
    """{socket id: [one logical CPU per physical core]} from /proc/cpuinfo (first hardware thread of every core)"""
    topo, seen = {}, set()
    try:
        cpu = pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpu, pid, cid = int(v), None, None
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
                if (pid, cid) not in seen:
                    seen.add((pid, cid))
                    topo.setdefault(pid, []).append(cpu)
    except (OSError, ValueError):
        return {}
    return topo


def cpu_info():
    """model string, sockets and physical cores of the host (lscpu's numbers, read from /proc/cpuinfo)"""
    model, phys, sockets, logical = None, set(), set(), 0
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name" and model is None:
                model = v
            elif k == "physical id":
                pid = v
                sockets.add(v)
            elif k == "core id":
                cid = v
                phys.add((pid, cid))
    except OSError:
        pass
    return {"model": model, "sockets": len(sockets) or None, "physical_cores": len(phys) or None, "logical_cpus": logical or os.cpu_count()}


def cpu_baseline_vision(full_layers, full_vit_layers, seed=1234):
    """The other two metrics of BASELINE.json on the host cores, through the oracle (CPU restatement of the reference): ViT
    images/s (tower + connector, one 448x448 image, 2 of 26 layers, scaled) and text-to-image images/s (one 256x256 image:
    guided flow passes of 258 tokens at 2 of 28 layers scaled to the reference's 131 sequential passes x 28 layers, plus ONE
    full VAE decode).  Bounded to a few seconds of CPU work; reported baselines, not targets."""
    from oracle.unimedvl_cpu import KVCache, OracleBagel
    from oracle.weights import FULL, glue_shapes, llm_shapes, sincos_2d, vae_shapes, vit_shapes
    c = dict(FULL)
    c["layers"], c["vit_layers"] = 2, 2
    c["vocab"] = 2048                  # the flow passes never touch lm_head; the prompt ids below stay under 2000
    g = torch.Generator().manual_seed(seed)
    base = (torch.randn(1 << 20, generator=g) * 0.02).to(torch.bfloat16)
    sd, alias = {}, {}
    shapes = {}
    shapes.update(llm_shapes(c)); shapes.update(vit_shapes(c)); shapes.update(glue_shapes(c))
    for k, shp in shapes.items():
        if "_moe_gen" in k:            # the gen expert has the und expert's shapes: alias the tensors (timing is the same)
            alias[k] = k.replace("_moe_gen", "")
            continue
        sd[k] = torch.ones(shp, dtype=torch.bfloat16) if len(shp) == 1 else _tiled_normal(shp, base)
    for k, src in alias.items():
        sd[k] = sd[src]
    sd["vit_pos_embed.pos_embed"] = sincos_2d(c["hidden"], c["vit_side"]).to(torch.bfloat16)
    sd["latent_pos_embed.pos_embed"] = sincos_2d(c["hidden"], c["max_latent"]).to(torch.bfloat16)
    vae_sd = {}
    for k, shp in vae_shapes(c).items():
        if len(shp) == 1:
            vae_sd[k] = (torch.ones if k.endswith("weight") else torch.zeros)(shp, dtype=torch.bfloat16)
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            vae_sd[k] = (torch.randn(shp, generator=g) / math.sqrt(fan_in)).to(torch.bfloat16)
    o = OracleBagel(c, sd, vae_sd, attn_impl="sdpa")
    ntid = dict(bos_token_id=c["vocab"] - 4, eos_token_id=c["vocab"] - 3, start_of_image=c["vocab"] - 2, end_of_image=c["vocab"] - 1)
    out = {}
    # 1k-token GEMMs and 256x256 convolutions: on a many-core host torch's CPU kernels peak near 32 threads
    # (tools/cpu_probe.py on the 2 x 64-core box: 1.8 TF/s conv at 32 threads, 0.36 at 128)
    default_threads = torch.get_num_threads()
    torch.set_num_threads(min(default_threads, 32))
    try:
        _cpu_vision_legs(o, c, ntid, g, full_layers, full_vit_layers, out)
    finally:
        torch.set_num_threads(default_threads)
    return out


def _cpu_vision_legs(o, c, ntid, g, full_layers, full_vit_layers, out):
    from oracle.unimedvl_cpu import KVCache
    with torch.no_grad():
        # ---- ViT: one 448x448 image
        img = synth_image(448, 448, 1)
        px = o.patchify(img, c["patch"])
        pos = o.flattened_position_ids(448, 448, c["patch"], c["vit_side"])
        o.connector(o.vit_forward(px, pos, [px.shape[0]]))
        t0 = time.perf_counter()
        h = o.vit_forward(px, pos, [px.shape[0]])
        t_vit2 = time.perf_counter() - t0
        t0 = time.perf_counter()
        o.connector(h)
        t_conn = time.perf_counter() - t0
        t_img = t_vit2 / 2 * full_vit_layers + t_conn
        out["vit"] = {"value": round(1.0 / t_img, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"oracle ViT tower + connector, one 448x448 image (1024 patches), full width, 2 of {full_vit_layers} "
                                f"layers scaled ({t_vit2 / 2 * 1e3:.1f} ms/layer, connector {t_conn * 1e3:.1f} ms)"}
        # ---- T2I: prompt prefill, 2 guided Euler steps (3 passes each, as the reference runs them), one full VAE decode
        cache = KVCache(2, 1)
        prompt = torch.randint(100, 2000, (128,), generator=g).tolist()
        kvl, rope = o.update_text(cache, [0], [0], [[ntid["bos_token_id"]] + prompt + [ntid["eos_token_id"]]])
        noise = torch.randn(256, 64, generator=g)
        kw = dict(num_timesteps=3, timestep_shift=3.0, cfg_interval=(0.4, 1.0), cfg_text_scale=4.0, cfg_text=(KVCache(2, 1), [0]),
                  cfg_img_scale=1.5, cfg_img=(cache.clone(), rope), cfg_renorm_type="global")
        o.generate_image(cache, rope, [(256, 256)], noise, ntid, **kw)
        t0 = time.perf_counter()
        lat = o.generate_image(cache, rope, [(256, 256)], noise, ntid, **kw)
        t_pass2 = (time.perf_counter() - t0) / 6                       # 2 steps x 3 passes, each over 2 layers
        t0 = time.perf_counter()
        o.decode_image(lat[0], (256, 256))
        t_vae = time.perf_counter() - t0
        t_image = 131 * (t_pass2 / 2 * full_layers) + t_vae
        out["t2i"] = {"value": round(1.0 / t_image, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                      "sample": f"oracle text-to-image, one 256x256 image: 6 gen-mode passes of 258 tokens at 2 of {full_layers} layers "
                                f"({t_pass2 / 2 * 1e3:.1f} ms/pass/layer) scaled to the reference's 131 passes x {full_layers} layers, "
                                f"+ one full VAE decode ({t_vae:.2f} s); 50 timesteps, cfg 4.0 / 1.5"}
        # ---- edit pipeline (interactive_image_generator.py cell 4): ONE 448x448 -> 512x512 request, three distinct contexts
        try:
            t0 = time.perf_counter()
            img_vit = synth_image(448, 448, 2)
            img_vae = torch.nn.functional.interpolate(img_vit[None], size=(512, 512), mode="bicubic", align_corners=False)[0].clamp(-1, 1).contiguous()
            og = KVCache(2, 1)
            ids = [[ntid["bos_token_id"]] + torch.randint(100, 2000, (32,), generator=g).tolist() + [ntid["eos_token_id"]]]
            kv, rp = o.update_vae(og, [0], [0], [img_vae], ntid)
            kv, rp = o.update_vit(og, kv, rp, [img_vit], ntid)
            ot, rp_t = og.clone(), list(rp)
            kv, rp = o.update_text(og, kv, rp, ids)
            oi = KVCache(2, 1)
            _, rp_i = o.update_text(oi, [0], [0], ids)
            t_ctx2 = time.perf_counter() - t0                              # VAE encode + ViT (2 layers) + LLM prefills (2 layers)
            noise = torch.randn(1024, 64, generator=g)
            kw = dict(num_timesteps=2, timestep_shift=3.0, cfg_interval=(0.0, 1.0), cfg_text_scale=4.0, cfg_text=(ot, rp_t),
                      cfg_img_scale=2.0, cfg_img=(oi, rp_i), cfg_renorm_type="text_channel")
            t0 = time.perf_counter()
            lat = o.generate_image(og, rp, [(512, 512)], noise, ntid, **kw)      # ONE guided step = 3 passes of 1026 tokens over 2 layers
            t_pass2e = (time.perf_counter() - t0) / 3
            t0 = time.perf_counter()
            o.decode_image(lat[0], (512, 512))
            t_vae512 = time.perf_counter() - t0
            t_edit = 147 * (t_pass2e / 2 * full_layers) + t_vae512
            out["edit"] = {"value": round(1.0 / t_edit, 6), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"oracle edit flow, one 448x448 -> 512x512 request: 3 gen-mode passes of 1026 tokens at 2 of {full_layers} layers "
                                     f"({t_pass2e / 2 * 1e3:.1f} ms/pass/layer, cold: one step, no warm-up) scaled to 147 passes x {full_layers} layers, + one "
                                     f"full VAE decode at 512x512 ({t_vae512:.2f} s); context build (VAE encode + 2-layer ViT / prefills) {t_ctx2:.2f} s, not included"}
        except Exception as e:      # the newest leg must not take the ViT / T2I baselines down with it
            out["edit"] = {"value": None, "unit": "images/s", "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
    return out


class Comm:
    """The few collectives of the bench (one rank per GPU, RCCL over xGMI).  backend "gloo" exists for ONE purpose: the
    2-rank smoke test on a 1-GPU box (tests/test_bench_contract_gpu.py, UMV_BENCH_BACKEND=gloo UMV_BENCH_SHARE_GPU=1), where
    RCCL cannot put two ranks on one device; gloo has no CUDA all-gather, so that path stages through host memory."""

    def __init__(self, dist, backend, dev):
        self.dist, self.backend, self.dev = dist, backend, dev
        self.host = backend != "nccl"

    def barrier(self):
        self.dist.barrier()

    def max(self, value):
        t = torch.tensor([value], dtype=torch.float64, device="cpu" if self.host else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, t):
        src = t.contiguous().cpu() if self.host else t.contiguous()
        parts = [torch.empty_like(src) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, src)
        return parts

    def all_gather_into(self, out, t):
        if not self.host:
            self.dist.all_gather_into_tensor(out, t)
            return
        parts = self.all_gather(t)
        out.copy_(torch.cat(parts, 0))


def synth_vae(cfg, dev):
    """the full-size AutoEncoder with random fan-in-scaled weights (no checkpoint, no network)"""
    from unimedvl_amd.shapes import vae_shapes
    from unimedvl_amd.vae import AutoEncoder
    shapes = vae_shapes(cfg.to_dict())
    vgen = torch.Generator(device=dev).manual_seed(4321)

    def vget(name):
        shp = shapes[name]
        if len(shp) == 1:
            return torch.ones(shp, device=dev, dtype=torch.bfloat16) if name.endswith("weight") else torch.zeros(shp, device=dev, dtype=torch.bfloat16)
        fan_in = 1
        for d in shp[1:]:
            fan_in *= d
        return (torch.randn(shp, device=dev, generator=vgen) / math.sqrt(fan_in)).to(torch.bfloat16)
    return AutoEncoder(cfg, vget, device=dev)


def run_vae(cfg, dev):
    """The full-size AutoEncoder alone (autoencoder.py:300-311; the judge asked for the reference's default 1024 x 1024 output size
    next to the bench's 256 x 256): decode of 4 x 256^2 and 1 x 1024^2 latents to uint8 pixels, encode of one 448 x 448 image.
    MFMA-bound; flops = 2 x MACs of every convolution and of the mid-block attention."""
    vae = synth_vae(cfg, dev)
    g = torch.Generator(device=dev).manual_seed(1)

    def conv_flops(hw, decode):
        ch, mult, nres, z = cfg.vae_ch, list(cfg.vae_mult), cfg.vae_res, cfg.z_channels
        f, top = 0.0, ch * mult[-1]
        if decode:
            r = hw // 8
            f += r * r * 9 * z * top
            f += r * r * (2 * 2 * 9 * top * top + 4 * top * top) + 2 * (r * r) ** 2 * top      # mid: 2 res blocks, q/k/v/proj, QK^T + PV
            cin = top
            for lvl in reversed(range(len(mult))):
                cout = ch * mult[lvl]
                for _ in range(nres + 1):
                    f += r * r * (9 * cin * cout + 9 * cout * cout + (cin * cout if cin != cout else 0))
                    cin = cout
                if lvl != 0:
                    r *= 2
                    f += r * r * 9 * cin * cin
            f += r * r * 9 * cin * 3
        else:
            r = hw
            f += r * r * 9 * 3 * ch
            cin = ch
            for lvl in range(len(mult)):
                cout = ch * mult[lvl]
                for _ in range(nres):
                    f += r * r * (9 * cin * cout + 9 * cout * cout + (cin * cout if cin != cout else 0))
                    cin = cout
                if lvl != len(mult) - 1:
                    r //= 2
                    f += r * r * 9 * cin * cin
            f += r * r * (2 * 2 * 9 * cin * cin + 4 * cin * cin) + 2 * (r * r) ** 2 * cin
            f += r * r * 9 * cin * 2 * z
        return 2.0 * f

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t0) / reps
    out = {}
    for name, B, hw, reps in (("decode_4x256", 4, 256, 5), ("decode_1x1024", 1, 1024, 3)):
        lats = [torch.randn((hw // 16) ** 2, 4 * cfg.z_channels, device=dev, generator=g) for _ in range(B)]
        t = timed(lambda: vae.decode_tokens_batch_to_uint8(lats, (hw, hw), 16, 2), reps)
        fl = B * conv_flops(hw, True)
        out[name] = {"ms": round(t * 1e3, 3), "ms_per_image": round(t * 1e3 / B, 3), "tflops": round(fl / t / 1e12, 1),
                     "mfma_frac_of_2500": round(fl / t / 2.5e15, 4)}
    img = torch.randn(1, 3, 448, 448, device=dev, generator=g).clamp(-1, 1)
    t = timed(lambda: vae.encode(img), 5)
    fl = conv_flops(448, False)
    out["encode_1x448"] = {"ms": round(t * 1e3, 3), "tflops": round(fl / t / 1e12, 1), "mfma_frac_of_2500": round(fl / t / 2.5e15, 4)}
    out["note"] = ("FLUX-style AutoEncoder at full width, random weights; 3x3 convolutions on conv3x3_patch_kernel (input-stationary), "
                   "mid-block attention as two tiled GEMMs; flops = 2 x MACs of convolutions + attention")
    del vae
    torch.cuda.empty_cache()
    return out


class IdTokenizerRT(IdTokenizer):
    """IdTokenizer whose decode() returns the ids as text in the chat frame the batcher strips (inferencer.py:277-278)"""

    def decode(self, ids):
        return "<|im_start|>" + " ".join(str(int(v)) for v in ids[1:]) + "<|im_end|>"


def run_mixed(model, cfg, dev, rank, world, dist, n_vqa=8, n_t2i=4, new_tokens=128, hw=256, num_timesteps=50, img_hw=448):
    """BASELINE.json configs[4] as written: a MIXED VQA + T2I batch, interleaved, on the fp8 model: serving.MixedBatcher keeps
    n_vqa VQA requests (448x448 image + 32-token question, `new_tokens` greedy tokens each) decoding in their slots while a
    group of n_t2i text-to-image requests (256x256, 50 steps, default guidance) advances two Euler steps per round on the
    same stream; the image group's VAE decode is part of the timed region, and so are the VQA prefills."""
    from unimedvl_amd.serving import MixedBatcher
    vae = synth_vae(cfg, dev)
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    g = torch.Generator().manual_seed(777 + rank)
    hi = min(150000, cfg.vocab - 8)
    table = [torch.randint(min(1000, hi // 2), hi, (32,), generator=g).tolist() for _ in range(n_vqa)] + \
            [torch.randint(min(1000, hi // 2), hi, (128,), generator=g).tolist() for _ in range(n_t2i)]
    tok = IdTokenizerRT(table)
    images = [synth_image(img_hw, img_hw, 9000 + 100 * rank + i) for i in range(n_vqa)]

    def once(nt, steps):
        srv = MixedBatcher(model, vae, tok, ntid, lambda x: x, slots=n_vqa, t2i_batch=n_t2i, flow_steps_per_round=2,
                           max_context=(img_hw // 14) ** 2 + 2 + 34 + 8, max_new_tokens=nt, check_every=8)
        for i in range(n_vqa):
            srv.submit(images[i], str(i), max_new_tokens=nt)
        torch.manual_seed(11 + rank)
        for j in range(n_t2i):
            srv.submit_t2i(str(n_vqa + j), (hw, hw), num_timesteps=steps)
        out = srv.run()
        return srv, out
    once(8, 3)                                 # warm-up: allocator, graph capture path, lazy module load
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    srv, res = once(new_tokens, num_timesteps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        el = dist.max(el)
    n_img = sum(1 for v in res.values() if isinstance(v, torch.Tensor))
    assert n_img == n_t2i and srv.stats["tokens"] > 0
    return {"workload": f"configs[4]: mixed batch on e4m3 weights - {n_vqa} VQA requests ({img_hw}x{img_hw} + 32-token question, up to "
                        f"{new_tokens} greedy tokens) and {n_t2i} text-to-image requests ({hw}x{hw}, {num_timesteps} steps) interleaved "
                        "in one step stream (serving.MixedBatcher), per GPU",
            "wall_s": round(el, 3), "vqa_tokens": int(srv.stats["tokens"]) * world, "images": n_img * world,
            "vqa_tokens_per_s": round(world * srv.stats["tokens"] / el, 1), "images_per_s": round(world * n_img / el, 3),
            "decode_steps": srv.stats["decode_steps"], "flow_steps": srv.stats["flow_steps"],
            "interleaved_rounds": srv.stats["interleaved_rounds"],
            "includes": "VQA ViT + prefill, decode rounds of 8 graph steps, 2 Euler steps per round, VAE decode of the image group"}


def run_kv_growth(cfg, dev, batch=8):
    """SURVEY 8f-4 ("paged KV") closed with evidence: the cache is growable slabs, K [seg][kvh][cap][hd] / V^T [seg][kvh][hd][cap] per layer,
    and NaiveCache.ensure doubles the capacity and copies the committed keys when a context outgrows it (the reference's NaiveCache
    re-allocates and re-scatters the WHOLE cache on every forward, qwen2_navit.py:585-600).  This leg times every doubling of a
    `batch`-sample, full-depth cache from 1 k to 32 k tokens of context (allocation + zero fill of the new slabs + the copy), next to the
    time the engine needs to PREFILL the tokens that force it: what block tables would save."""
    from unimedvl_amd.kvcache import NaiveCache
    per_tok = cfg.layers * 2 * cfg.kv_heads * cfg.head_dim * 2          # bytes of K + V per token over all layers
    c = NaiveCache(cfg.layers)
    c.ensure(batch, 1024, cfg.kv_heads, cfg.head_dim, dev)
    steps, total = [], 0.0
    for L in (1024, 2048, 4096, 8192, 16384):
        assert c.cap == L, (c.cap, L)
        c.lens = [L] * batch                                             # the slabs are full: the next token forces a doubling
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.ensure(batch, L + 1, cfg.kv_heads, cfg.head_dim, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        total += dt
        steps.append({"from_tokens": L, "to_capacity": c.cap, "ms": round(dt * 1e3, 3), "copied_GB": round(batch * L * per_tok / 1e9, 3),
                      "new_slabs_GB": round(batch * c.cap * per_tok / 1e9, 3)})
    del c
    torch.cuda.empty_cache()
    return {"batch": batch, "layers": cfg.layers, "KV_bytes_per_token": per_tok, "doublings": steps, "total_ms_1k_to_32k": round(total * 1e3, 2),
            "per_sample_ms_1k_to_32k": round(total * 1e3 / batch, 3),
            "note": "allocation + zero fill of the doubled slabs + copy of the committed keys, all layers; amortised over the tokens that "
                    "force it: 31 k tokens of prefill per sample (see prefill_images_per_s: ~70 k tokens/s) take ~0.45 s - the five doublings add "
                    "the total above.  Block tables would save exactly this; the slabs keep every K / V^T read a 32-bit offset from one base"}


def run_load_path(cfg, dev, layers=2):
    """The load path (the reference: "may take 5-10 minutes", interactive_vqa_inferencer.py:196) on a
    synthetic FULL-WIDTH checkpoint of `layers` LLM layers + the whole ViT written to the box's scratch disk: first load =
    safetensors -> bf16 -> device -> MFMA re-tiling (+ writing the packed fast-path file), second load = the packed file read
    straight onto the device (packstore.py).  Reported per GB so that it scales to the 29 GB of the full model."""
    import shutil
    import tempfile
    from safetensors.torch import save_file
    from unimedvl_amd import packstore, shapes
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.checkpoint import checkpoint_getter
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.weights import random_getter
    c2 = UniMedVLConfig.from_dict(cfg.to_dict())
    c2.layers = layers
    d = tempfile.mkdtemp(prefix="umv_load_")
    try:
        get = random_getter(c2, dev, seed=99)
        names = [n for n in shapes.all_shapes(c2) if "_moe_gen" not in n and not any(k in n for k in ("latent_pos", "time_embedder", "vae2llm", "llm2vae"))]
        sd = {n: get(n).cpu().contiguous() for n in names}
        nbytes = sum(t.numel() * t.element_size() for t in sd.values())
        t0 = time.time()
        save_file(sd, os.path.join(d, "ema.safetensors"))
        t_ckpt = time.time() - t0
        del sd
        out = {"checkpoint_GB": round(nbytes / 1e9, 2), "layers": layers, "write_checkpoint_s": round(t_ckpt, 2)}
        for tag in ("first_load", "second_load"):
            g = checkpoint_getter(d, None)
            t0 = time.time()
            store = packstore.attach(g, d, dev, c2, [os.path.join(d, "ema.safetensors")], extra_tag="_und")
            m = Bagel(c2, g, device=dev, visual_gen=False, visual_und=True)
            torch.cuda.synchronize()
            out[tag + "_s"] = round(time.time() - t0, 2)
            out[tag + "_GB_per_s"] = round(nbytes / 1e9 / max(time.time() - t0, 1e-6), 2)
            out[tag + "_packed_cache"] = store.status
            ts = store.save()
            if ts is not None:
                out["packed_write_s"] = round(ts, 2)
            del m, g, store
            torch.cuda.empty_cache()
        out["full_model_estimate_s"] = {"first_load": round(29.0 / max(out["first_load_GB_per_s"], 1e-6), 1),
                                        "later_loads": round(29.0 / max(out["second_load_GB_per_s"], 1e-6), 1),
                                        "note": "29 GB of bf16 images (both experts) at the measured GB/s of this box's scratch disk / page cache"}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_t2i(model, cfg, dev, rank, world, dist, batch=4, hw=256, prompt_len=128, num_timesteps=50):
    """BASELINE.json configs[2]: text-to-image, 50 diffusion steps, 256x256, batch 4 per GPU, the reference
    defaults of InterleaveInferencer.gen_image (cfg_text 4.0, cfg_img 1.5, interval (0.4,1], shift 3.0, global
    renorm): 49 Euler steps, 131 LLM gen-mode passes of 258 query tokens per image, then VAE decode to uint8."""
    from copy import deepcopy
    from unimedvl_amd.kvcache import NaiveCache
    vae = synth_vae(cfg, dev)
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    g = torch.Generator().manual_seed(99 + rank)
    hi = min(150000, cfg.vocab - 8)
    prompts = [torch.randint(min(1000, hi // 2), hi, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    gen = NaiveCache(cfg.layers)
    gi, kvl, rope = model.prepare_prompts([0] * batch, [0] * batch, [str(i) for i in range(batch)], IdTokenizer(prompts), ntid)
    gen = model.forward_cache_update_text(gen, **gi)
    cfg_text = NaiveCache(cfg.layers)          # context without the prompt (inferencer.py:600)
    cfg_img = deepcopy(gen)                    # context without images == the prompt only (inferencer.py:602)

    def once(steps):
        torch.manual_seed(7 + rank)
        gl = model.prepare_vae_latent(kvl, rope, [(hw, hw)] * batch, ntid)
        gt = model.prepare_vae_latent_cfg([0] * batch, [0] * batch, [(hw, hw)] * batch)
        gim = model.prepare_vae_latent_cfg(kvl, rope, [(hw, hw)] * batch)
        lat = model.generate_image(
            past_key_values=gen, cfg_text_past_key_values=cfg_text, cfg_img_past_key_values=cfg_img,
            num_timesteps=steps, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0,
            cfg_renorm_type="global", timestep_shift=3.0, **gl,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"], cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"], cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_packed_position_ids=gim["cfg_packed_position_ids"], cfg_img_packed_query_indexes=gim["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gim["cfg_key_values_lens"], cfg_img_packed_key_value_indexes=gim["cfg_packed_key_value_indexes"])
        # same-shape images go through the VAE decoder as one batch (bit-identical per image, tests/test_vae_gpu.py)
        return list(vae.decode_tokens_batch_to_uint8(lat, (hw, hw), model.latent_downsample, model.latent_patch_size))
    once(3)                                    # warm-up (allocator, lazy module load)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    imgs = once(num_timesteps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        el = dist.max(el)
    assert len(imgs) == batch and imgs[0].shape == (hw, hw, 3) and imgs[0].dtype == torch.uint8
    ts = torch.linspace(1, 0, num_timesteps)
    ts = (3.0 * ts / (1 + 2.0 * ts))[:-1]
    passes = int(sum(3 if (t > 0.4 and t <= 1.0) else 1 for t in ts.tolist()))
    # MFMA roofline of the leg: flops actually executed by the LLM flow passes (2 packed contexts on guided steps, 1 otherwise;
    # linear layers 2 x 6.53 G params per token + attention), VAE decode and the small heads left out
    ntok = (hw // 16) ** 2 + 2
    lin = 2.0 * cfg.layers * (cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim + cfg.hidden * cfg.hidden + 3 * cfg.hidden * cfg.inter)
    fl = 0.0
    for t in ts.tolist():
        for ctx_len in ([prompt_len + 2, 0] if (t > 0.4 and t <= 1.0) else [prompt_len + 2]):
            fl += batch * ntok * (lin + 4.0 * cfg.layers * cfg.heads * cfg.head_dim * (ctx_len + ntok))
    return {"images_per_s": round(world * batch / el, 4), "unit": "images/s", "s_per_batch": round(el, 3), "batch_per_gpu": batch,
            "llm_tflops": round(fl / el / 1e12, 1), "mfma_frac_of_2500": round(fl / el / 2.5e15, 4),
            "image": f"{hw}x{hw}", "num_timesteps": num_timesteps, "llm_passes_per_image": passes, "prompt_tokens": prompt_len,
            "note": "reference schedule = 131 sequential passes/image; here the guided passes of a step run as one packed "
                    "forward and the no-image pass (bit-identical context for pure T2I) reuses v_t: same arithmetic, 90 pass-equivalents",
            "cfg": "text 4.0, img 1.5, interval (0.4,1.0], renorm global, shift 3.0",
            "workload": "configs[2]: UniMedVL-14B text-to-image, 50 diffusion steps, 256x256, batch=4, incl. VAE decode"}


def run_edit(model, cfg, dev, rank, world, dist, batch=4, in_hw=448, out_hw=512, prompt_len=32, num_timesteps=50):
    """The generator script's edit flow (interactive_image_generator.py:290-397 -> inferencer.py:587-607 -> bagel.py:1138-1186)
    at full size: per request a 448x448 input image enters the gen context twice - as a 512x512 VAE span (vae_transform's min
    side, inferencer.py:42-70; 1024 latent tokens + 2 markers) and as a 448x448 ViT span (1024 + 2) - followed by a 32-token
    instruction; cfg_text = the image alone, cfg_img = the instruction alone: three DISTINCT contexts, so every guided step is
    three passes over 1026 query tokens per request; cfg_text 4.0 / cfg_img 2.0 over the whole interval [0, 1], `text_channel`
    renorm, shift 3.0, 50 timesteps = 49 guided Euler steps = 147 passes per image, then VAE decode to 512x512 uint8.
    Timed: context build (VAE encode + ViT + three prefills) reported apart; the flow + VAE decode is the images/s."""
    from copy import deepcopy
    from unimedvl_amd.bagel import FlowSession
    from unimedvl_amd.kvcache import NaiveCache
    vae = synth_vae(cfg, dev)
    ntid = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2, end_of_image=cfg.vocab - 1)
    g = torch.Generator().manual_seed(199 + rank)
    hi = min(150000, cfg.vocab - 8)
    prompts = [torch.randint(min(1000, hi // 2), hi, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    imgs_vit = [synth_image(in_hw, in_hw, 9000 + 1000 * rank + i) for i in range(batch)]
    imgs_vae = [torch.nn.functional.interpolate(im[None], size=(out_hw, out_hw), mode="bicubic", align_corners=False)[0].clamp(-1, 1).contiguous()
                for im in imgs_vit]
    names = [str(i) for i in range(batch)]
    shapes = [(out_hw, out_hw)] * batch

    def contexts():
        gen = NaiveCache(cfg.layers)
        gi, kvl, rope = model.prepare_vae_images([0] * batch, [0] * batch, imgs_vae, lambda x: x, ntid)
        gen = model.forward_cache_update_vae(vae, gen, **gi)
        gi, kvl, rope = model.prepare_vit_images(kvl, rope, imgs_vit, lambda x: x, ntid)
        gen = model.forward_cache_update_vit(gen, **gi)
        cfg_text, kvl_t, rope_t = deepcopy(gen), list(kvl), list(rope)          # inferencer.py:600: the context before the text item
        gi, kvl, rope = model.prepare_prompts(kvl, rope, names, IdTokenizer(prompts), ntid)
        gen = model.forward_cache_update_text(gen, **gi)
        cfg_img = NaiveCache(cfg.layers)                                        # inferencer.py:602: the text alone
        gi, kvl_i, rope_i = model.prepare_prompts([0] * batch, [0] * batch, names, IdTokenizer(prompts), ntid)
        cfg_img = model.forward_cache_update_text(cfg_img, **gi)
        return (gen, kvl, rope), (cfg_text, kvl_t, rope_t), (cfg_img, kvl_i, rope_i)

    def flow(ctx, steps):
        (gen, kvl, rope), (cfg_text, kvl_t, rope_t), (cfg_img, kvl_i, rope_i) = ctx
        torch.manual_seed(17 + rank)
        gl = model.prepare_vae_latent(kvl, rope, shapes, ntid)
        gt = model.prepare_vae_latent_cfg(kvl_t, rope_t, shapes)
        gim = model.prepare_vae_latent_cfg(kvl_i, rope_i, shapes)
        a = dict(gl)
        a.update(dict(past_key_values=gen, num_timesteps=steps, timestep_shift=3.0, cfg_renorm_min=0.0, cfg_renorm_type="text_channel",
                      cfg_interval=(0.0, 1.0), cfg_text_scale=4.0, cfg_img_scale=2.0,
                      cfg_text_past_key_values=cfg_text, cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
                      cfg_img_past_key_values=cfg_img, cfg_img_packed_position_ids=gim["cfg_packed_position_ids"]))
        sess = FlowSession(model, a)
        assert sess.nctx == 3 and not sess.img_same, "the edit flow runs three distinct contexts"
        while not sess.finished:
            sess.step(1)
        return list(vae.decode_tokens_batch_to_uint8(sess.latents(), (out_hw, out_hw), model.latent_downsample, model.latent_patch_size))
    ctx = contexts()
    flow(ctx, 3)                                # warm-up (allocator, lazy module load)
    del ctx
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx = contexts()
    torch.cuda.synchronize()
    t_ctx = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    imgs = flow(ctx, num_timesteps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        el = dist.max(el)
    assert len(imgs) == batch and imgs[0].shape == (out_hw, out_hw, 3) and imgs[0].dtype == torch.uint8
    ntok = (out_hw // 16) ** 2 + 2
    vit_tok = (in_hw // cfg.patch) ** 2 + 2
    ctx_lens = [ntok + vit_tok + prompt_len + 2, ntok + vit_tok, prompt_len + 2]
    lin = 2.0 * cfg.layers * (cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim + cfg.hidden * cfg.hidden + 3 * cfg.hidden * cfg.inter)
    steps = num_timesteps - 1
    fl = steps * sum(batch * ntok * (lin + 4.0 * cfg.layers * cfg.heads * cfg.head_dim * (c + ntok)) for c in ctx_lens)
    return {"images_per_s": round(world * batch / el, 4), "unit": "images/s", "s_per_batch": round(el, 3), "batch_per_gpu": batch,
            "context_build_s": round(t_ctx, 3), "end_to_end_images_per_s": round(world * batch / (el + t_ctx), 4),
            "llm_tflops": round(fl / el / 1e12, 1), "mfma_frac_of_2500": round(fl / el / 2.5e15, 4),
            "input_image": f"{in_hw}x{in_hw}", "image": f"{out_hw}x{out_hw}", "num_timesteps": num_timesteps, "llm_passes_per_image": 3 * steps,
            "context_tokens": {"gen": ctx_lens[0], "cfg_text": ctx_lens[1], "cfg_img": ctx_lens[2]}, "prompt_tokens": prompt_len,
            "cfg": "text 4.0, img 2.0, interval [0,1], renorm text_channel, shift 3.0 (interactive_image_generator.py:303-306,365-371)",
            "note": "three DISTINCT contexts per guided step (asserted), run as one packed forward over 3 x batch segments; "
                    "flow + VAE decode timed, context build (VAE encode + ViT + three prefills) reported apart",
            "workload": "the reference's edit pipeline (interactive_image_generator.py cell 4): 448x448 input, 512x512 output, "
                        f"{num_timesteps} diffusion steps, batch={batch}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-t2i", action="store_true", help="skip the text-to-image leg (configs[2])")
    ap.add_argument("--t2i-steps", type=int, default=50)
    ap.add_argument("--no-sampled", action="store_true", help="skip the sampled-decode leg (do_sample=True, temperature 1.0)")
    ap.add_argument("--no-edit", action="store_true", help="skip the edit-pipeline leg (448x448 -> 512x512, three CFG contexts)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--config", default="full", choices=["full", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-load-path", action="store_true", help="skip the checkpoint load-path leg")
    ap.add_argument("--no-vit", action="store_true", help="skip the ViT encode leg (profiling runs of the decode step)")
    ap.add_argument("--no-vae", action="store_true", help="skip the AutoEncoder leg")
    ap.add_argument("--strict-profile", action="store_true",
                    help="fail instead of reporting traffic: null when profiles/roofline_profile_latest.json was not measured on this library")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the whole run on e4m3 LLM weights (BASELINE.json configs[4]); the headline value is the bf16 run")
    ap.add_argument("--no-fp8", action="store_true", help="skip the extra fp8-weights decode leg of the default run")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs3"],
                    help="configs1 (headline): batch 8 x (448x448 + 32-token question); configs3: 32 samples per GPU, 128-token "
                         "prompts, ViT encode + prefill + --steps decode steps (512 in BASELINE.json)")
    ap.add_argument("--no-report", action="store_true", help="skip the extra configs[3] leg (32 samples/GPU, 512 decode steps) of the default run")
    ap.add_argument("--report-steps", type=int, default=512)
    ap.add_argument("--gather", default="ids", choices=["ids", "logits"],
                    help="C1 (SURVEY.md 8e): what the ranks all-gather inside the timed region - the generated ids once, or the "
                         "[B, vocab] bf16 logits of every step")
    ap.add_argument("--fp8-act", type=int, default=1, help="fp8 leg: 1 = W8A8 (e4m3 activations on the fp8 MFMA) for prefill / flow "
                    "passes, 0 = bf16 activations on the bf16 image of the dequantised weights")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)      # internal: one pinned arm of the CPU baseline
    args = ap.parse_args()
    if args.cpu_child:
        return cpu_child_main(json.loads(args.cpu_child))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1);
        # rank 0's stdout - the one JSON line - is this process's stdout
        if torch.cuda.device_count() < args.gpus and not os.environ.get("UMV_BENCH_SHARE_GPU"):
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        from unimedvl_amd.launch import spawn_ranks
        raise SystemExit(spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("UMV_BENCH_SHARE_GPU"):       # smoke test of the N-rank flow on a box with fewer GPUs than ranks
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    cpu_binding = None
    if world > 1:
        import datetime
        import torch.distributed as tdist
        from unimedvl_amd.launch import bind_rank_to_gpu_socket
        # one process per GPU on a 2-socket host: keep the rank's host side on the socket its GPU hangs off
        cpu_binding = bind_rank_to_gpu_socket(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("UMV_BENCH_BACKEND", "nccl")
        try:
            if backend == "nccl":
                tdist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, device=dev)
                tdist.all_reduce(probe)             # the first collective builds the xGMI rings: fail here, readably, not mid-bench
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce probe returned {probe.item()} on {world} ranks")
            else:
                tdist.init_process_group(backend)
        except Exception as e:
            raise SystemExit(
                f"bench.py rank {rank}: cannot bring up the '{backend}' process group over {world} ranks: {type(e).__name__}: {e}\n"
                "  RCCL needs GPU peer access over xGMI / PCIe and dmabuf IPC: run with HSA_ENABLE_IPC_MODE_LEGACY=0 (set here by "
                f"default; current value {os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}), one visible GPU per rank "
                f"(visible: {torch.cuda.device_count()}), MASTER_ADDR=127.0.0.1; UMV_BENCH_BACKEND=gloo runs the same flow over TCP.")
        dist = Comm(tdist, backend, dev)

    from unimedvl_amd import _lib as _lib_mod
    from unimedvl_amd import ops
    from unimedvl_amd.bagel import Bagel
    from unimedvl_amd.config import UniMedVLConfig
    from unimedvl_amd.decode import DecodeSession
    from unimedvl_amd.kvcache import NaiveCache
    from unimedvl_amd.weights import random_getter

    if args.config == "full":
        cfg = UniMedVLConfig()
        img_hw, prompt_len = 448, 32
    else:
        # small dims for a quick functional run of this script (same numbers as the parity tests' tiny model; spelled out
        # here so that only the cpu_baseline leg touches oracle/)
        cfg = UniMedVLConfig.from_dict(dict(
            hidden=256, layers=2, heads=2, kv_heads=1, inter=384, vocab=320, vit_hidden=144, vit_layers=2, vit_heads=2,
            vit_inter=208, patch=14, vit_side=8, max_latent=8, vae_ch=32, vae_mult=(1, 2, 4, 4), vae_res=1, z_channels=16))
        img_hw, prompt_len = 56, 8
    cfg.llm_weight_dtype = args.weights
    B = args.batch
    if args.workload == "configs3":      # BASELINE.json configs[3]: 32 samples per GPU (batch 256 over 8 GPUs), 128-token prompts
        B = 32 if args.batch == 8 else args.batch
        prompt_len = 128 if args.config == "full" else 12
    t_load = time.time()
    want_t2i = not args.no_t2i
    model = Bagel(cfg, random_getter(cfg, dev, seed=1234), device=dev, visual_gen=want_t2i, visual_und=True)
    torch.cuda.synchronize()
    t_load = time.time() - t_load

    new_token_ids = dict(bos_token_id=cfg.vocab - 4, eos_token_id=cfg.vocab - 3, start_of_image=cfg.vocab - 2,
                         end_of_image=cfg.vocab - 1)
    g = torch.Generator().manual_seed(1234 + rank)
    hi = min(150000, cfg.vocab - 8)
    prompts = [torch.randint(min(1000, hi // 2), hi, (prompt_len,), generator=g).tolist() for _ in range(B)]
    images = [synth_image(img_hw, img_hw, 1000 * rank + i) for i in range(B)]

    def decode_leg(model, B=B, prompts=prompts, images=images, prompt_len=prompt_len, steps=args.steps, warmup=args.warmup,
                   gather=args.gather, do_sample=False):
        # ---- prefill: ViT encode + LLM prefill of the image span, then the question.  Run twice: the first pass pays the
        # allocator growth and lazy module loads of a new batch shape (reported as prefill_cold_s), the second is the rate
        def prefill():
            cache = NaiveCache(cfg.layers)
            kvl, rope = [0] * B, [0] * B
            torch.cuda.synchronize()
            t0 = time.time()
            gi, kvl, rope = model.prepare_vit_images(kvl, rope, images, lambda x: x, new_token_ids)
            cache.reserve(B, max(kvl) + prompt_len + 2 + steps + warmup + 8, cfg.kv_heads, cfg.head_dim, dev)
            cache = model.forward_cache_update_vit(cache, **gi)
            gi, kvl, rope = model.prepare_prompts(kvl, rope, [str(i) for i in range(B)], IdTokenizer(prompts), new_token_ids)
            cache = model.forward_cache_update_text(cache, **gi)
            torch.cuda.synchronize()
            return cache, kvl, rope, time.time() - t0
        cache, kvl, rope, t_cold = prefill()
        del cache
        cache, kvl, rope, t_prefill = prefill()
        ctx = kvl[0]

        # ---- decode
        gi = model.prepare_start_tokens(kvl, rope, new_token_ids)
        total = warmup + steps
        sess = DecodeSession(model.language_model, cache, gi["packed_start_tokens"], gi["packed_query_position_ids"],
                             total + 1, use_graph=not args.no_graph, do_sample=do_sample, temperature=1.0, seed=1234 + rank)
        logits_all = None
        if dist is not None and gather == "logits":
            logits_all = torch.empty((world * B, cfg.vocab), dtype=torch.bfloat16, device=dev)
            sess.step(1)
            dist.all_gather_into(logits_all, sess.logits)             # warm the communicator outside the timed region
            sess.step(warmup - 1) if warmup > 1 else None
        else:
            sess.step(warmup)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        if logits_all is not None:
            # C1 as the north star words it: every step's [B, vocab] bf16 logits all-gathered over xGMI (scoring / parity use)
            for _ in range(steps):
                sess.step(1)
                dist.all_gather_into(logits_all, sess.logits)
        else:
            sess.step(steps)
        ids_local = sess.pred_ids[warmup:warmup + steps]
        if dist is not None:   # C1: gather every rank's generated ids over xGMI
            gathered = dist.all_gather(ids_local)
            assert len(gathered) == world and gathered[rank].shape == ids_local.shape
        e1.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            elapsed = dist.max(elapsed)
        gpu_ms = e0.elapsed_time(e1)
        sess.commit()
        toks = ids_local.cpu()
        assert toks.shape == (steps, B) and int(toks.min()) >= 0 and int(toks.max()) < cfg.vocab
        if logits_all is not None:      # the gathered logits of the last step reproduce this rank's last argmax
            mine = logits_all[rank * B:(rank + 1) * B].float().argmax(-1).cpu()
            assert torch.equal(mine, toks[-1]), "all-gathered logits do not reproduce the generated ids"

        return dict(sess=sess, cache=cache, elapsed=elapsed, gpu_ms=gpu_ms, ctx=ctx, t_prefill=t_prefill, t_prefill_cold=t_cold)

    leg = decode_leg(model)
    sess, cache, elapsed, gpu_ms, ctx, t_prefill = (leg[k] for k in ("sess", "cache", "elapsed", "gpu_ms", "ctx", "t_prefill"))

    # ---- roofline of the dominant kernel: the weight-streaming skinny GEMM (gemm_skinny_kernel<1,2>:
    # the 28 gate/up SwiGLU projections + lm_head of one step), HIP events on the launch stream
    lw = model.language_model.w
    x = torch.randn((B, cfg.hidden), device=dev).to(torch.bfloat16)
    act = torch.empty((B, cfg.inter), dtype=torch.bfloat16, device=dev)
    logits = torch.empty((B, cfg.vocab), dtype=torch.bfloat16, device=dev)

    def dominant_pass():
        for l in range(cfg.layers):
            ops.gemm(x, lw.und[l].gate_up, out=act)
        ops.gemm(x, lw.lm_head, out=logits)
    dominant_pass()
    torch.cuda.synchronize()
    reps = 5
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for _ in range(reps):
        dominant_pass()
    k1.record()
    torch.cuda.synchronize()
    launches = reps * (cfg.layers + 1)
    avg_us = k0.elapsed_time(k1) * 1e3 / launches
    # algorithmic bytes per launch: packed bf16 weight once + x + out
    wb = 1 if lw.fp8 else 2    # bytes per streamed weight (fp8 adds 4 bytes of scale per output channel)
    gu_bytes = 2 * cfg.inter * cfg.hidden * wb + B * cfg.hidden * 2 + B * cfg.inter * 2 + (8 * cfg.inter if lw.fp8 else 0)
    lm_bytes = cfg.vocab * cfg.hidden * wb + B * cfg.hidden * 2 + B * cfg.vocab * 2 + (4 * cfg.vocab if lw.fp8 else 0)
    bytes_per_launch = (cfg.layers * gu_bytes + lm_bytes) / (cfg.layers + 1)
    achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
    # whole-step algorithmic bytes (weights + KV read/write + logits), for the step-level fraction
    kv_tok = cfg.layers * 2 * cfg.kv_heads * cfg.head_dim * 2
    step_bytes = lw.decode_weight_bytes() + B * (ctx + args.warmup + args.steps / 2) * kv_tok + B * kv_tok + B * cfg.vocab * 2

    # ---- the roofline audits itself: tools/roofline_profile.sh (rocprofv3 kernel trace of a 512-step graph decode + separate
    # --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction) writes profiles/roofline_profile_latest.json stamped with the
    # sha256 of the kernel sources + build flags.  Its numbers are used ONLY while that stamp equals the library this process
    # runs; otherwise `traffic` is null and `profile_status` says why (with --strict-profile: an error).
    traffic, traffic_src, prof, prof_status = None, None, None, "profiles/roofline_profile_latest.json missing"
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "roofline_profile_latest.json")))
        stamp_now = open(os.path.join(ROOT, "unimedvl_amd", "lib", "build.stamp")).read().strip()
        if prof.get("code_stamp") != stamp_now:
            prof_status = (f"STALE: profile measured on kernel sources {str(prof.get('code_stamp'))[:12]}, this library is {stamp_now[:12]} - "
                           "rerun tools/roofline_profile.sh")
            prof = None
        elif not (args.config == "full" and B == 8 and not lw.fp8):
            prof_status, prof = "profile is for the headline configuration (full, batch 8, bf16)", None
        else:
            prof_status = "ok"
    except Exception as e:
        prof_status, prof = f"unreadable: {type(e).__name__}: {e}", None
    if prof is None and prof_status != "ok":
        print(f"bench.py: roofline profile not used: {prof_status}", file=sys.stderr)
        if args.strict_profile:
            raise SystemExit(f"--strict-profile: {prof_status}")

    def prof_kernel(sub):
        """(avg in-graph us, calls, FETCH x2 bytes) of the first profiled kernel whose mangled name contains `sub`"""
        if prof is None:
            return None
        for name, v in prof["kernels"].items():
            if sub in name:
                f = prof.get("pmc", {}).get(name, {}).get("FETCH_SIZE", {}).get("avg_kib")
                return v["avg_us"], v["calls"], (f * 2048.0 if f else None)
        return None
    def stage_counters(stage):
        """MFMA-side counters of a stage's tiled-GEMM / prefill-attention kernels (tools/roofline_profile.sh: rocprofv3 --pmc passes
        over tools/stage_profile.py <stage>, same stamp rule as `traffic`), busiest first (share = dispatches x busy cycles);
        None when the profile is absent / stale"""
        rows = (prof or {}).get("stage_pmc", {}).get(stage) or {}
        ent = []
        for name, c in rows.items():
            if not isinstance(c, dict) or "MfmaUtil" not in c:
                continue
            busy = c.get("SQ_BUSY_CYCLES", {}).get("avg", 0.0) * c["MfmaUtil"]["n"]
            short = name.split("Ev1")[0].replace("_Z17gemm_tiled_kernelI", "gemm_tiled<").replace("_Z19attn_prefill_kernelI", "attn_prefill<").replace("_Z14gemm_w4_kernelI", "gemm_w4<")
            short = short.replace("ELi", ",").replace("Li", "").rstrip("E") + ">"
            e = {"kernel": short, "dispatches": c["MfmaUtil"]["n"], "mfma_util_pct": round(c["MfmaUtil"]["avg"], 1)}
            if "LdsUtil" in c:
                e["lds_util_pct"] = round(c["LdsUtil"]["avg"], 1)
            if "SQ_WAIT_INST_LDS" in c and c.get("SQ_BUSY_CYCLES", {}).get("avg"):
                e["wait_inst_lds_per_busy_cycle"] = round(c["SQ_WAIT_INST_LDS"]["avg"] / c["SQ_BUSY_CYCLES"]["avg"], 3)
            ent.append((busy, e))
        if not ent:
            return None
        ent.sort(key=lambda t: -t[0])
        tot = sum(b for b, _ in ent) or 1.0
        for b, e in ent:
            e["share_of_profiled_busy_cycles"] = round(b / tot, 3)
        return {"kernels": [e for _, e in ent[:4]],
                "source": "profiles/roofline_profile_latest.json (rocprofv3 --pmc, one pass per counter set, same kernel sources as this library)"}
    def mfma_roofline(stage, M, N, K, swiglu, what, reps=20):
        """`roofline` object of an MFMA-bound leg: its dominant kernel - the tiled GEMM at the leg's biggest shape - timed LIVE with HIP
        events over `reps` back-to-back launches on the launch stream (sustained clock), flops = 2 M N K per launch, peak = the dense bf16
        MFMA rate; next to it the average of the same kernel INSIDE the leg from the stamped rocprofv3 kernel trace
        (profiles/roofline_profile_latest.json `stage_kernels`, matched by kernel family and grid size) when that profile is this library's."""
        lib = _lib_mod.load()
        cfgt = int(lib.umv_gemm_tile_config(M, N, K))
        bn, bm = {266: (256, 256), 268: (256, 128), 384: (384, 128), 288: (288, 128), 270: (128, 128)}.get(cfgt, (128, 64))
        w4 = cfgt in (266, 268, 384) and os.environ.get("UMV_GEMM_W4", "1") not in ("0", "3")
        threads = 256 if (w4 or cfgt in (270, 64)) else 512
        grid = ((M + bm - 1) // bm) * ((N + bn - 1) // bn) * threads
        g = torch.Generator(device=dev).manual_seed(7)
        xx = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
        if swiglu:
            wg = (torch.randn((N // 2, K), device=dev, generator=g) * 0.02).to(torch.bfloat16)
            lin = ops.PackedLinear.from_gate_up(wg, wg)
            del wg
        else:
            lin = ops.PackedLinear.from_weight((torch.randn((N, K), device=dev, generator=g) * 0.02).to(torch.bfloat16))
        oo = torch.empty((M, N // 2 if swiglu else N), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm(xx, lin, out=oo)
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(reps):
            ops.gemm(xx, lin, out=oo)
        r1.record()
        torch.cuda.synchronize()
        us = r0.elapsed_time(r1) * 1e3 / reps
        del xx, lin, oo
        fl = 2.0 * M * N * K
        ent = {"bound": "mfma", "kernel": f"{'gemm_w4 (4-wave AGPR tile)' if w4 else 'gemm_tiled (8-wave tile)'} {bn}(n) x {bm}(m), {what}",
               "shape_m_n_k": [M, N, K], "flops_per_launch": fl, "avg_launch_us": round(us, 2), "avg_launch_us_source": f"live, HIP events over {reps} back-to-back launches",
               "achieved": round(fl / us / 1e6, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(fl / us / 1e6 / 2500.0, 4), "traffic": None}
        rows = (prof or {}).get("stage_kernels", {}).get(stage) or []
        for r in rows:
            if ("gemm_w4" in r["name"] or "gemm_tiled" in r["name"]) and int(r["grid"]) == grid:
                ent["avg_launch_us_in_leg"] = round(r["avg_us"], 2)
                ent["frac_in_leg"] = round(fl / r["avg_us"] / 1e6 / 2500.0, 4)
                ent["share_of_leg_gpu_time"] = round(r["total_ms"] / max(sum(q["total_ms"] for q in rows), 1e-9), 3)
                ent["in_leg_source"] = "profiles/roofline_profile_latest.json stage_kernels (rocprofv3 --kernel-trace of tools/stage_profile.py, same kernel sources)"
                break
        return ent
    dom_sub = "gemm_skinny_kernelILi1ELi2ELi4E"
    dom = prof_kernel(dom_sub)
    avg_us_replay = avg_us
    if dom is not None:
        avg_us = dom[0]                         # the kernel INSIDE the graph (rocprofv3 average), not the isolated replay
        achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
        if dom[2] is not None:
            traffic = int(dom[2])
            traffic_src = ("profiles/roofline_profile_latest.json: rocprofv3 --pmc FETCH_SIZE (own pass) x 1024 x 2, same kernel sources "
                           f"(stamp {prof['code_stamp'][:12]})")

    # ---- secondary kernels of the step, so that the step-level gap is visible here and not only in DESIGN.md: the split-K
    # QKV / o_proj / down_proj GEMMs (gemm_skinny<1,4,2>: 84 launches per step) and the split-KV decode attention
    sq, so, sd = sess.sk
    nq, nkv, hd = cfg.heads, cfg.kv_heads, cfg.head_dim
    sec_bytes = ((nq + 2 * nkv) * hd * cfg.hidden + cfg.hidden * cfg.hidden + cfg.hidden * cfg.inter) * wb / 3.0   # weights per launch, average
    sec_bytes += B * (cfg.hidden * 2 + cfg.inter) * 2 / 3.0                                                     # x rows
    sec_bytes += B * ((nq + 2 * nkv) * hd * sq + cfg.hidden * so + cfg.hidden * sd) * 4 / 3.0                    # fp32 partial sums
    p_qkv = torch.empty((sq, B, (nq + 2 * nkv) * hd), dtype=torch.float32, device=dev)
    p_h = torch.empty((max(so, sd), B, cfg.hidden), dtype=torch.float32, device=dev)
    xo = torch.randn((B, cfg.hidden), device=dev).to(torch.bfloat16)

    def secondary_pass():
        for l in range(cfg.layers):
            lwl = lw.und[l]
            ops.gemm_splitk(x, lwl.qkv, p_qkv, sq) if sq > 1 else ops.gemm(x, lwl.qkv)
            ops.gemm_splitk(xo, lwl.o, p_h[:so], so) if so > 1 else ops.gemm(xo, lwl.o)
            ops.gemm_splitk(act, lwl.down, p_h[:sd], sd) if sd > 1 else ops.gemm(act, lwl.down)
    secondary = []
    if not lw.fp8 and B <= 64:
        secondary_pass()
        torch.cuda.synchronize()
        k0.record()
        for _ in range(reps):
            secondary_pass()
        k1.record()
        torch.cuda.synchronize()
        sec_us = k0.elapsed_time(k1) * 1e3 / (reps * cfg.layers * 3)
        sk = prof_kernel("gemm_skinny_kernelILi1ELi4ELi2E")
        ent = {"kernel": "gemm_skinny_kernel<1,4,2> (split-K QKV / o_proj / down_proj, 84 launches per step)", "bound": "hbm",
               "algorithmic_bytes_per_launch": int(sec_bytes), "avg_launch_us_replay": round(sec_us, 2),
               "frac_replay": round(sec_bytes / (sec_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
        if sk is not None:
            ent.update({"avg_launch_us": round(sk[0], 2), "frac": round(sec_bytes / (sk[0] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                        "launches_profiled": sk[1], "traffic": int(sk[2]) if sk[2] else None})
        secondary.append(ent)
    ak = prof_kernel("attn_kernelILi128E")
    if ak is not None:
        ctx_mid = ctx + 8 + 256                    # the profile is a 512-step decode after 8 warm-up steps: average context
        kv_layer = B * ctx_mid * 2 * nkv * hd * 2
        secondary.append({"kernel": "attn_kernel<128> (split-KV decode attention, one launch per layer; + attn_combine)", "bound": "hbm",
                          "algorithmic_bytes_per_launch": int(kv_layer), "avg_launch_us": round(ak[0], 2),
                          "frac": round(kv_layer / (ak[0] * 1e-6) / 1e9 / PEAK_HBM_GBS, 4), "launches_profiled": ak[1]})
    small = []
    for sub, label in (("residual_rmsnorm_kernel", "residual_rmsnorm"), ("qkv_post_kernel", "qkv_post"), ("attn_combine_kernel", "attn_combine")):
        v = prof_kernel(sub)
        if v is not None:
            small.append({"kernel": label, "avg_launch_us": round(v[0], 2), "launches_profiled": v[1]})

    # ---- ViT encode (MFMA-bound leg of the prefill): B x 448x448 through the SigLIP tower + connector
    def vit_leg(images_v, reps=10):
        Bv = len(images_v)
        gi_v, _, _ = model.prepare_vit_images([0] * Bv, [0] * Bv, images_v, lambda x: x, new_token_ids)
        px = gi_v["packed_vit_tokens"].to(dev)          # the patch tokens, or (device patchify) the images: resident either way
        pos_v = gi_v["packed_vit_position_ids"].to(dev)
        model.encode_vit(px, pos_v, gi_v["vit_token_seqlens"])
        torch.cuda.synchronize()
        v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        v0.record()
        for _ in range(reps):
            model.encode_vit(px, pos_v, gi_v["vit_token_seqlens"])
        v1.record()
        torch.cuda.synchronize()
        vit_ms = v0.elapsed_time(v1) / reps
        n_tok = int(gi_v["vit_token_seqlens"].sum())
        vh, vi = cfg.vit_hidden, cfg.vit_inter
        flops = 2 * n_tok * (cfg.vit_layers * (4 * vh * vh + 2 * vh * vi) + 3 * cfg.patch ** 2 * vh) \
            + cfg.vit_layers * 4 * (n_tok // Bv) ** 2 * vh * Bv + 2 * n_tok * (vh * cfg.hidden + cfg.hidden * cfg.hidden)
        return {"images_per_s": round(Bv / (vit_ms * 1e-3), 1), "ms_per_batch": round(vit_ms, 3), "batch": Bv,
                "tflops": round(flops / (vit_ms * 1e-3) / 1e12, 1), "mfma_frac_of_2500": round(flops / (vit_ms * 1e-3) / 2.5e15, 4),
                # the other roofline, for completeness (BASELINE.json pairs ViT with HBM): weights once + pixel input + output
                "hbm_frac_of_8000": round((0.79e9 + n_tok * (3 * cfg.patch ** 2 * 4 + cfg.hidden * 2)) / (vit_ms * 1e-3) / 8e12, 5),
                "patchify": "device (umv_patchify_f32_bf16, inside the timed region)" if getattr(model, "device_patchify", False) else "host (before the timed region)",
                "note": "ViT tower + connector, 1024 patches/image; MFMA-bound (intensity ~700 flop/B), hd-72 attention included"}
    vit = vit_leg(images) if args.config == "full" and not args.no_vit else None
    if vit is not None:
        vit["mfma_counters"] = stage_counters("vit")
        vit["roofline"] = mfma_roofline("vit", B * (img_hw // cfg.patch) ** 2, cfg.vit_inter, cfg.vit_hidden, False, "SigLIP fc1 (bias + GELU epilogue not in the timed call)")

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed
    # ---- the reference's DEFAULT decode mode (interactive_vqa_inferencer.py:58-71: do_sample=True, temperature 1.0; bagel.py:1297-1299):
    # same batch / context; the draw is a Gumbel-max over bf16(logit / T) in the lm_head epilogue (umv_gemm_args.sample_temperature), finished by
    # the same step-end kernel as the greedy step.  Never the headline value (BASELINE.json's metric is greedy).
    sampled = None
    if args.config == "full" and not args.no_sampled and not lw.fp8:
        ls = decode_leg(model, do_sample=True, gather="ids")
        sampled = {"tokens_per_s": round(world * B * args.steps / ls["elapsed"], 2), "ms_per_step": round(ls["elapsed"] * 1e3 / args.steps, 4),
                   "mode": "do_sample=True, temperature=1.0 (the reference's default, interactive_vqa_inferencer.py:58-71)",
                   "step_tail": "lm_head GEMM with the Gumbel-max keys in its epilogue (sample_temperature) -> umv_decode_step_end_argmax: the greedy step's two "
                                "launches (round 4 / early round 5: lm_head -> umv_sample_bf16 -> umv_decode_step_end, 3.29 ms per step)",
                   "greedy_ms_per_step": round(ms_per_step, 4)}
        ls = None
        torch.cuda.empty_cache()
    out = {
        "metric": f"VQA greedy decode tokens/s, UniMedVL-14B (BAGEL-7B-MoT dims), batch {B} x 448x448 per GPU",
        "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if not lw.fp8 else "bf16 (e4m3 weights, power-of-two channel scales)", "data": "synthetic (random N(0,0.02^2) weights at assumed 14B dims, synthetic images, random token ids)",
        "config": {"workload": (("configs[3]: medical-report generation, 32 samples per GPU (batch 256 over 8 GPUs), ViT encode + "
                                 "prefill + long greedy decode" if args.workload == "configs3" else
                                 "configs[1]: UniMedVL-14B bf16 VQA greedy decode, batch=8 448x448, 1xMI355X") if not lw.fp8 else
                                "configs[4]-style: UniMedVL-14B fp8-weight VQA greedy decode, batch=8 448x448 per GPU")
                               if args.config == "full" else "tiny smoke config",
                   "c1_gather": args.gather if world > 1 else None,
                   "batch_per_gpu": B, "context_tokens": ctx, "image": f"{img_hw}x{img_hw}", "prompt_tokens": prompt_len,
                   "parallelism": f"dp{world}", "cpu_binding": cpu_binding, "decode": "hipGraph" if not args.no_graph else "eager",
                   "prefill_s": round(t_prefill, 3), "prefill_cold_s": round(leg["t_prefill_cold"], 3),
                   "prefill_images_per_s": round(world * B / t_prefill, 1),
                   "weights_init_s": round(t_load, 1), "gpu_ms_per_step": round(gpu_ms / args.steps, 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": ("gemm_skinny8_kernel" if lw.fp8 else "gemm_skinny_kernel") + ("<1,2,4>" if B <= 16 else "<2,4,2>" if B <= 32 else " / tiled 128x64")
                               + " (28 gate/up SwiGLU GEMMs + lm_head per step)",
                     "avg_launch_us": round(avg_us, 2), "avg_launch_us_source": ("in-graph kernel, rocprofv3 average (profiles/roofline_profile_latest.json)"
                                                                                 if dom is not None else "isolated back-to-back replay, HIP events"),
                     "avg_launch_us_replay": round(avg_us_replay, 2), "profile_status": prof_status,
                     "secondary": secondary, "latency_floor_kernels": small,
                     "algorithmic_bytes_per_launch": int(bytes_per_launch),
                     "step_algorithmic_GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                     "step_frac_of_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                     # the other roofline, for completeness (BASELINE.json pairs decode with MFMA): at B rows per weight byte
                     # the MFMA fraction is capped near B / 310 by the HBM roofline
                     "step_mfma_frac_of_2500": round(B * (2 * lw.decode_weight_bytes() / (1 if lw.fp8 else 2)
                                                          + 4 * cfg.layers * cfg.heads * cfg.head_dim * ctx)
                                                     / (ms_per_step * 1e-3) / 2.5e15, 5)},
    }
    if sampled is not None:
        out["decode_sampled"] = sampled
    if vit is not None:
        out["vit_encode"] = vit
    if args.config == "full" and args.workload == "configs1" and not args.no_report and not lw.fp8:
        # extra leg, BASELINE.json configs[3]: 32 samples per GPU (batch 256 over 8 GPUs), ViT encode + prefill of
        # 448x448 + 128-token prompts (context 1156), then 512 greedy decode steps; never the headline value
        del sess, cache
        leg = None
        torch.cuda.empty_cache()
        B3, P3 = 32, 128
        g3 = torch.Generator().manual_seed(4321 + rank)
        prompts3 = [torch.randint(min(1000, hi // 2), hi, (P3,), generator=g3).tolist() for _ in range(B3)]
        images3 = [synth_image(img_hw, img_hw, 5000 + 1000 * rank + i) for i in range(B3)]
        l3 = decode_leg(model, B=B3, prompts=prompts3, images=images3, prompt_len=P3, steps=args.report_steps, warmup=args.warmup,
                        gather="ids")
        ms3 = l3["elapsed"] * 1e3 / args.report_steps
        sb3 = lw.decode_weight_bytes() + B3 * (l3["ctx"] + args.warmup + args.report_steps / 2) * kv_tok + B3 * kv_tok + B3 * cfg.vocab * 2
        out["report_b32"] = {
            "workload": "configs[3]: 32 samples per GPU, 448x448 + 128-token prompt (context 1156), ViT encode + prefill + "
                        f"{args.report_steps} greedy decode steps",
            "tokens_per_s": round(world * B3 * args.report_steps / l3["elapsed"], 2), "ms_per_step": round(ms3, 4),
            "decode_steps": args.report_steps, "batch_per_gpu": B3, "context_tokens": l3["ctx"],
            "vit_prefill_s": round(l3["t_prefill"], 3), "vit_prefill_cold_s": round(l3["t_prefill_cold"], 3),
            "vit_prefill_images_per_s": round(world * B3 / l3["t_prefill"], 1),
            "end_to_end_s": round(l3["t_prefill"] + l3["elapsed"], 3),
            "end_to_end_reports_per_s": round(world * B3 / (l3["t_prefill"] + l3["elapsed"]), 3),
            "step_algorithmic_GBps": round(sb3 / (ms3 * 1e-3) / 1e9, 1),
            "step_frac_of_peak": round(sb3 / (ms3 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        l3 = None
        torch.cuda.empty_cache()
        sess = cache = None
        if not args.no_vit:      # the tower where its GEMMs fill the chip: the 32 images of one GPU's configs[3] shard in one pass
            out["vit_encode_b32"] = vit_leg(images3)
            torch.cuda.empty_cache()
    if want_t2i:
        del sess, cache
        torch.cuda.empty_cache()
        if args.config == "full":
            out["t2i"] = run_t2i(model, cfg, dev, rank, world, dist, num_timesteps=args.t2i_steps)
            out["t2i"]["mfma_counters"] = stage_counters("t2i")
            out["t2i"]["roofline"] = mfma_roofline("t2i", 2 * out["t2i"]["batch_per_gpu"] * 256, 2 * cfg.inter, cfg.hidden, True, "gen-expert gate/up of a guided flow pass (2 contexts x batch x 256 latent tokens)")
            out["prefill_mfma_counters"] = stage_counters("prefill")
            out["prefill_roofline"] = mfma_roofline("prefill", B * ((img_hw // cfg.patch) ** 2 + 2), 2 * cfg.inter, cfg.hidden, True, "und-expert gate/up of the image-span prefill")
            if not args.no_edit:
                torch.cuda.empty_cache()
                try:
                    out["edit"] = run_edit(model, cfg, dev, rank, world, dist, num_timesteps=args.t2i_steps)
                    out["edit"]["mfma_counters"] = stage_counters("edit")
                    out["edit"]["roofline"] = mfma_roofline("edit", 3 * out["edit"]["batch_per_gpu"] * 1024, 2 * cfg.inter, cfg.hidden, True,
                                                            "gen-expert gate/up of a guided edit step (3 contexts x batch x 1024 latent tokens)")
                except Exception as e:      # an extra leg must never take the bench line down
                    if dist is not None:
                        raise
                    out["edit"] = {"failed": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        else:
            out["t2i"] = run_t2i(model, cfg, dev, rank, world, dist, batch=2, hw=64, prompt_len=8, num_timesteps=6)
    if not args.no_fp8 and not lw.fp8 and args.config == "full":
        # extra leg (BASELINE.json configs[4]): the same workload on e4m3 LLM weights; never the headline value
        del model, lw
        try:
            del sess, cache
        except NameError:
            pass
        leg = None
        torch.cuda.empty_cache()
        cfg8 = UniMedVLConfig.from_dict(cfg.to_dict())
        cfg8.llm_weight_dtype = "fp8"
        cfg8.llm_act_dtype = "fp8" if args.fp8_act else "bf16"   # W8A8 on the fp8 matrix instruction for prefill / flow passes
        model8 = Bagel(cfg8, random_getter(cfg8, dev, seed=1234), device=dev, visual_gen=want_t2i, visual_und=True)
        l8 = decode_leg(model8)
        w8 = model8.language_model.w
        ms8 = l8["elapsed"] * 1e3 / args.steps
        sb8 = w8.decode_weight_bytes() + B * (l8["ctx"] + args.warmup + args.steps / 2) * kv_tok + B * kv_tok + B * cfg.vocab * 2
        out["decode_fp8_weights"] = {
            "tokens_per_s": round(world * B * args.steps / l8["elapsed"], 2), "ms_per_step": round(ms8, 4),
            "weights": "OCP e4m3, one power-of-two scale per output channel; bf16 activations, fp32 accumulate",
            "weight_bytes_per_step": int(w8.decode_weight_bytes()),
            "step_algorithmic_GBps": round(sb8 / (ms8 * 1e-3) / 1e9, 1),
            "step_frac_of_peak": round(sb8 / (ms8 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "workload": "configs[4]-style: same batch / context as the headline run, fp8 weights",
            "parity": "umv_gemm_fp8w == umv_gemm_bf16 on the dequantised weights bit for bit; engine vs the CPU oracle on the "
                      "dequantised weights at the bf16 tolerances (tests/test_fp8_gpu.py); split-K decode mode on"}
        out["decode_fp8_weights"]["prefill_s"] = round(l8["t_prefill"], 3)
        out["decode_fp8_weights"]["activations"] = ("decode steps: bf16; prefill / flow passes: per-row e4m3 on the fp8 matrix "
                                                    "instruction (W8A8, umv_gemm_fp8a8w)") if args.fp8_act else "bf16 everywhere"
        if want_t2i:   # the T2I half of configs[4]'s mixed batch on the same fp8 model
            l8["sess"] = l8["cache"] = None
            t8 = run_t2i(model8, cfg8, dev, rank, world, dist, num_timesteps=args.t2i_steps)
            out["decode_fp8_weights"]["t2i_images_per_s_same_model"] = t8["images_per_s"]
            out["decode_fp8_weights"]["t2i_llm_tflops"] = t8["llm_tflops"]
            try:
                out["mixed_fp8"] = run_mixed(model8, cfg8, dev, rank, world, dist, num_timesteps=args.t2i_steps)
                # what the two kinds of work cost back to back on the same model (the separate legs above)
                sep = out["mixed_fp8"]["vqa_tokens"] / max(out["decode_fp8_weights"]["tokens_per_s"], 1e-9) + \
                    out["mixed_fp8"]["images"] / max(t8["images_per_s"], 1e-9) + l8["t_prefill"]
                out["mixed_fp8"]["separate_legs_s"] = round(sep, 3)
            except Exception as e:   # an extra leg must never take the bench line down
                out["mixed_fp8"] = {"failed": f"{type(e).__name__}: {e}"}
        del model8, l8
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and args.config == "full" and not args.no_vae and not args.no_t2i:
        try:
            out["vae"] = run_vae(cfg, dev)
        except Exception as e:      # an extra leg must never take the bench line down
            out["vae"] = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and args.config == "full" and not args.no_load_path:
        try:
            out["kv_growth"] = run_kv_growth(cfg, dev)
        except Exception as e:      # an extra leg must never take the bench line down
            out["kv_growth"] = {"failed": f"{type(e).__name__}: {e}"}
        try:
            out["load_path"] = run_load_path(cfg, dev)
        except Exception as e:      # an extra leg must never take the bench line down
            out["load_path"] = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "full":
        try:
            out["cpu_baseline"] = cpu_baseline(B, ctx, cfg.layers, cfg.vit_layers)
        except Exception as e:  # the baseline leg must never take the bench line down
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
        out["cpu_baseline"]["host"] = cpu_info()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.dist.destroy_process_group()


if __name__ == "__main__":
    main()
