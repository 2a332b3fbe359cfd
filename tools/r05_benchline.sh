#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout 1800 python bench.py > $O/r05_bench_line.json 2> $O/r05_bench_err.txt
tail -c 300 $O/r05_bench_err.txt
python - <<PY
import json
d = json.loads(open("$O/r05_bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "steps", d["steps"], "prefill", d["config"]["prefill_s"])
print("roofline frac", r["frac"], "traffic", r["traffic"], "status", r["profile_status"], "step_frac", r["step_frac_of_peak"])
for k in ("decode_sampled", "edit", "t2i", "vit_encode", "vit_encode_b32", "report_b32", "decode_fp8_weights"):
    v = d.get(k) or {}
    print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, list)) and a not in ("note", "workload", "cfg", "mode", "step_tail", "patchify", "parity", "weights", "activations")})
    if "roofline" in v: print("   roofline", {a: v["roofline"].get(a) for a in ("avg_launch_us", "frac", "avg_launch_us_in_leg", "frac_in_leg", "share_of_leg_gpu_time")})
    if v.get("mfma_counters"): print("   counters", v["mfma_counters"]["kernels"][:2])
print("prefill", d.get("prefill_roofline", {}).get("frac"), d.get("prefill_roofline", {}).get("frac_in_leg"))
c = d["cpu_baseline"]; print("cpu", c["value"], c.get("runs"))
PY
