#!/bin/bash
# round 5: bench line with the new legs; then (ablation build on the box) tile-level trace of the 4-wave tile at short and long K
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_third; mkdir -p $O
timeout 1200 python bench.py --steps 64 --warmup 8 > $O/bench_line.json 2> $O/bench_err.txt
tail -c 600 $O/bench_err.txt
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "prefill", d["config"]["prefill_s"])
    for k in ("decode_sampled", "edit", "t2i", "vit_encode", "vit_encode_b32", "report_b32"):
        v = d.get(k) or {}
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, list)) and a not in ("note", "workload", "cfg", "mode", "step_tail")})
    c = d.get("cpu_baseline", {})
    print("cpu", c.get("value"), c.get("cores"), c.get("runs"), c.get("reference_estimate_tokens_per_s"))
except Exception as e:
    print("bench line unreadable", e)
PY
UMV_GEMM_ABLATIONS=1 python -m unimedvl_amd.build > $O/abl_build.txt 2>&1 || tail -20 $O/abl_build.txt
for S in 8192,3456,1152 8192,8192,8192 2048,37888,3584; do
  UMV_GEMM_ABLATIONS=1 UMV_GEMM_TILE=94662 SHAPE=$S timeout 300 python tools/w4_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $O/w4_tile_trace.txt
done
