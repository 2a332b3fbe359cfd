#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_seventh; mkdir -p $O
timeout 1500 python -m pytest tests/test_fullwidth_gpu.py -x -q -s -m gpu -k "continuous_batcher" > $O/serving_test.txt 2>&1
grep -a "continuous batcher\|passed\|failed\|Error" $O/serving_test.txt | tail -8
timeout 1500 python bench.py --steps 64 --warmup 8 > $O/bench_line.json 2> $O/bench_err.txt
tail -c 400 $O/bench_err.txt
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "prefill", d["config"]["prefill_s"], "step_frac", d["roofline"]["step_frac_of_peak"])
    for k in ("decode_sampled", "edit", "t2i", "vit_encode", "vit_encode_b32", "report_b32"):
        v = d.get(k) or {}
        print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, list)) and a not in ("note", "workload", "cfg", "mode", "step_tail", "patchify")})
        if "roofline" in v: print("   roofline", v["roofline"])
    print("prefill_roofline", d.get("prefill_roofline"))
    c = d.get("cpu_baseline", {})
    print("cpu", c.get("value"), c.get("cores"), c.get("runs")); print(c.get("sample")); print({k: c.get(k) for k in ("vit", "t2i", "edit", "vision_failed")})
except Exception as e:
    print("bench line unreadable", e)
PY
