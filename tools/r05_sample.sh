#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_sample; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_decode_engine_gpu.py tests/test_serving_gpu.py tests/test_inferencer_gpu.py -q -m gpu -k "sampl or decode or chat or inferencer or batch" > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 900 python bench.py --steps 64 --warmup 8 --no-t2i --no-fp8 --no-report --no-vit --no-load-path --no-cpu-baseline > $O/bench.json 2> $O/err.txt
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("greedy", d["ms_per_step"], "sampled", d.get("decode_sampled"))
PY
