#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fourth; mkdir -p $O
tools/bin/store_bench 2>&1 | grep -v amdgpu.ids | tee $O/store_bench.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py tests/test_fp8_mfma_gpu.py tests/test_engine_gpu.py tests/test_gemm_decode_gpu.py tests/test_splitk_gpu.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 900 python bench.py --steps 32 --warmup 8 --no-t2i --no-fp8 --no-report --no-vit --no-load-path --no-sampled > $O/bench_cpu.json 2> $O/bench_err.txt
python - <<PY
import json
d = json.loads(open("$O/bench_cpu.json").read().strip().splitlines()[-1])
c = d["cpu_baseline"]
print(c["value"], c["cores"], c["runs"]); print(c["sample"]); print(c.get("edit"), c.get("vision_failed"))
PY
