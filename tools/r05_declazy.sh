#!/bin/bash
# lazy softmax reference in the decode / short-prefill attention kernel (attn_kernel): A/B of the decode step, bit identity with the prefill kernels
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_declazy; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_kernel_branches_gpu.py tests/test_kernels_gpu.py tests/test_splitk_gpu.py -q -m gpu -k "attn" 2>&1 | tail -4 | tee $O/tests.txt
for lz in 0 1 0 1; do
  UMV_ATTN_LAZY=$lz timeout 900 python bench.py --steps 64 --warmup 8 --no-t2i --no-vit --no-load-path --no-cpu-baseline --no-sampled > $O/bench$lz.json 2> $O/err.txt
  python - <<PY | tee -a $O/ab.txt
import json
d = json.loads(open("$O/bench$lz.json").read().strip().splitlines()[-1])
print("UMV_ATTN_LAZY=$lz  B=8", d["ms_per_step"], " B=32", d.get("report_b32", {}).get("ms_per_step"), " fp8", d.get("decode_fp8_weights", {}).get("ms_per_step"))
PY
done
