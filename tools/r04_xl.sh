#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_xl; mkdir -p $O; rm -f $O/*_b.txt
for XL in 0 2; do
  echo "== SKINNY_XL=$XL B=8" | tee -a $O/skinny_b.txt
  UMV_SKINNY_XL=$XL timeout 300 python tools/skinny_bench.py 8 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/skinny_b.txt
  echo "== SKINNY_XL=$XL B=8 SPLIT" | tee -a $O/skinny_b.txt
  SPLIT=4 UMV_SKINNY_XL=$XL timeout 300 python tools/skinny_bench.py 8 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/skinny_b.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernel_branches_gpu.py tests/test_splitk_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/tests_b.txt
for XL in 1 2; do
  echo "== XL=$XL" | tee -a $O/bench_b.txt
  UMV_SKINNY_XL=$XL timeout 900 python bench.py --no-cpu-baseline --no-t2i --no-vit --no-load-path 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('report_b32',{}).get('tokens_per_s'), d.get('report_b32',{}).get('ms_per_step'), d.get('decode_fp8_weights',{}).get('tokens_per_s'))" | tee -a $O/bench_b.txt
done
