#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_gemm; mkdir -p $O
for X in "" "266" "270" "266,270" "266,268,384,270"; do
  echo "== XLINE=$X" | tee -a $O/xline.txt
  UMV_GEMM_XLINE=$X timeout 300 python tools/stage_profile.py t2i 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/xline.txt
  UMV_GEMM_XLINE=$X REPS=10 timeout 300 python tools/stage_profile.py prefill 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/xline.txt
  UMV_GEMM_XLINE=$X REPS=20 timeout 300 python tools/stage_profile.py vit 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/xline.txt
done
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernel_branches_gpu.py tests/test_splitk_gpu.py tests/test_edge_gpu.py -x -q 2>&1 | tail -5 | tee -a $O/xline_tests.txt
