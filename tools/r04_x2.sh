#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_gemm; mkdir -p $O; rm -f $O/x2.txt
SH="8192,1152,1152;8192,1152,4304;2048,4608,3584;515,1152,1160;130,4608,3584;300,288,96"
for T in 288 388; do
  echo "== tile $T" | tee -a $O/x2.txt
  UMV_GEMM_TILE=$T UMV_GEMM_XLINE=2 SHAPES="$SH" timeout 300 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/x2.txt
done
for V in "UMV_GEMM_XLINE=1" "UMV_GEMM_XLINE=2" "UMV_GEMM_XLINE=1 UMV_GEMM_RASTER=2" "UMV_GEMM_XLINE=1 UMV_GEMM_RASTER=8" "UMV_GEMM_XLINE=1 UMV_GEMM_RASTER=1"; do
  echo "== $V" | tee -a $O/x2.txt
  env $V timeout 300 python tools/stage_profile.py t2i 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/x2.txt
  env $V REPS=10 timeout 300 python tools/stage_profile.py prefill 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/x2.txt
  env $V REPS=20 timeout 300 python tools/stage_profile.py vit 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/x2.txt
done
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_serving_gpu.py tests/test_splitk_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/x2.txt
