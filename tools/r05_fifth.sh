#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fifth; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "gemm" > $O/tests.txt 2>&1
tail -15 $O/tests.txt
for l in 0 1; do
  for st in "t2i 4" "prefill 8" "vit 8" "vit 32"; do
    echo "LEAN=$l $st: $(UMV_GEMM_LEAN_EPI=$l REPS=10 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stage_lean.txt
  done
done
UMV_GEMM_ABLATIONS=1 python -m unimedvl_amd.build > $O/abl_build.txt 2>&1 || tail -20 $O/abl_build.txt
for S in 8192,3456,1152 2048,37888,3584; do
  UMV_GEMM_ABLATIONS=1 UMV_GEMM_TILE=94662 SHAPE=$S timeout 300 python tools/w4_trace.py 2>&1 | grep -v amdgpu.ids | grep -A12 "tile level" | tee -a $O/w4_tile_trace.txt
done
