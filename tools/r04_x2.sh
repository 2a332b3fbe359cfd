#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_tiles; mkdir -p $O; rm -f $O/t324.txt
SH="2048,3584,18944;2048,3584,3584;515,3584,1160;300,448,96"
for T in 268 324; do
  echo "== tile $T" | tee -a $O/t324.txt
  UMV_GEMM_TILE=$T SHAPES="$SH" timeout 300 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/t324.txt
done
for SHP in 2048,3584,18944 2048,3584,3584 1024,3584,18944; do
  for T in 268 324 270; do
    SHAPE=$SHP SECONDS=2 UMV_GEMM_TILE=$T timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/t324.txt
  done
done
