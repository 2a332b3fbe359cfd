#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_gemm; mkdir -p $O
for T in ${TILES:-266 9666 9667 9668 9663}; do
  UMV_GEMM_TILE=$T SECONDS=3 timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tee -a $O/power2.txt
done
