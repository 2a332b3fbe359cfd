#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_gemm; mkdir -p $O
SH="8192,8192,8192;4096,4096,4096;2048,4608,3584;2048,3584,18944;2048,3584,3584;8208,4608,3584;8208,3584,3584;8192,1152,4304;8192,1152,1152;8192,4304,1152;1000,1152,1160;300,520,1096;700,3584,96;515,1152,4304"
for T in ${TILES:-266 366 268 368 384 484 270 370}; do
  echo "== tile $T" | tee -a $O/ab_px.txt
  UMV_GEMM_TILE=$T SHAPES="$SH" timeout 300 python tools/gemm_ab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_px.txt
done
for T in 266 366; do
  UMV_GEMM_TILE=$T SECONDS=3 timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tee -a $O/power_px.txt
  UMV_GEMM_TILE=$T SECONDS=3 SHAPE=2048,37888,3584,1 timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tee -a $O/power_px.txt
done
for T in ${BTILES:-266 366}; do
  echo "== bench tile $T" | tee -a $O/bench_px.txt
  UMV_GEMM_TILE=$T timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_px.txt
  UMV_GEMM_TILE=$T timeout 300 python tools/gemm_bench.py --flow 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_px.txt
done
