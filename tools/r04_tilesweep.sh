#!/bin/bash
# sustained (3 s loop, power-limited clocks) tile sweep on the model's MFMA-bound GEMM shapes
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_tiles; mkdir -p $O; rm -f $O/sweep.txt
for SH in 2048,4608,3584 2048,3584,3584 2048,3584,18944 2048,37888,3584,1 8208,4608,3584 8208,3584,3584 8208,3584,18944 8192,3456,1152 8192,1152,1152 8192,4304,1152 8192,1152,4304 32768,3456,1152 32768,1152,4304; do
  for T in 266 268 384 288 270; do
    XL=1; [ "$T" = "288" ] && XL=2
    SHAPE=$SH SECONDS=2 UMV_GEMM_TILE=$T UMV_GEMM_XLINE=$XL timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/sweep.txt
  done
  SHAPE=$SH SECONDS=2 UMV_GEMM_TILE=288 UMV_GEMM_XLINE=1 timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/tile=  288/tile=288old/' | tee -a $O/sweep.txt
  SHAPE=$SH SECONDS=2 timeout 120 python tools/gemm_power.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/sweep.txt
done
