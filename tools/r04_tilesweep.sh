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
for SK in "3,4,4" "1,4,4" "1,1,4" "3,1,4" "2,2,4" "3,4,8" "1,2,4"; do
  echo "== SPLITK=$SK" | tee -a $O/splitk.txt
  UMV_DECODE_SPLITK=$SK timeout 600 python bench.py --no-cpu-baseline --no-t2i --no-vit --no-load-path --no-fp8 --no-report 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/splitk.txt
done
