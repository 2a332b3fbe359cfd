#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0; export HSA_ENABLE_COREDUMP=0
O=gpurun_out/r05_persist; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "w4 or lean or ragged_and_unaligned or gemm_tiled" > $O/tests.txt 2>&1
tail -6 $O/tests.txt
for p in 0 1; do
  for sh in "8192,3456,1152" "8192,4304,1152" "2048,37888,3584,swiglu" "8208,37888,3584,swiglu" "8208,3584,18944" "8192,8192,8192"; do
    echo "PERSIST=$p $(UMV_GEMM_W4_PERSIST=$p SHAPE=$sh SECONDS=1 python tools/gemm_power.py 2>&1 | tail -1)" | tee -a $O/shapes.txt
  done
done
for p in 0 1; do for st in "vit 8" "vit 32" "prefill 8" "t2i 4"; do echo "PERSIST=$p $st: $(UMV_GEMM_W4_PERSIST=$p REPS=20 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stages.txt; done; done
