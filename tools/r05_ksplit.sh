#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_ksplit; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "k_split or w4 or lean" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
for k in 0 1; do for sh in "2048,3584,18944" "1024,3584,18944" "516,3584,18944" "258,3584,18944" "1026,3584,18944"; do
  echo "KSPLIT=$k $(UMV_GEMM_KSPLIT=$k SHAPE=$sh SECONDS=0.6 timeout 120 python tools/gemm_power.py 2>&1 | tail -1)" | tee -a $O/shapes.txt
done; done
for rep in 1 2; do for k in 0 1; do for st in "t2i 4" "t2i 1" "prefill 1"; do
  echo "rep $rep KSPLIT=$k $st: $(UMV_GEMM_KSPLIT=$k REPS=20 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stages.txt
done; done; done
