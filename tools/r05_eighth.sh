#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_eighth; mkdir -p $O
for rep in 1 2; do for w in 1 2; do for st in "vit 8" "vit 32"; do
  echo "rep $rep W4=$w $st: $(UMV_GEMM_W4=$w REPS=20 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/vit_w4.txt
done; done; done
for b in 1 2 4; do echo "t2i B=$b: $(timeout 600 python tools/stage_profile.py t2i $b 2>&1 | tail -1)" | tee -a $O/t2i_batch.txt; done
echo "edit B=1: $(timeout 900 python tools/stage_profile.py edit 1 2>&1 | tail -1)" | tee -a $O/t2i_batch.txt
