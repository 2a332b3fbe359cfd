#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_policy_e2e; mkdir -p $O
for rep in 1 2; do for m in 0 1; do for st in "vit 8" "vit 32" "t2i 4" "edit 4"; do
  echo "rep $rep MODEL=$m $st: $(UMV_GEMM_TILE_MODEL=$m REPS=20 STEPS=20 timeout 900 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/ab.txt
done; done; done
