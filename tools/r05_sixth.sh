#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_sixth; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "lean" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
SECONDS=1 W4_ARMS=0,2 SHAPES="8192,3456,1152;8192,4304,1152;8192,1152,4304;8192,1152,1152;32768,3456,1152;32768,4304,1152;32768,1152,4304;32768,1152,1152" timeout 900 python tools/w4_ab.py time > $O/w4_vit_shapes.txt 2>&1
cat $O/w4_vit_shapes.txt
for w in 1 2; do for st in "vit 8" "vit 32"; do
  echo "W4=$w $st: $(UMV_GEMM_W4=$w REPS=10 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stage_w4.txt
done; done
