#!/bin/bash
# round 5, first GPU call of the session: edit-pipeline tests, then the 4-wave tiles end to end (UMV_GEMM_W4=0 / 1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_first; mkdir -p $O
timeout 900 python -m pytest tests/test_fullwidth_gpu.py -x -q -s -m gpu -k "edit_pipeline" > $O/edit_tests.txt 2>&1
tail -5 $O/edit_tests.txt
for w in 0 1; do
  for st in "t2i 4" "prefill 8" "vit 8" "vit 32"; do
    echo "W4=$w $st: $(UMV_GEMM_W4=$w REPS=10 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stage_w4.txt
  done
done
SECONDS=1 SHAPES="8192,3456,1152;8192,4304,1152;8192,1152,4304;8192,1152,1152;2064,4608,3584;2048,3584,18944;2048,37888,3584,swiglu;8208,37888,3584,swiglu;8208,3584,3584" timeout 900 python tools/w4_ab.py time > $O/w4_ab_time.txt 2>&1
cat $O/w4_ab_time.txt
