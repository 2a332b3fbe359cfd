#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_policy_e2e; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "gemm" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for st in "vit 8" "vit 32" "prefill 8" "t2i 4" "t2i 1" "edit 4"; do echo "$st: $(REPS=20 STEPS=50 timeout 900 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stages.txt; done
