#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_n1152; mkdir -p $O
timeout 900 python -m pytest tests/test_kernel_branches_gpu.py -x -q -m gpu -k "w4 or lean" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
SHAPE=8192,4304,1152 SECONDS=1 python tools/gemm_power.py 2>&1 | tail -1 | tee -a $O/fc1.txt
SHAPE=32768,4304,1152 SECONDS=1 python tools/gemm_power.py 2>&1 | tail -1 | tee -a $O/fc1.txt
ARMS=288,4384,466 SHAPES="8192,1152,4304;8192,1152,1152;32768,1152,4304;32768,1152,1152" SECONDS=1 timeout 1200 python tools/tile_arms.py 2>&1 | grep -v amdgpu | tee $O/arms.txt
for st in "vit 8" "vit 32"; do echo "$st: $(REPS=20 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stages.txt; done
