#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_n1152; mkdir -p $O
ARMS=288,4384,466,468 SHAPES="2064,4608,3584;1032,4608,3584;516,4608,3584;4104,4608,3584;12288,4608,3584;2048,1152,4304;4096,1152,4304;16384,1152,4304" SECONDS=0.7 timeout 1500 python tools/tile_arms.py 2>&1 | grep -v amdgpu | tee $O/arms2.txt
