#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_n1152; mkdir -p $O
ARMS=466,468,4384 SHAPES="8208,4608,3584;1024,37888,3584,swiglu;32768,4304,1152;2064,37888,3584,swiglu;16500,1280,18944;12288,3584,18944;8192,4304,1152;272,37888,3584,swiglu" SECONDS=0.7 timeout 1500 python tools/tile_arms.py 2>&1 | grep -v amdgpu | tee $O/arms3.txt
