#!/bin/bash
# round 5: w4 policy (K >= 2048) - tests + row-count sweep of the LLM GEMM shapes, UMV_GEMM_W4=0 vs 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_second; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernel_branches_gpu.py tests/test_fullwidth_gpu.py -x -q -s -m gpu -k "edit_pipeline or w4 or ragged_and_unaligned or gemm_tiled" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
for M in 96 128 272 516 1032 4104; do
  S="$M,4608,3584;$M,3584,3584;$M,37888,3584,swiglu;$M,3584,18944"
  SECONDS=0.6 W4_ARMS=0,1 SHAPES="$S" timeout 600 python tools/w4_ab.py time >> $O/w4_rows.txt 2>&1
done
SECONDS=0.6 W4_ARMS=0,1 SHAPES="8208,4608,3584;2064,3584,3584;2064,3584,18944;8208,3584,18944" timeout 600 python tools/w4_ab.py time >> $O/w4_rows.txt 2>&1
cat $O/w4_rows.txt
