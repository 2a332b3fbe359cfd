#!/bin/bash
# Round-4 tiled-GEMM sweep (inside gpurun): correctness + timing of the 32x32x16 variants next to the shipped tiles.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_gemm; mkdir -p $O
SH="8192,8192,8192;4096,4096,4096;2048,4608,3584;2048,3584,18944;2048,3584,3584;8208,4608,3584;8208,3584,3584;8192,1152,4304;8192,1152,1152;8192,4304,1152;1000,1152,1160"
for T in ${TILES:-266 566 268 568 384 684 270 570}; do
  echo "== tile $T" | tee -a $O/ab.txt
  UMV_GEMM_TILE=$T SHAPES="$SH" timeout 300 python tools/gemm_ab.py 2>&1 | tee -a $O/ab.txt
done
for T in ${BTILES:-266 566}; do
  echo "== bench tile $T" | tee -a $O/bench.txt
  UMV_GEMM_TILE=$T timeout 300 python tools/gemm_bench.py 2>&1 | tee -a $O/bench.txt
  UMV_GEMM_TILE=$T timeout 300 python tools/gemm_bench.py --flow 2>&1 | tee -a $O/bench.txt
done
for T in ${PTILES:-266 566}; do
  echo "== pmc tile $T 8192^3" | tee -a $O/pmc.txt
  UMV_GEMM_TILE=$T SHAPE=8192,8192,8192 timeout 900 bash tools/pmc_gemm.sh pmcg_$T 2>&1 | tee -a $O/pmc.txt
done
