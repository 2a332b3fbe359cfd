#!/bin/bash
# the 8 x 34-token question prefill (272 rows): which tile for each of the four GEMM shapes
for sh in 272,37888,3584,1 272,4608,3584 272,3584,3584 272,3584,18944 130,37888,3584,1 130,3584,18944 1040,37888,3584,1 1040,3584,18944; do
for t in 0 384 268 270 266 64 288; do SHAPE=$sh UMV_GEMM_TILE=$t python tools/gemm_bench.py 2>/dev/null | tail -1; done; echo; done
