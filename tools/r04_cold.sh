#!/bin/bash
for t in 0 384 268 266; do echo "== UMV_GEMM_TILE=$t"; UMV_GEMM_TILE=$t python tools/vit_gemm_cold.py 8192 2>&1 | grep -E "^(out|fc2) "; UMV_GEMM_TILE=$t python tools/vit_gemm_cold.py 2048 2>&1 | grep -E "^(llm_qkv) "; done
