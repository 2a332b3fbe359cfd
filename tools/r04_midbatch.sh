#!/bin/bash
# A/B of the full-line x staging on the 33..64-row and 65..128-row decode GEMM paths (tuning only)
mkdir -p gpurun_out
F="--no-t2i --no-vit --no-vae --no-cpu-baseline --no-load-path --no-fp8 --no-report --steps 64 --warmup 8"
run() { echo "== B=$B $*"; env "$@" python bench.py $F --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
B=64;  run X=0; run UMV_SPLITK_M64=1; run UMV_SPLITK_TILED_MIN=33
B=40;  run X=0; run UMV_SPLITK_M64=1; run UMV_SPLITK_TILED_MIN=33
done
