#!/bin/bash
F="--no-t2i --no-vit --no-vae --no-cpu-baseline --no-load-path --no-fp8 --no-report --steps 64 --warmup 8"
run() { echo -n "B=$B $* : "; env "$@" python bench.py $F --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" | cut -c1-200; }
for rep in 1 2; do for B in 128 96 72; do run X=0; run UMV_GEMM_TILE=392; run UMV_GEMM_TILE=396; done; done
