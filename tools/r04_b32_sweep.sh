#!/bin/bash
# B = 17..32 decode step: the 128 x 64 full-line tile against the weight-streaming kernel on gate/up (tuning only)
F="--no-t2i --no-vit --no-vae --no-cpu-baseline --no-load-path --no-fp8 --no-report --steps 64 --warmup 8"
run() { echo -n "B=$B $* : "; env "$@" python bench.py $F --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for B in 32 24 17; do
run X=0; run UMV_GEMM_M64_MIN=17; run X=0; run UMV_GEMM_M64_MIN=17
done
