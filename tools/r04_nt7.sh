#!/bin/bash
# 7 n-tiles per workgroup x 8 K splits = 256 workgroups for the N = 3584 decode GEMMs (o_proj / down_proj): trial
F="--no-t2i --no-vit --no-vae --no-cpu-baseline --no-load-path --no-fp8 --no-report --steps 128 --warmup 8"
run() { echo -n "B=$B $* : "; env "$@" python bench.py $F --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
B=8
run X=0
run UMV_SKINNY_NT7=8 UMV_DECODE_SPLITK=3,4,8
run UMV_SKINNY_NT7=8 UMV_DECODE_SPLITK=3,8,8
run UMV_SKINNY_NT7=0 UMV_DECODE_SPLITK=3,4,8
run UMV_SKINNY_NT7=4 UMV_DECODE_SPLITK=3,4,4
B=32
run X=0
run UMV_SKINNY_NT7=8 UMV_DECODE_SPLITK=3,4,8
run UMV_SKINNY_NT7=8 UMV_DECODE_SPLITK=3,8,8
done
