#!/bin/bash
F="--no-t2i --no-vit --no-vae --no-cpu-baseline --no-load-path --no-fp8 --no-report --steps 64 --warmup 8"
run() { echo -n "B=$B $* : "; env "$@" python bench.py $F --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" | cut -c1-200; }
for B in 128 96; do for sk in 6,8,8 6,8,16 7,9,18 12,8,16 4,4,8 6,4,16 9,9,9; do run UMV_DECODE_SPLITK=$sk; done; done
