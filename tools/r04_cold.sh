#!/bin/bash
for rep in 1 2; do
for pf in "" 1152 "1152,4304" "1152,4304,3456"; do echo -n "vit PREFETCH_N=$pf : "; PREFETCH_N=$pf REPS=30 python tools/stage_profile.py vit 2>&1 | tail -1; done
done
