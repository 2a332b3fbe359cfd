#!/bin/bash
# m-super-block size of the tiled GEMM above 16k rows (configs[3]: 32 images per GPU), re-swept on the full-line kernels
for rep in 1 2; do
for v in "" 0 4 8 16 32; do echo -n "prefill B=32 MSB=$v : "; UMV_GEMM_MSB=$v REPS=6 python tools/stage_profile.py prefill 32 2>&1 | tail -1; done
for v in "" 0 8 16 32; do echo -n "vit B=32 MSB=$v : "; UMV_GEMM_MSB=$v REPS=10 python tools/stage_profile.py vit 32 2>&1 | tail -1; done
done
