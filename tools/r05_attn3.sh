#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_attn3; mkdir -p $O
python tools/attn_diff.py 2>&1 | grep -v amdgpu | tee $O/diff.txt
for p in 0 1; do for c in llm vit; do echo "PIPE=$p $(UMV_ATTN_PIPE=$p CASE=$c python tools/attn_trace.py | head -1)" | tee -a $O/time.txt; done; done
for p in 0 1; do for st in "vit 8" "vit 32" "prefill 8"; do echo "PIPE=$p $st: $(UMV_ATTN_PIPE=$p REPS=20 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/time.txt; done; done
timeout 1500 python -m pytest tests/test_kernel_branches_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py -q -m gpu -k "attn or batch_independence or vit or t2i or vqa" > $O/tests.txt 2>&1
tail -4 $O/tests.txt
