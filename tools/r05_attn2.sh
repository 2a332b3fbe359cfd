#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_attn2; mkdir -p $O
for c in llm vit; do CASE=$c timeout 300 python tools/attn_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $O/time.txt; done
timeout 2400 python -m pytest tests/test_kernel_branches_gpu.py tests/test_kernels_gpu.py tests/test_attn_prefill32_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py -q -m gpu -k "not gemm" > $O/tests.txt 2>&1
tail -30 $O/tests.txt
for st in "vit 8" "vit 32" "prefill 8"; do echo "$st: $(REPS=10 timeout 600 python tools/stage_profile.py $st 2>&1 | tail -1)" | tee -a $O/stage.txt; done
