#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_attn_trace; mkdir -p $O
UMV_ATTN_TRACE=1 python -m unimedvl_amd.build > $O/build.txt 2>&1 || tail -20 $O/build.txt
for c in llm vit; do CASE=$c UMV_ATTN_TRACE=1 timeout 300 python tools/attn_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $O/trace.txt; done
