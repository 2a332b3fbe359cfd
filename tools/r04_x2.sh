#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_qkv; mkdir -p $O; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_engine_gpu.py tests/test_edge_gpu.py tests/test_inferencer_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/tests.txt
for V in 0 1; do
  echo "== UMV_QKV_POST_VEC=$V" | tee -a $O/ab.txt
  UMV_QKV_POST_VEC=$V timeout 300 python tools/stage_profile.py t2i 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ab.txt
  UMV_QKV_POST_VEC=$V REPS=10 timeout 300 python tools/stage_profile.py prefill 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ab.txt
done
