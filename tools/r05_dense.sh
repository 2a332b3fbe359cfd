#!/bin/bash
# dense (token, head) packing of the q-tiles (G = 7: 16 pairs per tile instead of 14): same bits, fewer tiles
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_dense; mkdir -p $O; rm -f $O/*.txt
for cfg in "UMV_ATTN_DENSE=0" "UMV_ATTN_DENSE=1" "UMV_ATTN_DENSE=1 UMV_ATTN_LAZY=0" "UMV_ATTN_SHARED=0"; do
  echo "== $cfg" >> $O/ab.txt
  env $cfg ATTN_AB_REF=1 timeout 300 python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || echo "FAILED rc=$?" >> $O/ab.txt
done
cat $O/ab.txt | cut -c1-175
timeout 900 python -m pytest tests/test_kernel_branches_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "attn" 2>&1 | tail -4
