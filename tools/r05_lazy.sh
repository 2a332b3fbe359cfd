#!/bin/bash
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_lazy; mkdir -p $O; rm -f $O/*.txt
echo "== UMV_ATTN_LAZY=0" >> $O/ab.txt
ATTN_AB_REF=1 ATTN_AB_SAVE=/tmp/ab0 UMV_ATTN_LAZY=0 timeout 300 python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || echo "FAILED rc=$?" >> $O/ab.txt
echo "== UMV_ATTN_LAZY=1" >> $O/ab.txt
ATTN_AB_REF=1 ATTN_AB_CMP=/tmp/ab0 UMV_ATTN_LAZY=1 timeout 300 python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || echo "FAILED rc=$?" >> $O/ab.txt
cat $O/ab.txt
