#!/bin/bash
# parity distributions of the flow / edit / prefill tests with the lazy-reference softmax (shipped) and with the exact-maximum kernels (UMV_ATTN_LAZY=0)
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_lazy_parity; mkdir -p $O
for lz in 1 0; do
  UMV_ATTN_LAZY=$lz timeout 1500 python -m pytest tests/test_fullwidth_gpu.py -q -m gpu -s -k "t2i or edit or configs1_prefill or flow" > $O/fullwidth_lazy$lz.txt 2>&1
  grep -E "latent deviation|pixels sample|edit|passed|failed|FAILED|Error" $O/fullwidth_lazy$lz.txt | cut -c1-330 > $O/summary_lazy$lz.txt
done
echo "== lazy"; cat $O/summary_lazy1.txt; echo "== exact"; cat $O/summary_lazy0.txt
