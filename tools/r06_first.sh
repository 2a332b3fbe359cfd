#!/bin/bash
# round 6, first GPU session: the lazy-softmax rare-path tests, the peaked-attention engine test, the sampling fix, pair-form A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_first; mkdir -p $O
timeout 1500 python -m pytest tests/test_attn_lazy_gpu.py -x -q -m gpu 2>&1 | tail -30 > $O/attn_lazy.txt
timeout 1500 python -m pytest tests/test_peaked_engine_gpu.py -x -q -m gpu -s 2>&1 | tail -60 > $O/peaked.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "sampling" 2>&1 | tail -15 > $O/sampling.txt
timeout 900 python -m pytest tests/test_kernel_branches_gpu.py -q -m gpu -k "attn" 2>&1 | tail -15 > $O/attn_branches.txt
for lazy in 1 2 0; do echo "== UMV_ATTN_LAZY=$lazy"; UMV_ATTN_LAZY=$lazy timeout 300 python tools/attn_ab.py; done > $O/attn_ab.txt 2>&1
timeout 600 python bench.py > $O/bench.txt 2>&1
tail -5 $O/*.txt
