#!/bin/bash
# lazy-softmax build: attention tests first, then the whole GPU suite, then the leg timings
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_lazy_suite; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernel_branches_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "attn" > $O/attn_tests.txt 2>&1; tail -15 $O/attn_tests.txt
timeout 3300 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1
tail -25 $O/tests.txt
