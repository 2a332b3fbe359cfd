#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_second; mkdir -p $O
t() { local name=$1; shift; local t0=$(date +%s); timeout 1500 python -m pytest "$@" -q -m gpu 2>&1 | tail -40 > $O/$name.txt; echo "$name: $(( $(date +%s) - t0 )) s: $(tail -1 $O/$name.txt)" >> $O/summary.txt; }
t attn_lazy tests/test_attn_lazy_gpu.py
t kv_growth tests/test_kv_growth_gpu.py
t sampling tests/test_kernels_gpu.py -k sampling
t bench_contract tests/test_bench_contract_gpu.py
t edit tests/test_fullwidth_gpu.py -k "edit_pipeline" -s
t fulldepth tests/test_fulldepth_gpu.py -s
timeout 900 python bench.py > $O/bench.txt 2>&1
cat $O/summary.txt
