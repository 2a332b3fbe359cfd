#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_third; mkdir -p $O
t() { local name=$1; shift; local t0=$(date +%s); timeout 1500 python -m pytest "$@" -q -m gpu 2>&1 | tail -40 > $O/$name.txt; echo "$name: $(( $(date +%s) - t0 )) s: $(tail -1 $O/$name.txt)" >> $O/summary.txt; }
t paged tests/test_paged_kv_gpu.py
t kernels_attn tests/test_kernels_gpu.py tests/test_kernel_branches_gpu.py -k "attn or qkv"
t serving tests/test_serving_gpu.py tests/test_kvcache_snapshot_gpu.py
B="python bench.py --no-t2i --no-vit --no-vae --no-load-path --no-cpu-baseline --no-fp8 --no-report --no-edit --no-sampled --steps 256"
for cfg in "0 0" "2 12" "2 8" "4 6" "4 4" "4 8" "2 16"; do set -- $cfg; echo "== wave_split $1 nsplit $2"; UMV_DECODE_WSPLIT=$1 UMV_DECODE_NSPLIT=$2 timeout 300 $B 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['value'], d['ms_per_step'])"; done > $O/wsplit.txt 2>&1
cat $O/summary.txt $O/wsplit.txt
