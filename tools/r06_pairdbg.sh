#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_pairdbg; mkdir -p $O
for L in 1024 1026 1040 1055 1056; do echo "== L=$L"; L=$L REPS=20 timeout 600 python tools/attn_pair_debug.py 2>&1 | grep "^pair dbg"; done > $O/pair5.txt
cat $O/pair5.txt
