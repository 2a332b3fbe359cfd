#!/bin/bash
# rocprofv3 kernel trace of one stage (tools/stage_profile.py) on the GPU box; summary -> gpurun_out/<tag>_kernel_stats.csv
# usage (inside gpurun): bash tools/profile_stage.sh <tag> <vit|prefill> [B]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o stage -- python tools/stage_profile.py "$@" > gpurun_out/$TAG/stage.log 2>&1 || true
grep ' ms per repetition' gpurun_out/$TAG/stage.log || tail -5 gpurun_out/$TAG/stage.log
DB=$(ls gpurun_out/$TAG/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_kernel_stats.csv
BY_GRID=1 python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_kernel_stats_by_grid.csv
rm -rf gpurun_out/$TAG
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.2f}us tot={int(r['TotalDurationNs'])/1e6:8.2f}ms {r['Percentage']:>6s}%")
PY
