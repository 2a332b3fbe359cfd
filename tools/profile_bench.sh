#!/bin/bash
# rocprofv3 kernel trace of the default bench command on the GPU box; summary -> gpurun_out/<tag>_kernel_stats.csv
# usage (inside gpurun): bash tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG -o bench -- python bench.py --no-cpu-baseline "$@" > gpurun_out/$TAG/bench.log 2>&1 || true
grep '^{' gpurun_out/$TAG/bench.log > gpurun_out/${TAG}_bench_line.json || true
DB=$(ls gpurun_out/$TAG/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_kernel_stats.csv
rm -f "$DB"
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:80]:80s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.2f}us min={int(r['MinNs'])/1e3:8.2f} max={int(r['MaxNs'])/1e3:9.2f} {r['Percentage']:>6s}%")
PY
cut -c1-400 gpurun_out/${TAG}_bench_line.json
