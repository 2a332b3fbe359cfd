#!/bin/bash
# final stamp: whole GPU suite, profile (so that the bench line audits itself against the same stamp), bench line
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_final; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
bash tools/roofline_profile.sh r05 2>&1 | tail -3
cp gpurun_out/r05_roofline_profile.json profiles/roofline_profile_latest.json
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/bench_line.json
