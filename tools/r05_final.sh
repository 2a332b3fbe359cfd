#!/bin/bash
# final stamp of round 5: whole GPU suite (experimental library built), lazy-softmax A/B, bench line, profile
cd $GRAFT_REPO_ROOT; ulimit -c 0
O=gpurun_out/r05_final; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
bash tools/r05_lazy.sh > /dev/null 2>&1; cp gpurun_out/r05_lazy/ab.txt $O/lazy_ab.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 600 $O/bench_line.json
