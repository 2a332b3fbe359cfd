#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_fullsuite; mkdir -p $O
timeout 3300 python -m pytest tests -q -m gpu -x > $O/tests.txt 2>&1
tail -8 $O/tests.txt
