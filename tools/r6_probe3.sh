#!/bin/bash
# tools/r6_probe3.sh -- round 6: the two new full-size tests (config 4 at 50 M pairs; 200 M-read pool through the 1-rank RCCL path) and a
# default bench.py run with the new legs (CPU baseline on the whole workload, the four-cell compression table)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe3; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or pool_200M" ) > $O/tests.txt 2>&1
( time python bench.py ) > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.txt; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
