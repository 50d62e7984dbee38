#!/bin/bash
# tools/r6_probe4.sh -- round 6: the 400 M pool through the multi-GPU path at world = 1 (two chain groups, exchanges on their own stream) against
# one context on the same pool; then the whole GPU test suite with its wall clock
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe4; mkdir -p $O
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2> $O/bench_pool.err
python tools/scale_probe.py 400000000,150,524288 400000000,150,524288 2>&1 | grep "^n=" > $O/single_400M.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^Reordering done" | tail -6 ) > $O/gputests.txt 2>&1
cut -c1-1800 $O/bench_pool400M_world1.json; cat $O/single_400M.txt; cat $O/gputests.txt
