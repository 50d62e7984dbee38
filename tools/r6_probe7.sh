#!/bin/bash
# tools/r6_probe7.sh -- round 6: the mark step with its independent loads issued together (k_ph_mark_wide): two-group tests, the headline's
# chain stage, the timeline
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe7; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_phases.py -m gpu -x -q 2>&1 | grep -v "^Reordering" | tail -3 ) > $O/tests.txt
python tools/ab_search.py 100000000,150 a= b= > $O/ab.txt 2>&1
bash tools/phase_timeline.sh $O/tl 100000000,150,65536 > /dev/null 2>&1; head -12 $O/tl/phase_timeline.txt > $O/timeline.txt; rm -rf $O/tl
cat $O/tests.txt $O/ab.txt $O/timeline.txt
