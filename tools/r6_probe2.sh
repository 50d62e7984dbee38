#!/bin/bash
# tools/r6_probe2.sh -- round 6: known-absent kernel without the minimizer array (LDS 7 488 -> 4 928 bytes per block), 5 against 6 waves
# per SIMD (x_w6.so: tools/xbuild.sh w6 -DSR_MC_WAVES=6), chain counts
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe2; mkdir -p $O
python tools/ab_search.py 100000000,150 base= off=known_absent:-1 > $O/ab_w5.txt 2>&1
SPRING_AMD_LIB=spring_amd/lib/x_w6.so python tools/ab_search.py 100000000,150 w6= > $O/ab_w6.txt 2>&1
for k in 81920 98304 131072; do python tools/ab_search.py 100000000,150,$k k$k= >> $O/ab_chains.txt 2>&1; done
(timeout 600 python -m pytest tests/test_gpu_phases.py -m gpu -x -q 2>&1 | tail -3) > $O/tests.txt
cat $O/ab_w5.txt $O/ab_w6.txt $O/ab_chains.txt $O/tests.txt
