#!/bin/bash
# tools/r6_probe6.sh -- round 6: the new fuzz over virtual ranks; then: what would the four-chain kernel (no deep-bin machinery) do on the
# deeper pools of the sweep?  library choice against fused = 3, deep_bins = -1 at 100x .. 6 400x (20 M reads)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe6; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k "virtual_ranks" 2>&1 | tail -4 ) > $O/fuzz.txt
for cov in 100 400 1600 6400; do
SP_OPTS="" python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/[cov=$cov library] /"
SP_OPTS="fused=3,deep_bins=-1" python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/[cov=$cov four-chain, no deep-bin machinery] /"
SP_OPTS="fused=3,deep_bins=-1,known_absent=-1" python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/[cov=$cov four-chain, no masks] /"
done > $O/mc_on_deep.txt 2>&1
cat $O/fuzz.txt; cut -c1-200 $O/mc_on_deep.txt
