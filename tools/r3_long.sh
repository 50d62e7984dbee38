#!/bin/bash
# tools/r3_long.sh -- k_long (long searches of deep-coverage pools finished by a block of 16 wavefronts): its parity tests,
# then the deep pools with the hand-over off (long_budget = -1) and at several budgets, each pool twice (the second
# run is the warm one).  Through gpurun; every step under its own timeout.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_long; mkdir -p $O
if [ "${1:-tests}" = "tests" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_searches or resumed or contended or default_chain_count" > $O/tests.txt 2>&1
  tail -3 $O/tests.txt
  timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "gpu_equals_oracle" > $O/fuzz.txt 2>&1
  tail -3 $O/fuzz.txt
fi
for b in ${BUDGETS:-0 24 8 64}; do
  lm=${b#*/}; b=${b%/*}; [ "$lm" = "$b" ] && lm=0
  echo "long_budget=$b long_min=$lm (0 = default)"; [ "$b" = 0 ] && b=-1
  SP_OPTS=long_budget=$b,long_min=$lm timeout 300 python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | sed 's/^/PhiX-like: /'
  for cov in ${COVS:-1600 6400 25600}; do SP_OPTS=long_budget=$b,long_min=$lm timeout 300 python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1; done
done > $O/ab_long.txt 2>&1
cat $O/ab_long.txt
