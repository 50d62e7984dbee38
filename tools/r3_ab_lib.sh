#!/bin/bash
# tools/r3_ab_lib.sh <other.so> -- A/B of two builds of the library on the deep pools, alternating on one box (chains stage)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in "" "$1"; do
  echo "lib=${lib:-default}"
  for cov in ${COVS:-400 1600 6400 25600}; do SPRING_AMD_LIB=$lib python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | cut -c1-150; done
  SPRING_AMD_LIB=$lib python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | cut -c1-120
done; done
