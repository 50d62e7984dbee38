#!/bin/bash
# tools/r3_ksweep.sh -- chain count on the contended pools (20 M reads at 1 600x / 6 400x / 25 600x, PhiX-like): chains stage per K, second warm run (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cov in 1600 6400 25600; do for K in 16384 32768 65536 131072 262144; do
  timeout 200 python tools/scale_probe.py 20000000,150,$K,10000,x,$cov 20000000,150,$K,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/cov=$cov /" | cut -c1-230
done; done
for K in 16384 32768 65536 131072 262144; do timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,$K 10000000,150,5400,$K 2>&1 | grep "^n=" | tail -1 | sed "s/^/PhiX /" | cut -c1-200; done
