#!/bin/bash
# tools/r6_probe5.sh -- round 6: with the known-absent masks, from which chain count does the four-chain kernel win?  25x and 100x pools of
# 10 .. 60 M reads (chain counts 9 765 .. 58 593), the library's mapping against fused = 3 (four chains per wavefront whatever the count)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe5; mkdir -p $O
for cov in 25 100; do
for n in 10000000 15000000 20000000 25000000 30000000 40000000 60000000; do
for f in 0 3; do
SP_OPTS="fused=$f" python tools/scale_probe.py $n,150,0,10000,x,$cov $n,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/[cov=$cov fused=$f] /"
done; done; done > $O/mc_threshold.txt 2>&1
cut -c1-170 $O/mc_threshold.txt
