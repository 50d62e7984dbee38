#!/bin/bash
# real-BSC sizes of deep-coverage pools at the chain counts the default could take (tools/compression_bsc.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PORT8=0
for cov in 1600 6400 25600; do KS=65536,131072,156250 timeout 900 python tools/compression_bsc.py 20000000 150 $cov; echo; done
KS=65536,78125 timeout 900 python tools/compression_bsc.py 10000000 150 277777
