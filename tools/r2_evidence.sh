#!/bin/bash
# tools/r2_evidence.sh -- the large one-off checks whose logs are committed under profiles/ (run through gpurun):
# oracle parity at 10 M and 50 M reads, BASELINE config 5's pool size on ONE GPU, real-BSC sizes vs chain count.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_evidence; mkdir -p $O
python tools/parity_10M.py 10000000 > $O/parity_10M.log 2>&1
python tools/scale_probe.py 400000000,150,65536 > $O/scale_400M_one_gpu.log 2>&1
python tools/compression_bsc.py 1000000 100 30 > $O/bsc_1Mx100.log 2>&1
KS=16,512,4096,0,65536 PORT8=0 python tools/compression_bsc.py 16000000 150 25 > $O/bsc_16Mx150.log 2>&1
python tools/parity_10M.py 50000000 > $O/parity_50M.log 2>&1
