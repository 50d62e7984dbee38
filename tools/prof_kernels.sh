#!/bin/bash
# tools/prof_kernels.sh <outdir> [scale_probe args] -- rocprofv3 --kernel-trace --stats over one scale_probe run; keeps the kernel summary
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
: "${1:?usage: see the header comment}"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python tools/scale_probe.py ${@:-100000000,150,0} > $O/run.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
