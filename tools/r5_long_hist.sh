#!/bin/bash
# tools/r5_long_hist.sh -- k_long's scan-duration histogram and idle clocks (SR_LONG_COUNT build x_lc.so) on a genome-like pool,
# split searches off / on
O=gpurun_out/r5d; mkdir -p $O
for sp in 0 64 16; do
  echo "== long_split=$sp" >> $O/hist.txt
  SPRING_AMD_LIB=spring_amd/lib/x_lc.so SP_OPTS="long_split=$sp" python tools/scale_probe.py ${1:-20000000,150,0,10000,gen,25} 2>&1 | grep -E "^n=|k_long" >> $O/hist.txt
done
cat $O/hist.txt
