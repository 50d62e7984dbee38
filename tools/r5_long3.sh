#!/bin/bash
# tools/r5_long3.sh -- the three-kernel k_long: its parity tests, then genome-like / deep pools at several part sizes
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "long or genome_like or tuning" > $O/tests.log 2>&1; tail -3 $O/tests.log
for sp in 0 -1 64 128 512; do
  echo "== long_split=$sp" >> $O/sweep.txt
  SP_OPTS="long_split=$sp" timeout 120 python tools/scale_probe.py 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
done
echo "== 100M default" >> $O/sweep.txt
timeout 200 python tools/scale_probe.py 100000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
echo "== 100M long_split=-1" >> $O/sweep.txt
SP_OPTS="long_split=-1" timeout 200 python tools/scale_probe.py 100000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
echo "== 5M default" >> $O/sweep.txt
timeout 100 python tools/scale_probe.py 5000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
echo "== PhiX-like" >> $O/sweep.txt
timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 2>&1 | tail -2 >> $O/sweep.txt
cat $O/sweep.txt
