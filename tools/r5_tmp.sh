O=gpurun_out/r5j; mkdir -p $O; rm -f $O/alt_perf.txt
for cov in 100 400 1600 6400 25600; do
 for o in "alternatives=1" "alternatives=2"; do
  echo "== ${cov}x $o" >> $O/alt_perf.txt
  SP_OPTS="$o" timeout 200 python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed 's/search_ms.*lost/lost/' >> $O/alt_perf.txt
 done
done
for o in "alternatives=1" "alternatives=2"; do
  echo "== PhiX-like $o" >> $O/alt_perf.txt
  SP_OPTS="$o" timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | tail -1 >> $O/alt_perf.txt
  echo "== genome-like 20M $o" >> $O/alt_perf.txt
  SP_OPTS="$o" timeout 200 python tools/scale_probe.py 20000000,150,0,10000,gen,25 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" | tail -1 | sed 's/search_ms.*lost/lost/' >> $O/alt_perf.txt
done
cat $O/alt_perf.txt
