O=gpurun_out/r5f; mkdir -p $O; rm -f $O/deep.txt
for o in "" "long_min=512" "long_min=1024" "long_budget=-1"; do
  echo "== PhiX-like $o" >> $O/deep.txt
  SP_OPTS="$o" timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 2>&1 | tail -1 >> $O/deep.txt
done
for cov in 400 1600 6400 25600; do
 for o in "" "long_budget=8" "long_budget=8,long_min=512"; do
  echo "== ${cov}x $o" >> $O/deep.txt
  SP_OPTS="$o" timeout 200 python tools/scale_probe.py 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' >> $O/deep.txt
 done
done
cat $O/deep.txt
