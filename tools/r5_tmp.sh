O=gpurun_out/r5h; mkdir -p $O; rm -f $O/b.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "long or genome_like or tuning" > $O/tests.log 2>&1; tail -1 $O/tests.log
timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | tail -2 >> $O/b.txt
SP_OPTS="long_budget=-1" timeout 200 python tools/deep_bins_probe.py 10000000,150,5400,0 2>&1 | tail -1 >> $O/b.txt
for i in 1 2; do
timeout 200 python tools/scale_probe.py 5000000,150,0,10000,gen,25 20000000,150,0,10000,gen,25 100000000,150,0,10000,gen,25 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' >> $O/b.txt
done
SP_OPTS="long_budget=8" timeout 200 python tools/scale_probe.py 20000000,150,0,10000,x,400 20000000,150,0,10000,x,25600 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' >> $O/b.txt
timeout 200 python tools/scale_probe.py 20000000,150,0,10000,x,400 20000000,150,0,10000,x,25600 2>&1 | grep "^n=" | sed 's/search_ms.*lost/lost/' >> $O/b.txt
cat $O/b.txt
