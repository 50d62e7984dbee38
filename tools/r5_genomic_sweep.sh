mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
bash tools/prof_timeline.sh $O/tl20 20000000,150,0,10000,gen,25 
cp $O/tl20/timeline.txt $O/timeline_gen20M.txt
for b in 0 2 4 16; do
  echo "== long_budget=$b" >> $O/sweep.txt
  SP_OPTS="long_budget=$b" python tools/scale_probe.py 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
done
for m in 256 512 1024 4096; do
  echo "== long_min=$m" >> $O/sweep.txt
  SP_OPTS="long_min=$m" python tools/scale_probe.py 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
done
for m in 256 1024; do
  echo "== long_budget=4 long_min=$m" >> $O/sweep.txt
  SP_OPTS="long_budget=4,long_min=$m" python tools/scale_probe.py 20000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
done
echo "== 100M default" >> $O/sweep.txt
python tools/scale_probe.py 100000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
echo "== 100M long_budget=4,long_min=512" >> $O/sweep.txt
SP_OPTS="long_budget=4,long_min=512" python tools/scale_probe.py 100000000,150,0,10000,gen,25 2>&1 | grep "^n=" >> $O/sweep.txt
cat $O/sweep.txt
