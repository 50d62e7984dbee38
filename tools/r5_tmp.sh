O=gpurun_out/r5n; mkdir -p $O
timeout 2400 python tools/parity_deep.py 20000000,150,120000000,0,gen > $O/parity_gen20M.log 2>&1; tail -2 $O/parity_gen20M.log
timeout 1200 python tools/parity_10M.py > $O/parity_10M.log 2>&1; tail -3 $O/parity_10M.log
