cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_evidence2; mkdir -p $O
python tools/parity_10M.py 10000000 > $O/parity_10M.log 2>&1
python tools/scale_probe.py 400000000,150,65536 > $O/scale_400M_one_gpu.log 2>&1
python tools/parity_10M.py 50000000 > $O/parity_50M.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
