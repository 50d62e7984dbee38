#!/bin/bash
# tools/r2_measure.sh -- the round's measurement set on one MI355X box (run through gpurun): GPU test suite, the
# default bench line, the same command under rocprofv3 --kernel-trace --stats, and the PMC passes.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_measure; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; grep -E "passed|failed|rror" $O/gputests.log | tail -5 > $O/gputests.summary
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --force-pool --pool-reads-per-gpu 50000000 --steps 2 > $O/bench_pool_world1.json 2> $O/bench_pool.err
python tools/pe_config4.py 1000000 50000000 > $O/pe_config4.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --files-sample 0 > $O/bench_profiled.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/prof
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
