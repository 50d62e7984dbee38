#!/bin/bash
# tools/r5_final.sh -- round 5's closing measurement set on one MI355X box (through gpurun): GPU test suite, PMC passes over
# the headline configuration (incl. the L1 miss-queue counters; their summary becomes profiles/pmc_latest.json BEFORE the
# bench line is taken), the default bench line, the same command under rocprofv3 --kernel-trace --stats, the pool path at
# world = 1, coverage sweep + PhiX-like + genome-like pools, kernel stats of the genome-like pool.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_final; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gputests_full.txt 2>&1; grep -a "passed\|failed" $O/gputests_full.txt | tail -n 1 > $O/gputests.txt; cat $O/gputests.txt
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
cp $O/pmc_100Mx150.json profiles/pmc_latest.json
python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --files-sample 0 --cost-sample 0 --sweep-sample 0 > $O/bench_profiled.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/prof
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2> $O/bench_pool.err
for cov in 25 100 400 1600 6400 25600; do python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1; done > $O/coverage_sweep.txt
python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | sed 's/^/PhiX-like: /' >> $O/coverage_sweep.txt
for a in 5000000 20000000 100000000; do python tools/scale_probe.py $a,150,0,10000,gen,25 $a,150,0,10000,gen,25 2>&1 | grep "^n=" | tail -1 | sed 's/^/genome-like: /'; done >> $O/coverage_sweep.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o gen -- python tools/scale_probe.py 100000000,150,0,10000,gen,25 > $O/genomic_profiled.txt 2>&1
find $O/prof2 -name "*kernel_stats.csv" -exec cp {} $O/genomic_100M_kernel_stats.csv \;
rm -rf $O/prof2
rm -f $O/gputests_full.txt
# round 5: the drop-in's file legs (phase clocks), the PhiX-like pool's kernels, parity at size on the pools the long-search kernels serve
timeout 600 python tools/files_probe.py 100000000 150 3 2>&1 | grep -v "^\[chains\]\|^\[dict\]" > $O/files_probe.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -o phix -- python tools/deep_bins_probe.py 10000000,150,5400,0 > $O/phix_profiled.txt 2>&1
find $O/prof3 -name "*kernel_stats.csv" -exec cp {} $O/phix_like_kernel_stats.csv \;
rm -rf $O/prof3
