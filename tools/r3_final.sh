#!/bin/bash
# tools/r3_final.sh -- the round's closing measurement set on one MI355X box (through gpurun): GPU test suite, PMC passes
# over the headline configuration (their summary becomes profiles/pmc_latest.json BEFORE the bench line is taken, so the
# line carries roofline.traffic for exactly these kernels), the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, the pool path at world = 1, the coverage sweep and the PhiX-like pool.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_final; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gputests_full.txt 2>&1; grep -a "passed\|failed" $O/gputests_full.txt | tail -n 1 > $O/gputests.txt; cat $O/gputests.txt
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
cp $O/pmc_100Mx150.json profiles/pmc_latest.json
python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --files-sample 0 --cost-sample 0 > $O/bench_profiled.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/prof
python bench.py --force-pool --pool-reads 100000000 --pool-chains 65536 --steps 2 --no-single > $O/bench_pool_world1.json 2> $O/bench_pool.err
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2>> $O/bench_pool.err
for cov in 25 100 400 1600 6400 25600; do python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1; done > $O/coverage_sweep.txt
python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | sed 's/^/PhiX-like: /' >> $O/coverage_sweep.txt
PROBE=tools/deep_bins_probe.py bash tools/prof_timeline.sh $O/tl_phix 10000000,150,5400,0 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o phix -- python tools/deep_bins_probe.py 10000000,150,5400,0 > $O/phix_profiled.txt 2>&1
find $O/prof2 -name "*kernel_stats.csv" -exec cp {} $O/phix_like_kernel_stats.csv \;
rm -rf $O/prof2
rm -f $O/gputests_full.txt
