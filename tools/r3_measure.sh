#!/bin/bash
# tools/r3_measure.sh -- round 3's measurement set on one MI355X box (run through gpurun): the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, the PMC passes over the headline configuration, the instruction mix,
# and the A/B of the two round kernels on the same box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3_measure; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --files-sample 0 --cost-sample 0 > $O/bench_profiled.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/prof
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
for mc in 3 2 3 2; do echo "fused=$mc (3: four chains per wavefront, 2: one)"; SP_OPTS=fused=$mc python tools/scale_probe.py 100000000,150,65536 2>&1 | tail -1; done > $O/ab_round_kernels.txt
python bench.py --force-pool --pool-reads 100000000 --pool-chains 65536 --steps 2 --no-single > $O/bench_pool_world1.json 2> $O/bench_pool.err
# the shared 400 M-read pool at world = 1 (what the N > 1 lines of the driver's scaling run are compared with)
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2>> $O/bench_pool.err
# throughput vs coverage (each pool twice: the second run is the warm one) and the PhiX-like pool
for cov in 25 400 1600 6400 25600; do python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1; done > $O/coverage_sweep.txt
python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | sed 's/^/PhiX-like: /' >> $O/coverage_sweep.txt
