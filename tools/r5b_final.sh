#!/bin/bash
# tools/r5b_final.sh -- round 5, second session: the closing measurement set with the two-group schedule (opts.phases) as the
# library's choice, on one MI355X box (through gpurun): PMC passes over the headline configuration (their summary becomes
# profiles/pmc_latest.json BEFORE the bench line is taken), the default bench line, the same command under rocprofv3
# --kernel-trace --stats, the two-group timeline from a kernel trace, one group against two on the headline pool and on
# the pools of the coverage sweep, the pool path at world = 1, parity at 20 M reads under two groups.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b_final; mkdir -p $O
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
cp $O/pmc_100Mx150.json profiles/pmc_latest.json
python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --files-sample 0 --cost-sample 0 --sweep-sample 0 > $O/bench_profiled.json 2> $O/prof.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
find $O/prof -name "*domain_stats.csv" -exec cp {} $O/bench_domain_stats.csv \;
rm -rf $O/prof
bash tools/phase_timeline.sh $O/tl 100000000,150,65536 > /dev/null 2>&1; cp $O/tl/phase_timeline.txt $O/phase_timeline_100M.txt; rm -rf $O/tl
{
for p in 1 2; do
SP_OPTS="phases=$p" python tools/scale_probe.py 100000000,150,65536 100000000,150,65536 2>&1 | grep "^n=" | tail -1 | sed "s/^/[phases=$p headline] /"
for n in 20000000 40000000; do SP_OPTS="phases=$p" python tools/scale_probe.py $n,150,0 $n,150,0 2>&1 | grep "^n=" | tail -1 | sed "s/^/[phases=$p 25x] /"; done
for cov in 100 400 1600 6400 25600; do SP_OPTS="phases=$p" python tools/scale_probe.py 20000000,150,0,10000,x,$cov 20000000,150,0,10000,x,$cov 2>&1 | grep "^n=" | tail -1 | sed "s/^/[phases=$p ${cov}x] /"; done
SP_OPTS="phases=$p" python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,0 2>&1 | grep "^n=" | tail -1 | sed "s/^/[phases=$p PhiX-like] /"
for a in 20000000 100000000; do SP_OPTS="phases=$p" python tools/scale_probe.py $a,150,0,10000,gen,25 $a,150,0,10000,gen,25 2>&1 | grep "^n=" | tail -1 | sed "s/^/[phases=$p genome-like] /"; done
done
} > $O/one_vs_two_groups.txt 2>&1
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2> $O/bench_pool.err
timeout 900 python tools/parity_10M.py 20000000 2 > $O/parity_20M_two_groups.txt 2>&1
timeout 600 python tools/files_probe.py 100000000 150 3 2>&1 | grep -v "^\[chains\]\|^\[dict\]" > $O/files_probe.txt
tail -2 $O/parity_20M_two_groups.txt; cut -c1-400 $O/bench.json
