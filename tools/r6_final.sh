#!/bin/bash
# tools/r6_final.sh -- round 6: the closing measurement set on one MI355X box (through gpurun).  PMC passes over the headline
# configuration (their summary becomes profiles/pmc_latest.json BEFORE the bench line is taken), the default bench line, the same
# command under rocprofv3 --kernel-trace --stats, the two-group timeline, the known-absent masks on / off in one process, the
# 400 M pool through the multi-GPU path at world = 1 against one context, the drop-in's laps, parity at 20 M reads under the
# library's schedule, and the whole GPU test suite with its wall clock.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_final; mkdir -p $O
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
python tools/ab_search.py 100000000,150 masks_on= masks_off=known_absent:-1 masks_on_again= > $O/ab_known_absent.txt 2>&1
python bench.py --force-pool --steps 2 --no-single > $O/bench_pool400M_world1.json 2> $O/bench_pool.err
python tools/scale_probe.py 400000000,150,524288 400000000,150,524288 2>&1 | grep "^n=" > $O/single_context_400M.txt
timeout 600 python tools/files_probe.py 100000000 150 3 2>&1 | grep -v "^\[chains\]\|^\[dict\]" > $O/files_probe.txt
timeout 900 python tools/parity_10M.py 20000000 2 > $O/parity_20M_two_groups.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^Reordering done\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path\|reads with N" ) > $O/gputests.txt 2>&1
tail -2 $O/parity_20M_two_groups.txt; grep -a "passed\|failed" $O/gputests.txt | tail -2; cut -c1-300 $O/bench.json
