#!/bin/bash
# tools/r6_probe1.sh -- round 6: PMC passes over the headline configuration with the known-absent search, then probe-plan
# variants (units of four shifts = probes per lane and batch: 4/8/16 = 1, 2, 4, then 4s)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6_probe1; mkdir -p $O
bash tools/pmc_probe.sh $O/pmc 100000000 > $O/pmc.log 2>&1
python tools/pmc_aggregate.py $O/pmc 100000000 $O/pmc_100Mx150.json > $O/pmc_aggregate.txt 2>&1
rm -rf $O/pmc
python tools/ab_search.py 100000000,150 base= p488=plan0:4/8/8/16 p4416=plan0:4/4/16 p816=plan0:8/16 p416=plan0:4/16 p44816=plan0:4/4/8/16 p8816=plan0:8/8/16 p161616=plan0:16/16 base2= > $O/ab_plans.txt 2>&1
cat $O/pmc_aggregate.txt; cat $O/ab_plans.txt
