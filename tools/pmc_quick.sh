#!/bin/bash
# tools/pmc_quick.sh <outdir> <reads>: one PMC pass (TCC read requests + L2 hits) over scale_probe
set -u
: "${1:?usage: see the header comment}"
out=$1; n=${2:-20000000}; export TMPDIR=/tmp; mkdir -p $out
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/tcc -o pmc -- python tools/scale_probe.py $n,150,65536 > $out/tcc.log 2>&1
python - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set); dur=collections.defaultdict(float)
for r in csv.DictReader(open("$out/tcc/pmc_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0][-24:]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for r in csv.DictReader(open("$out/tcc/pmc_kernel_trace.csv")):
    k=r["Kernel_Name"].split("(")[0][-24:]; dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for k in agg:
    if "k_search" in k or "k_apply" in k:
        n=len(cnt[k]); print(k, "launches", n, "avg_us %.1f"%(dur[k]/n), "rdreq/launch %.3fM"%(agg[k]["TCC_EA0_RDREQ_sum"]/n/1e6), "hit %.2f"%(agg[k]["TCC_HIT_sum"]/(agg[k]["TCC_HIT_sum"]+agg[k]["TCC_MISS_sum"])))
PY
