#!/bin/bash
# tools/pmc_caches.sh <outdir> <reads> -- instruction / scalar-data cache behaviour and issue cycles of the chain kernels
set -u
: "${1:?usage: see the header comment}"
out=$1; n=${2:-20000000}; export TMPDIR=/tmp; mkdir -p $out
run() { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o pmc -- python tools/scale_probe.py $n,150,0 > $out/$name.log 2>&1
}
run c1 SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH
run c2 SQ_WAVES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run c3 SQ_WAVES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
python - <<PY
import csv, collections
csv.field_size_limit(1<<30)
for p in ("c1","c2","c3"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    try:
        for r in csv.DictReader(open("$out/%s/pmc_counter_collection.csv"%p)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")[:22]
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    except Exception as e:
        print(p, "failed", e); continue
    for k in agg:
        if "k_round" in k:
            w=agg[k]["SQ_WAVES"] or 1
            print(p, k, "per wave:", " ".join("%s=%.1f"%(c.replace("SQ_","").replace("SQC_","C_"),v/w) for c,v in sorted(agg[k].items()) if c!="SQ_WAVES"))
PY
rm -rf $out/c1 $out/c2 $out/c3
