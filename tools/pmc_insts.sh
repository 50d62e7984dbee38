#!/bin/bash
# tools/pmc_insts.sh <outdir> <reads> [lib.so] -- dynamic instruction mix of the chain kernels (one rocprofv3 --pmc pass each)
set -u
: "${1:?usage: see the header comment}"
out=$1; n=${2:-100000000}; export TMPDIR=/tmp; mkdir -p $out
[ -n "$3" ] && export SPRING_AMD_LIB=$3
run() { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o pmc -- python tools/scale_probe.py $n,150,65536 > $out/$name.log 2>&1
}
run i1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES
run i2 SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python - <<PY
import csv, collections, glob
csv.field_size_limit(1<<30)
for p in ("i1","i2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open("$out/%s/pmc_counter_collection.csv"%p)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:22]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for r in csv.DictReader(open("$out/%s/pmc_kernel_trace.csv"%p)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")[:22]; dur[k]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    with open("$out/%s.summary.txt"%p,"w") as o:
        for k in agg:
            if "k_round" in k or "k_mg" in k or "k_search" in k or "k_apply" in k:
                n=len(cnt[k]); w=agg[k]["SQ_WAVES"] or 1
                o.write("%s launches %d avg_us %.1f waves/launch %.0f | per wave: %s\n"%(k,n,dur[k]/n,w/n," ".join("%s=%.1f"%(c.replace("SQ_",""),v/w) for c,v in sorted(agg[k].items()) if c!="SQ_WAVES")))
PY
rm -rf $out/i1 $out/i2
