#!/bin/bash
# tools/pmc_any.sh <outdir> <reads> <lib.so or -> COUNTER [COUNTER...]: one rocprofv3 --pmc pass over scale_probe with the
# given counters; prints per-kernel totals per launch for the chain kernels.  PMC_ARGS: the scale_probe.py argument
# (default "<reads>,150,65536" = the headline pool; e.g. PMC_ARGS=20000000,150,0,10000,gen,25 for a genome-like pool).
set -u
out=$1; n=$2; lib=$3; shift 3
export TMPDIR=/tmp; mkdir -p "$out"
[ "$lib" != "-" ] && export SPRING_AMD_LIB=$lib
tag=$(echo "$*" | md5sum | cut -c1-6)
timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/p_$tag" -o pmc -- python tools/scale_probe.py "${PMC_ARGS:-$n,150,65536}" > "$out/p_$tag.log" 2>&1
python - "$out/p_$tag" <<'PY'
import csv, collections, sys, glob, os
csv.field_size_limit(1 << 30)
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
cc = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "pmc_kernel_trace.csv"), recursive=True)[0]
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for r in csv.DictReader(open(kt)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k in agg:
    if "k_round" in k or "k_mg" in k or "k_long" in k:
        n = len(cnt[k])
        print(k, "launches", n, "avg_us %.1f" % (dur[k] / n), "| per launch:", " ".join("%s=%.4g" % (c, v / n) for c, v in sorted(agg[k].items())))
PY
rm -rf "$out/p_$tag"
