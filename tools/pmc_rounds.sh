#!/bin/bash
# tools/pmc_rounds.sh <outdir> <probe.py> <probe arg> COUNTER [COUNTER...]: one rocprofv3 --pmc pass over a probe run;
# prints the counters of the round kernel per launch, every 16th round (one line per launch: round, duration, counters).
set -eu
out=$1; probe=$2; arg=$3; shift 3
export TMPDIR=/tmp; mkdir -p "$out"
tag=$(echo "$*" | md5sum | cut -c1-6)
timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/p_$tag" -o pmc -- python "$probe" "$arg" > "$out/p_$tag.log" 2>&1 || true
python - "$out/p_$tag" <<'PY'
import csv, collections, sys, glob, os
csv.field_size_limit(1 << 30)
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "pmc_counter_collection.csv"), recursive=True)[0]
kt = glob.glob(os.path.join(d, "**", "pmc_kernel_trace.csv"), recursive=True)[0]
rows = collections.defaultdict(dict); names = set()
for r in csv.DictReader(open(cc)):
    if "k_round" not in r["Kernel_Name"]: continue
    rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); names.add(r["Counter_Name"])
dur = {}
for r in csv.DictReader(open(kt)):
    if "k_round" in r["Kernel_Name"]: dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = sorted(names)
print("round us " + " ".join(names))
for i, k in enumerate(sorted(rows)):
    if i % 16 == 0 or i < 6:
        print(i, "%.0f" % dur.get(k, 0), " ".join("%.4g" % rows[k].get(c, 0) for c in names))
PY
rm -rf "$out/p_$tag"
