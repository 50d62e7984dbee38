#!/bin/bash
# tools/prof_timeline.sh <outdir> [probe args] -- per-launch durations of the chain-phase kernels over a run
# (averages per 32 rounds, then the slowest launches).  PROBE=tools/deep_bins_probe.py selects the deep-pool generator.
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
: "${1:?usage: see the header comment}"
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=$1; shift; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$O/prof" -o p -- python ${PROBE:-tools/scale_probe.py} ${@:-100000000,150,0} > "$O/run.log" 2>&1 || true
python - "$O" <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
f = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:32]
    if "k_round" in k or "k_long" in k or "k_seeds" in k or "k_mg_mark" in k or "k_trim" in k:
        rows[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(O + "/timeline.txt", "w") as o:
    for k, v in rows.items():
        v.sort()
        o.write("%s launches %d total %.1f ms\n" % (k, len(v), sum(d for _, d in v) / 1e6))
        for i in range(0, len(v), 32):
            seg = v[i:i + 32]
            o.write("  rounds %5d..%5d  avg %8.1f us  max %8.1f us\n" % (i, i + len(seg) - 1, sum(d for _, d in seg) / len(seg) / 1e3, max(d for _, d in seg) / 1e3))
        o.write("  first 16: " + " ".join("%.0f" % (d / 1e3) for _, d in v[:16]) + "\n")
        o.write("  last 160: " + " ".join("%.0f" % (d / 1e3) for _, d in v[-160:]) + "\n")
PY
rm -rf "$O/prof"
